#include "scene.hpp"

#include <cmath>
#include <map>

namespace {

// unit face normal, left as is when degenerate (reference: scene.cpp:29-54)
void face_normal(float N[3], const float a[3], const float b[3], const float c[3])
{
    const float u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    const float w[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    N[0] = w[1] * u[2] - w[2] * u[1];
    N[1] = w[2] * u[0] - w[0] * u[2];
    N[2] = w[0] * u[1] - w[1] * u[0];
    const float len2 = N[0] * N[0] + N[1] * N[1] + N[2] * N[2];
    if (len2 > 0.0f) {
        const float len = sqrtf(len2);
        N[0] /= len; N[1] /= len; N[2] /= len;
    }
}

}  // namespace

bool Scene::loadObj(const char *filename)
{
    objl::Mesh mesh;
    if (!objl::load_obj(filename, mesh, error)) return false;
    materials_ = mesh.materials;
    // default material, always last.  The reference appends a value-initialised record
    // (scene.cpp:142), i.e. ALL fields zero -- not the parser's defaults.
    objl::MaterialDef zero;
    zero.shininess = zero.ior = zero.dissolve = 0.0f;
    materials_.push_back(zero);
    const int n_mat = (int)materials_.size();
    auto clamp_material = [&](int m) { return (m < 0 || m >= n_mat) ? n_mat - 1 : m; };

    for (const objl::Shape &shape : mesh.shapes) {
        geometry geo;
        const size_t faces = shape.indices.size() / 3;
        std::map<int, int> group_of;                     // material -> group, in order of first use
        for (size_t f = 0; f < faces; f++) {
            const int m = clamp_material(shape.material_ids[f]);
            if (!group_of.count(m)) {
                const int g = (int)group_of.size();
                group_of[m] = g;
                geo.material_ids.push_back((uint32_t)m);
            }
        }
        geo.indicess.resize(group_of.size());

        // smoothing-group normals: sum of unit face normals per vertex, normalised
        bool smooth = false;
        for (unsigned id : shape.smoothing_ids) if (id > 0) { smooth = true; break; }
        std::map<int, vec3> smooth_normal;
        if (smooth) {
            for (size_t f = 0; f < faces; f++) {
                const int vi[3] = {shape.indices[3 * f].v, shape.indices[3 * f + 1].v, shape.indices[3 * f + 2].v};
                float nrm[3];
                face_normal(nrm, &mesh.vertices[3 * vi[0]], &mesh.vertices[3 * vi[1]], &mesh.vertices[3 * vi[2]]);
                for (int i = 0; i < 3; i++) {
                    auto it = smooth_normal.find(vi[i]);
                    if (it != smooth_normal.end()) { it->second.v[0] += nrm[0]; it->second.v[1] += nrm[1]; it->second.v[2] += nrm[2]; }
                    else smooth_normal[vi[i]] = vec3{{nrm[0], nrm[1], nrm[2]}};
                }
            }
            for (auto &kv : smooth_normal) {
                float *v = kv.second.v;
                const float len2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
                if (len2 > 0.0f) { const float len = sqrtf(len2); v[0] /= len; v[1] /= len; v[2] /= len; }
            }
        }

        for (size_t f = 0; f < faces; f++) {
            const objl::Index *c = &shape.indices[3 * f];
            vec3 pos[3], normal[3];
            for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) pos[i].v[k] = mesh.vertices[3 * c[i].v + k];
            bool need_normal = true;
            if (!mesh.normals.empty() && c[0].vn >= 0 && c[1].vn >= 0 && c[2].vn >= 0) {
                for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) normal[i].v[k] = mesh.normals[3 * c[i].vn + k];
                need_normal = false;
            }
            if (need_normal && !smooth_normal.empty()) {
                for (int i = 0; i < 3; i++) normal[i] = smooth_normal[c[i].v];
                need_normal = false;
            }
            if (need_normal) {
                face_normal(normal[0].v, pos[0].v, pos[1].v, pos[2].v);
                normal[1] = normal[0];
                normal[2] = normal[0];
            }
            const int g = group_of[clamp_material(shape.material_ids[f])];
            for (int i = 0; i < 3; i++) {
                pos[i].v[0] = -pos[i].v[0];               // the reference mirrors x
                normal[i].v[0] = -normal[i].v[0];
                geo.indicess[g].push_back((uint32_t)geo.vertices_pos.size());
                geo.vertices_pos.push_back(xyz{pos[i]});
                geo.vertices_xyzn.push_back(xyzn{pos[i], normal[i]});
            }
            triangle_count++;
        }
        geos.push_back(geo);
    }
    return true;
}

void Scene::release()
{
    geos.clear();
    materials_.clear();
    triangle_count = 0;
}
