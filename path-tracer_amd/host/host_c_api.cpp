// C entry points over the host classes, for tests and scripting (ctypes).
#include <cstring>

#include "ray_tracer.hpp"

extern "C" {

// ---- loader only (no GPU) ----------------------------------------------------
Scene *pth_scene_load(const char *path)
{
    Scene *s = new Scene();
    if (!s->loadObj(path)) { delete s; return nullptr; }
    return s;
}
void pth_scene_free(Scene *s) { delete s; }
int pth_scene_triangle_count(const Scene *s) { return s->triangle_count; }
int pth_scene_shape_count(const Scene *s) { return (int)s->geos.size(); }
int pth_scene_material_count(const Scene *s) { return (int)s->materials_.size(); }
// flattened primitives in the order RayTracer::loadScene produces; returns the count
int pth_scene_flatten(const Scene *s, pt_PrimitiveRender *out)
{
    int n = 0;
    for (const auto &geo : s->geos)
        for (size_t pg = 0; pg < geo.indicess.size(); ++pg) {
            const auto &idx = geo.indicess[pg];
            for (size_t t = 0; t < idx.size() / 3; ++t) {
                pt_PrimitiveRender &p = out[n++];
                p.matid = (int32_t)geo.material_ids[pg];
                p.pad = -1;
                for (int k = 0; k < 3; k++) p.vertices[k] = geo.vertices_xyzn[idx[t * 3 + k]];
            }
        }
    return n;
}
// 12 floats per material: diffuse, specular, emission, shininess, ior, dissolve
void pth_scene_materials(const Scene *s, float *out)
{
    for (const auto &m : s->materials_) {
        std::memcpy(out, m.diffuse, 12); std::memcpy(out + 3, m.specular, 12); std::memcpy(out + 6, m.emission, 12);
        out[9] = m.shininess; out[10] = m.ior; out[11] = m.dissolve;
        out += 12;
    }
}
int pth_snapshot_due(int frame_count) { return RayTracer::snapshotDue(frame_count) ? 1 : 0; }
float pth_parse_real(const char *text) { return objl::parse_real(text, text + std::strlen(text), 0.0); }

// ---- RayTracer (GPU) -----------------------------------------------------------
RayTracer *pth_tracer_new(int device) { RayTracer *r = new RayTracer(); r->device = device; return r; }
void pth_tracer_free(RayTracer *r) { delete r; }
const char *pth_tracer_error(const RayTracer *r) { return r->lastError().c_str(); }
int pth_tracer_load(RayTracer *r, const char *obj, int w, int h)
{
    r->width = w; r->height = h;
    r->loadScene(obj);
    if (!r->ok()) return -1;
    r->loadShaders();
    r->createBuffers();
    return r->ok() ? r->primitive_count : -1;
}
int pth_tracer_run(RayTracer *r, int frames)
{
    for (int i = 0; i < frames && r->ok(); i++) r->run();
    return r->ok() ? r->frame_count : -1;
}
int pth_tracer_read(RayTracer *r, float *rgba32f, uint8_t *rgba8)
{
    if (rgba32f) { auto v = r->readOutput(); std::memcpy(rgba32f, v.data(), v.size() * 4); }
    if (rgba8) { auto v = r->readDisplay(); std::memcpy(rgba8, v.data(), v.size()); }
    return r->ok() ? 0 : -1;
}
int pth_tracer_frame_count(const RayTracer *r) { return r->frame_count; }
// what the application does to the public members: the ACC toggle writes frame_count (CS.cpp:524-527),
// the window callback calls resize (CS.cpp:205)
void pth_tracer_set_frame_count(RayTracer *r, int v) { r->frame_count = v; }
int pth_tracer_resize(RayTracer *r, int w, int h) { r->resize(w, h); return r->ok() ? 0 : -1; }
void pth_tracer_set_tiles(RayTracer *r, int tx, int ty) { r->tiles_x = tx; r->tiles_y = ty; }
void pth_tracer_set_tile_seed_per_dispatch(RayTracer *r, int on) { r->tile_seed_per_dispatch = on != 0; }
void pth_tracer_set_build_flags(RayTracer *r, unsigned flags) { r->build_flags = flags; }
int pth_tracer_rebuild(RayTracer *r) { r->buildRadixTree(); return r->ok() ? 0 : -1; }
int pth_tracer_members(RayTracer *r, pt_Bound *bb, pt_PrimitiveRay *sorted_rays, pt_TreeNode *tree)
{
    if (bb) *bb = r->bb_;
    if (sorted_rays) std::memcpy(sorted_rays, r->ray_primitives.data(), r->ray_primitives.size() * sizeof(pt_PrimitiveRay));
    if (tree) { auto t = r->readTree(); std::memcpy(tree, t.data(), t.size() * sizeof(pt_TreeNode)); }
    return r->ok() ? 0 : -1;
}

}  // extern "C"
