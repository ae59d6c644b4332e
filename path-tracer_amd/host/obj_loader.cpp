// OBJ / MTL reader of the host facade.  Own code, except that read_double below follows the decimal
// reader of tinyobjloader v1.1.1 (tryParseDouble; Copyright (c) 2012-2018 Syoyo Fujita and many
// contributors, MIT licence -- the parser the reference application vendors as tiny_obj_loader.h):
// parsed floats must match that parser bit for bit, so its digit-by-digit arithmetic is reproduced.
#include "obj_loader.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace objl {
namespace {

inline bool is_digit(char c) { return c >= '0' && c <= '9'; }
inline bool is_blank(char c) { return c == ' ' || c == '\t'; }

// The decimal reader of the reference's parser: digits are accumulated into a
// double one at a time (integer part: m = m*10 + d; fraction: m += d * 10^-k
// with a table for k < 8), an exponent is applied as ldexp(m * 5^e, e), and the
// result is narrowed to float.  Not strtod: the last bit can differ.
bool read_double(const char *s, const char *end, double *result)
{
    if (s >= end) return false;
    double mantissa = 0.0;
    int exponent = 0;
    char sign = '+', exp_sign = '+';
    const char *cur = s;
    if (*cur == '+' || *cur == '-') { sign = *cur; cur++; }
    else if (!is_digit(*cur)) return false;
    int read = 0;
    while (cur != end && is_digit(*cur)) { mantissa = mantissa * 10 + (*cur - '0'); cur++; read++; }
    if (read == 0) return false;
    bool have_exp = false;
    if (cur != end) {
        if (*cur == '.') {
            cur++;
            read = 1;
            static const double lut[] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
            while (cur != end && is_digit(*cur)) {
                mantissa += (*cur - '0') * (read < 8 ? lut[read] : std::pow(10.0, -read));
                read++;
                cur++;
            }
            have_exp = (cur != end && (*cur == 'e' || *cur == 'E'));
        } else if (*cur == 'e' || *cur == 'E') {
            have_exp = true;
        }
    }
    if (have_exp) {
        cur++;
        if (cur != end && (*cur == '+' || *cur == '-')) { exp_sign = *cur; cur++; }
        else if (cur == end || !is_digit(*cur)) return false;
        read = 0;
        while (cur != end && is_digit(*cur)) { exponent = exponent * 10 + (*cur - '0'); cur++; read++; }
        if (exp_sign == '-') exponent = -exponent;
        if (read == 0) return false;
    }
    *result = (sign == '+' ? 1 : -1) * (exponent ? std::ldexp(mantissa * std::pow(5.0, exponent), exponent) : mantissa);
    return true;
}

// cursor over one line
struct Cursor {
    const char *p;
    void skip_blank() { while (is_blank(*p)) p++; }
    // next blank-delimited token as [begin, end)
    void token(const char *&b, const char *&e)
    {
        skip_blank();
        b = p;
        while (*p && !is_blank(*p) && *p != '\r') p++;
        e = p;
    }
    float real(double fallback = 0.0)
    {
        const char *b, *e;
        token(b, e);
        return parse_real(b, e, fallback);
    }
    int integer()
    {
        skip_blank();
        int v = std::atoi(p);
        while (*p && !is_blank(*p) && *p != '\r') p++;
        return v;
    }
    bool at_end() const { return *p == '\0' || *p == '\r' || *p == '\n'; }
};

// 1-based / negative-relative OBJ index -> 0-based.  Zero is invalid.  A negative index counts back from the
// `count` elements read so far and must name one of them.  A positive index may point FORWARD (the reference's
// parser does no range check, tiny_obj_loader.h fixIndex): it is checked later, against the elements present
// when it is first dereferenced -- vertices when the face's group is closed and triangulated (faces_in_range),
// normals at the end of the file (load_obj).
bool fix_index(int idx, int count, int &out, bool dereferenced = true)
{
    if (idx > 0) { out = idx - 1; return true; }
    if (idx == 0) return false;
    out = count + idx;
    return !dereferenced || out >= 0;
}

void skip_index_field(const char *&p) { p += std::strcspn(p, "/ \t\r"); }

// one "v", "v/vt", "v//vn" or "v/vt/vn" reference
bool read_corner(const char *&p, int nv, int nvn, int nvt, Index &out)
{
    Index c;
    if (!fix_index(std::atoi(p), nv, c.v)) return false;
    skip_index_field(p);
    if (*p != '/') { out = c; return true; }
    p++;
    if (*p == '/') {
        p++;
        if (!fix_index(std::atoi(p), nvn, c.vn)) return false;
        skip_index_field(p);
        out = c;
        return true;
    }
    if (!fix_index(std::atoi(p), nvt, c.vt, false)) return false;     // texture coordinates are parsed, never read
    skip_index_field(p);
    if (*p != '/') { out = c; return true; }
    p++;
    if (!fix_index(std::atoi(p), nvn, c.vn)) return false;
    skip_index_field(p);
    out = c;
    return true;
}

struct Face { std::vector<Index> corners; unsigned smoothing = 0; };

// even-odd point-in-polygon test (used on the candidate ear)
bool point_in_polygon(int n, const float *px, const float *py, float x, float y)
{
    bool inside = false;
    for (int i = 0, j = n - 1; i < n; j = i++)
        if (((py[i] > y) != (py[j] > y)) && (x < (px[j] - px[i]) * (y - py[i]) / (py[j] - py[i]) + px[i])) inside = !inside;
    return inside;
}

void emit_triangle(Shape &shape, const Index &a, const Index &b, const Index &c, int material, unsigned smoothing)
{
    shape.indices.push_back(a);
    shape.indices.push_back(b);
    shape.indices.push_back(c);
    shape.material_ids.push_back(material);
    shape.smoothing_ids.push_back(smoothing);
}

// every vertex a face of the group names exists by now (the group is about to be triangulated from `v`)
bool faces_in_range(const std::vector<Face> &faces, const std::vector<float> &v)
{
    const int nv = (int)(v.size() / 3);
    for (const Face &face : faces)
        for (const Index &c : face.corners)
            if (c.v < 0 || c.v >= nv) return false;
    return true;
}

// Appends the faces to the shape as triangles.  Polygons are ear-clipped in the
// coordinate plane chosen from the first corner whose cross product exceeds
// 1e-4 in some component (drop x when |cx| is the strict maximum, else drop y
// unless |cz| is the strict maximum, in which case keep x and y).
bool append_faces(Shape &shape, const std::vector<Face> &faces, int material, const std::string &name,
                  const std::vector<float> &v)
{
    if (faces.empty()) return false;
    for (const Face &face : faces) {
        size_t npoly = face.corners.size();
        size_t ax0 = 1, ax1 = 2;
        for (size_t k = 0; k < npoly; k++) {
            const size_t i0 = (size_t)face.corners[k % npoly].v, i1 = (size_t)face.corners[(k + 1) % npoly].v,
                         i2 = (size_t)face.corners[(k + 2) % npoly].v;
            const float e0x = v[i1 * 3] - v[i0 * 3], e0y = v[i1 * 3 + 1] - v[i0 * 3 + 1], e0z = v[i1 * 3 + 2] - v[i0 * 3 + 2];
            const float e1x = v[i2 * 3] - v[i1 * 3], e1y = v[i2 * 3 + 1] - v[i1 * 3 + 1], e1z = v[i2 * 3 + 2] - v[i1 * 3 + 2];
            const float cx = std::fabs(e0y * e1z - e0z * e1y);
            const float cy = std::fabs(e0z * e1x - e0x * e1z);
            const float cz = std::fabs(e0x * e1y - e0y * e1x);
            const float eps = 0.0001f;
            if (cx > eps || cy > eps || cz > eps) {
                if (!(cx > cy && cx > cz)) {
                    ax0 = 0;
                    if (cz > cx && cz > cy) ax1 = 1;
                }
                break;
            }
        }
        float area = 0;
        for (size_t k = 0; k < npoly; k++) {
            const size_t i0 = (size_t)face.corners[k % npoly].v, i1 = (size_t)face.corners[(k + 1) % npoly].v;
            area += (v[i0 * 3 + ax0] * v[i1 * 3 + ax1] - v[i0 * 3 + ax1] * v[i1 * 3 + ax0]) * 0.5f;
        }
        std::vector<Index> rest = face.corners;
        int rounds_left = 10;
        size_t guess = 0;
        while (rest.size() > 3 && rounds_left > 0) {
            npoly = rest.size();
            if (guess >= npoly) { rounds_left--; guess -= npoly; }
            Index ear[3];
            float px[3], py[3];
            for (size_t k = 0; k < 3; k++) {
                ear[k] = rest[(guess + k) % npoly];
                px[k] = v[(size_t)ear[k].v * 3 + ax0];
                py[k] = v[(size_t)ear[k].v * 3 + ax1];
            }
            const float cross = (px[1] - px[0]) * (py[2] - py[1]) - (py[1] - py[0]) * (px[2] - px[1]);
            if (cross * area < 0.0f) { guess++; continue; }          // reflex corner
            bool blocked = false;
            for (size_t o = 3; o < npoly; o++) {
                const size_t vi = (size_t)rest[(guess + o) % npoly].v;
                if (point_in_polygon(3, px, py, v[vi * 3 + ax0], v[vi * 3 + ax1])) { blocked = true; break; }
            }
            if (blocked) { guess++; continue; }
            emit_triangle(shape, ear[0], ear[1], ear[2], material, face.smoothing);
            rest.erase(rest.begin() + (long)((guess + 1) % npoly));
        }
        if (rest.size() == 3) emit_triangle(shape, rest[0], rest[1], rest[2], material, face.smoothing);
    }
    shape.name = name;
    return true;
}

void load_mtl(std::istream &in, std::vector<MaterialDef> &materials, std::map<std::string, int> &by_name)
{
    MaterialDef cur;
    bool has_d = false;
    std::string line;
    while (std::getline(in, line)) {
        const size_t last = line.find_last_not_of(" \t\r\n");
        if (last == std::string::npos) continue;
        line.erase(last + 1);
        Cursor c{line.c_str()};
        c.skip_blank();
        const char *t = c.p;
        if (*t == '\0' || *t == '#') continue;
        auto key = [&](const char *k) { const size_t n = std::strlen(k); return std::strncmp(t, k, n) == 0 && is_blank(t[n]); };
        if (key("newmtl")) {
            if (!cur.name.empty()) { by_name.insert({cur.name, (int)materials.size()}); materials.push_back(cur); }
            cur = MaterialDef();
            has_d = false;
            cur.name = t + 7;
        } else if (key("Ka")) { c.p = t + 2; for (float &x : cur.ambient) x = c.real(); }
        else if (key("Kd")) { c.p = t + 2; for (float &x : cur.diffuse) x = c.real(); }
        else if (key("Ks")) { c.p = t + 2; for (float &x : cur.specular) x = c.real(); }
        else if (key("Kt") || key("Tf")) { c.p = t + 2; for (float &x : cur.transmittance) x = c.real(); }
        else if (key("Ke")) { c.p = t + 2; for (float &x : cur.emission) x = c.real(); }
        else if (key("Ni")) { c.p = t + 2; cur.ior = c.real(); }
        else if (key("Ns")) { c.p = t + 2; cur.shininess = c.real(); }
        else if (key("illum")) { c.p = t + 6; cur.illum = c.integer(); }
        else if (key("d")) { c.p = t + 1; cur.dissolve = c.real(); has_d = true; }
        else if (key("Tr")) { c.p = t + 2; if (!has_d) cur.dissolve = 1.0f - c.real(); }     // `d` wins over `Tr`
    }
    by_name.insert({cur.name, (int)materials.size()});
    materials.push_back(cur);
}

}  // namespace

float parse_real(const char *begin, const char *end, double fallback)
{
    double v = fallback;
    read_double(begin, end, &v);
    return static_cast<float>(v);
}

bool load_obj(const std::string &path, Mesh &out, std::string &err)
{
    out = Mesh();
    std::ifstream in(path.c_str());
    if (!in) { err = "cannot open " + path; return false; }
    const size_t slash = path.find_last_of("/\\");
    const std::string dir = (slash == std::string::npos) ? std::string() : path.substr(0, slash + 1);

    std::map<std::string, int> by_name;
    std::vector<Face> group;
    std::string name;
    int material = -1;
    unsigned smoothing = 0;
    Shape shape;
    std::string line;
    while (std::getline(in, line)) {
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        if (line.empty()) continue;
        Cursor c{line.c_str()};
        c.skip_blank();
        const char *t = c.p;
        if (*t == '\0' || *t == '#') continue;
        if (t[0] == 'v' && is_blank(t[1])) {
            c.p = t + 2;
            for (int k = 0; k < 3; k++) out.vertices.push_back(c.real());
        } else if (t[0] == 'v' && t[1] == 'n' && is_blank(t[2])) {
            c.p = t + 3;
            for (int k = 0; k < 3; k++) out.normals.push_back(c.real());
        } else if (t[0] == 'v' && t[1] == 't' && is_blank(t[2])) {
            c.p = t + 3;
            for (int k = 0; k < 2; k++) out.texcoords.push_back(c.real());
        } else if (t[0] == 'f' && is_blank(t[1])) {
            const char *p = t + 2;
            p += std::strspn(p, " \t");
            Face face;
            face.smoothing = smoothing;
            while (*p != '\0' && *p != '\r' && *p != '\n') {
                Index corner;
                if (!read_corner(p, (int)(out.vertices.size() / 3), (int)(out.normals.size() / 3),
                                 (int)(out.texcoords.size() / 2), corner)) {
                    err = "bad face index in: " + line;
                    return false;
                }
                face.corners.push_back(corner);
                p += std::strspn(p, " \t\r");
            }
            group.push_back(face);
        } else if (std::strncmp(t, "usemtl", 6) == 0 && is_blank(t[6])) {
            const auto it = by_name.find(std::string(t + 7));
            const int next = (it == by_name.end()) ? -1 : it->second;
            if (next != material) {
                // materials are per face: the faces read so far go into the current shape
                if (!faces_in_range(group, out.vertices)) { err = "face names a vertex that does not exist when its group ends"; return false; }
                append_faces(shape, group, material, name, out.vertices);
                group.clear();
                material = next;
            }
        } else if (std::strncmp(t, "mtllib", 6) == 0 && is_blank(t[6])) {
            std::stringstream names(std::string(t + 7));
            std::string file;
            while (std::getline(names, file, ' ')) {
                if (file.empty()) continue;
                std::ifstream m(file.c_str());
                if (!m) m.open((dir + file).c_str());
                if (m) { load_mtl(m, out.materials, by_name); break; }
            }
        } else if (t[0] == 'g' && is_blank(t[1])) {
            if (!faces_in_range(group, out.vertices)) { err = "face names a vertex that does not exist when its group ends"; return false; }
            append_faces(shape, group, material, name, out.vertices);
            if (!shape.indices.empty()) out.shapes.push_back(shape);
            shape = Shape();
            group.clear();
            Cursor names{t + 2};
            const char *b, *e;
            names.token(b, e);
            name = std::string(b, e);
        } else if (t[0] == 'o' && is_blank(t[1])) {
            if (!faces_in_range(group, out.vertices)) { err = "face names a vertex that does not exist when its group ends"; return false; }
            if (append_faces(shape, group, material, name, out.vertices)) out.shapes.push_back(shape);
            shape = Shape();
            group.clear();
            name = t + 2;
        } else if (t[0] == 's' && is_blank(t[1])) {
            const char *p = t + 2;
            p += std::strspn(p, " \t");
            if (*p == '\0' || *p == '\r') continue;
            if (std::strlen(p) >= 3) {
                // three or more characters: only "off" is understood, anything else leaves the id unchanged
                if (p[0] == 'o' && p[1] == 'f' && p[2] == 'f') smoothing = 0;
            } else {
                const int id = std::atoi(p);
                smoothing = id < 0 ? 0u : (unsigned)id;
            }
        }
    }
    if (!faces_in_range(group, out.vertices)) { err = "face names a vertex that does not exist when its group ends"; return false; }
    const bool any = append_faces(shape, group, material, name, out.vertices);
    if (any || !shape.indices.empty()) out.shapes.push_back(shape);
    // normals are read only after the whole file has been parsed (scene.cpp:134-290), so a face may name any
    // normal of the file
    const int nvn = (int)(out.normals.size() / 3);
    for (const Shape &sh : out.shapes)
        for (const Index &c : sh.indices)
            if (c.vn >= nvn) { err = "face names a normal that the file does not have"; return false; }
    return true;
}

}  // namespace objl
