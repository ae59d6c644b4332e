// Wavefront OBJ / MTL reader for the subset the tracer's scenes use.
//
// The reference parses its scenes with tinyobjloader v1.1.1 (vendored header,
// reference: tiny_obj_loader.h:1227-1611 LoadMtl, :1703-2041 LoadObj,
// :1012-1213 face export + ear clipping, :520-633 number parsing).  This is an
// independent implementation of the same published behaviour, including the
// quirks that decide the bytes of the flattened scene: the digit-by-digit
// decimal parser, ear clipping of polygons in the plane of the first
// non-degenerate corner, shapes cut at `g` / `o`, per-face materials cut at
// `usemtl`, `d` winning over `Tr`, and smoothing-group ids longer than two
// characters being ignored.
#pragma once
#include <string>
#include <vector>

namespace objl {

struct Index { int v = -1, vt = -1, vn = -1; };

struct Shape {
    std::string name;
    std::vector<Index> indices;            // 3 per triangle
    std::vector<int> material_ids;         // per triangle
    std::vector<unsigned> smoothing_ids;   // per triangle
};

struct MaterialDef {
    std::string name;
    float ambient[3] = {0, 0, 0}, diffuse[3] = {0, 0, 0}, specular[3] = {0, 0, 0};
    float transmittance[3] = {0, 0, 0}, emission[3] = {0, 0, 0};
    float shininess = 1.f, ior = 1.f, dissolve = 1.f;
    int illum = 0;
};

struct Mesh {
    std::vector<float> vertices;           // xyz
    std::vector<float> normals;            // xyz
    std::vector<float> texcoords;          // uv
    std::vector<Shape> shapes;
    std::vector<MaterialDef> materials;
};

// Returns false (and a message) when the file cannot be opened or a face has a
// zero index.  `mtllib` files are looked up relative to the working directory
// first (as the reference does) and then next to the OBJ file.
bool load_obj(const std::string &path, Mesh &out, std::string &err);

// exposed for tests
float parse_real(const char *begin, const char *end, double fallback);

}  // namespace objl
