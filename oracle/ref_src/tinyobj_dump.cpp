// TEST INFRASTRUCTURE.  Driver around the reference's own vendored OBJ/MTL
// parser (tiny_obj_loader.h v1.1.1, compiled from where it lies under the
// reference checkout: see oracle/Makefile, target _ref/tinyobj_dump).  It calls
// LoadObj exactly as the reference does (scene.cpp:136-142: default
// triangulation, no mtl base dir, default material appended last) and dumps the
// parse result as a flat little-endian binary that tests/gen_golden.py reads.
//
// usage: tinyobj_dump scene.obj out.bin      (run with cwd = the obj's directory,
//        because the reference resolves `mtllib` relative to the cwd)
#define TINYOBJLOADER_IMPLEMENTATION
#include "tiny_obj_loader.h"
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

static void put_i(FILE* f, int32_t v) { fwrite(&v, 4, 1, f); }
static void put_f(FILE* f, float v) { fwrite(&v, 4, 1, f); }

int main(int argc, char** argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s scene.obj out.bin\n", argv[0]); return 2; }
    tinyobj::attrib_t attrib;
    std::vector<tinyobj::shape_t> shapes;
    std::vector<tinyobj::material_t> materials;
    std::string err;
    bool ret = tinyobj::LoadObj(&attrib, &shapes, &materials, &err, argv[1]);
    if (!ret) { fprintf(stderr, "LoadObj failed: %s\n", err.c_str()); return 1; }
    materials.push_back(tinyobj::material_t());   // scene.cpp:142
    FILE* f = fopen(argv[2], "wb");
    if (!f) return 1;
    put_i(f, 0x4a424f54);                          // magic
    put_i(f, (int32_t)(attrib.vertices.size() / 3));
    put_i(f, (int32_t)(attrib.normals.size() / 3));
    put_i(f, (int32_t)shapes.size());
    put_i(f, (int32_t)materials.size());
    for (float v : attrib.vertices) put_f(f, v);
    for (float v : attrib.normals) put_f(f, v);
    for (auto& s : shapes) {
        int32_t nf = (int32_t)(s.mesh.indices.size() / 3);
        put_i(f, nf);
        for (auto& idx : s.mesh.indices) { put_i(f, idx.vertex_index); put_i(f, idx.normal_index); }
        for (int32_t i = 0; i < nf; i++) put_i(f, s.mesh.material_ids[i]);
        for (int32_t i = 0; i < nf; i++) put_i(f, (int32_t)s.mesh.smoothing_group_ids[i]);
    }
    for (auto& m : materials) {
        for (int k = 0; k < 3; k++) put_f(f, m.diffuse[k]);
        for (int k = 0; k < 3; k++) put_f(f, m.specular[k]);
        for (int k = 0; k < 3; k++) put_f(f, m.emission[k]);
        put_f(f, m.shininess);
        put_f(f, m.ior);
        put_f(f, m.dissolve);
    }
    fclose(f);
    return 0;
}
