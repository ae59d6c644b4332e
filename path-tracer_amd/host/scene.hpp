// Scene: OBJ -> per-shape geometry with per-material index lists, as the
// reference's Scene class exposes it (reference: scene.hpp:7-42).  Headless: the
// D3D vertex/index buffers of the reference (buildBuffers) have no counterpart.
#pragma once
#include <cstdint>
#include <vector>

#include "obj_loader.hpp"
#include "structs.hpp"

class Scene
{
public:
    // Loads and converts the OBJ (reference: scene.cpp:134-290): per shape, faces are
    // grouped by material in order of first use; normals come from the file, else from
    // smoothing groups, else from the face; x of positions and normals is negated; a
    // default material is appended after the file's materials.  Returns false and keeps
    // `error` on failure (the reference asserts).
    bool loadObj(const char *filename);
    void release();

    struct geometry
    {
        std::vector<xyzn> vertices_xyzn;
        std::vector<xyz> vertices_pos;
        std::vector<std::vector<uint32_t>> indicess;   // one index list per material group
        std::vector<uint32_t> material_ids;            // material of each group
    };

    std::vector<geometry> geos;
    std::vector<objl::MaterialDef> materials_;
    int triangle_count = 0;
    std::string error;
};
