/*
 * Interchange buffer layouts of the path tracer's hot path.
 *
 * These are the byte layouts that cross the library boundary (host <-> device
 * and caller <-> library).  They are bit-identical to the reference's shared
 * host/shader structs (reference: structs.hpp:2-77, structs.hlsli:1-40); the
 * kernels are free to use other layouts internally.
 *
 * Plain C, no dependencies; usable from C, C++ and HIP device code.
 */
#ifndef PTAMD_STRUCTS_H
#define PTAMD_STRUCTS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference: math3d.h:3-11 */
typedef struct pt_vec3 { float v[3]; } pt_vec3;
typedef struct pt_vec4 { float v[4]; } pt_vec4;

/* reference: structs.hpp:2-7 / structs.hlsli:1-5 -- 28 bytes */
typedef struct pt_Bound {
    pt_vec3 min;
    float   padding0;
    pt_vec3 max;
} pt_Bound;

/* reference: structs.hpp:10-18 / structs.hlsli:7-14 -- 48 bytes.
 * Internal node i (0 <= i < n): left/right children, parent.
 * Leaf n+k: index = primitive id, bound = primitive box.  Root is node 0 and
 * parent[0] stays 0.  A child id >= n denotes a leaf. */
typedef struct pt_TreeNode {
    int32_t  left;
    int32_t  right;
    int32_t  parent;
    int32_t  index;
    pt_Bound bound;
    int32_t  padding;
} pt_TreeNode;

/* reference: structs.hpp:26-30 / structs.hlsli:22-26 -- 24 bytes */
typedef struct pt_xyzn {
    pt_vec3 pos;
    pt_vec3 normal;
} pt_xyzn;

/* reference: structs.hpp:33-38 / structs.hlsli:28-33 -- 80 bytes */
typedef struct pt_PrimitiveRender {
    pt_xyzn vertices[3];
    int32_t matid;
    int32_t pad;            /* the reference host writes -1 (ray_tracer.cpp:36) */
} pt_PrimitiveRender;

/* reference: structs.hpp:40-45 -- 40 bytes, host-side only */
typedef struct pt_PrimitiveRay {
    uint64_t morton_code;
    pt_Bound bound;
    int32_t  index;
} pt_PrimitiveRay;

/* reference: structs.hpp:48-52 / structs.hlsli:16-20 -- 32 bytes */
typedef struct pt_PrimitiveRayCS {
    pt_Bound bound;
    int32_t  index;
} pt_PrimitiveRayCS;

/* reference: structs.hpp:53-60 / ray_tracing.hlsl:17-24 -- 48 bytes */
typedef struct pt_Material {
    float   ke;
    float   ns;
    float   tr;
    float   ni;
    pt_vec4 kd;
    pt_vec4 ks;
} pt_Material;

#define PT_MAX_MATERIALS 16

/* reference: structs.hpp:62-68 / ray_tracing.hlsl:27-32 -- 784 bytes */
typedef struct pt_CB_Radix {
    int32_t     node_count;
    pt_vec3     pos;                    /* eye position (g_pos) */
    pt_Material mats[PT_MAX_MATERIALS];
} pt_CB_Radix;

/* reference: structs.hpp:70-77 / ray_tracing.hlsl:33-40 -- 48 bytes */
typedef struct pt_CB_RT {
    float    viewportDims[2];
    float    tanHalfFovY;
    uint32_t framecount;
    int32_t  offset[4];
    int32_t  random[4];
} pt_CB_RT;

/* reference: structs.hpp:79 */
#define PT_MORTON_CODE_LENGTH_EACH 21

#ifdef __cplusplus
}
static_assert(sizeof(pt_Bound) == 28, "Bound layout");
static_assert(sizeof(pt_TreeNode) == 48, "TreeNode layout");
static_assert(sizeof(pt_xyzn) == 24, "xyzn layout");
static_assert(sizeof(pt_PrimitiveRender) == 80, "PrimitiveRender layout");
static_assert(sizeof(pt_PrimitiveRay) == 40, "PrimitiveRay layout");
static_assert(sizeof(pt_PrimitiveRayCS) == 32, "PrimitiveRayCS layout");
static_assert(sizeof(pt_Material) == 48, "Material layout");
static_assert(sizeof(pt_CB_Radix) == 784, "CB_Radix layout");
static_assert(sizeof(pt_CB_RT) == 48, "CB_RT layout");
#else
_Static_assert(sizeof(pt_Bound) == 28, "Bound layout");
_Static_assert(sizeof(pt_TreeNode) == 48, "TreeNode layout");
_Static_assert(sizeof(pt_xyzn) == 24, "xyzn layout");
_Static_assert(sizeof(pt_PrimitiveRender) == 80, "PrimitiveRender layout");
_Static_assert(sizeof(pt_PrimitiveRay) == 40, "PrimitiveRay layout");
_Static_assert(sizeof(pt_PrimitiveRayCS) == 32, "PrimitiveRayCS layout");
_Static_assert(sizeof(pt_Material) == 48, "Material layout");
_Static_assert(sizeof(pt_CB_Radix) == 784, "CB_Radix layout");
_Static_assert(sizeof(pt_CB_RT) == 48, "CB_RT layout");
#endif

#endif /* PTAMD_STRUCTS_H */
