/*
 * TEST INFRASTRUCTURE -- not part of the product.
 *
 * CPU restatement (plain C) of the reference path tracer's hot path, used only
 * as the checker in tests/, in __graft_entry__.smoke() and as the
 * "cpu_baseline" leg of bench.py.  Nothing in the shipped library links,
 * loads or calls this code.
 *
 * PARITY UNPINNED for the kernels (topology, refit, RNG, traversal, shading,
 * accumulate): the reference ships no tests, golden vectors or known-answer
 * data for them, and its HLSL cannot be compiled or run here, so these
 * functions are pinned only by following the reference source text cited at
 * each function.  The OBJ/MTL parse is pinned by the reference's own vendored
 * parser compiled from where it lies (oracle/_ref, see oracle/Makefile); the
 * converged image is compared with the reference's one shipped snapshot.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stdint.h>
#include "../include/ptamd_structs.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scene flattening (scene.cpp:134-290, ray_tracer.cpp:23-48) --------- */
typedef struct orc_mesh {
    int n_vertices, n_normals;         /* counts of xyz triples */
    const float *vertices;             /* attrib.vertices */
    const float *normals;              /* attrib.normals  */
    int n_shapes;
    const int *shape_face_offset;      /* n_shapes+1 prefix offsets into the face arrays */
    const int *face_v;                 /* 3 vertex indices per triangle */
    const int *face_vn;                /* 3 normal indices per triangle (-1 = none) */
    const int *face_mat;               /* material id per triangle (may be <0) */
    const int *face_smooth;            /* smoothing group id per triangle */
    int n_materials;                   /* INCLUDING the default material appended last */
} orc_mesh;
/* returns the triangle count written to out (capacity = total faces) */
int orc_flatten(const orc_mesh *m, pt_PrimitiveRender *out);

/* ---- acceleration-structure build --------------------------------------- */
void     orc_bounds(const pt_PrimitiveRender *prims, int n, pt_PrimitiveRay *out);
void     orc_scene_bound(const pt_PrimitiveRay *rays, int n, pt_Bound *bb);
uint64_t orc_morton(const pt_Bound *small_bound, const pt_Bound *big_bound);
void     orc_morton_all(pt_PrimitiveRay *rays, int n, const pt_Bound *bb);
/* ascending by morton_code, ties by ascending original index (stable) */
void     orc_sort(pt_PrimitiveRay *rays, int n);
void     orc_strip(const pt_PrimitiveRay *rays, int n, pt_PrimitiveRayCS *out);
/* tree has 2n+1 entries; entry 2n is the all-zero out-of-bounds node */
void     orc_radix_tree(int n, pt_TreeNode *tree);           /* textbook loops   */
void     orc_radix_tree_literal(int n, pt_TreeNode *tree);   /* HLSL int32 loops */
void     orc_radix_tree_codes(int n, const uint64_t *codes, pt_TreeNode *tree);   /* NON-PARITY: keyed on the sorted codes */
void     orc_refit(int n, const pt_PrimitiveRayCS *sorted, pt_TreeNode *tree);

/* ---- RNG ------------------------------------------------------------------ */
uint32_t orc_rand_init(uint32_t val0, uint32_t val1);
float    orc_rand_next(uint32_t *s);
/* seed of frame f = output number 16*f of the 32-bit Mersenne twister seeded
 * with base_seed (the reference draws 16 values per frame from its generator
 * and the shader reads the first: ray_tracer.cpp:293-318) */
void     orc_frame_seeds(uint32_t base_seed, uint32_t first_frame, uint32_t count, uint32_t *out);

/* ---- G-buffer by ray casting (stands in for fixed_pipeline.hlsl) ---------- */
typedef struct orc_camera {
    float eye[3], at[3], up[3];
    float fovy, znear, zfar;
} orc_camera;

typedef struct orc_scene {
    int n;                              /* node_count = triangle count */
    const pt_PrimitiveRender *prims;    /* n    */
    const pt_TreeNode *tree;            /* 2n+1 */
    const pt_CB_Radix *cb;
    int stack_size;                     /* 15 = reference */
    int max_segments;                   /* rays per path at most; 0 = reference (14: itr reaches MAX_ITR * 3) */
    int near_first;                     /* 0 = reference (left child first); 1 = NON-PARITY nearer child first */
} orc_scene;

typedef struct orc_counters {
    uint64_t paths, rays, node_visits, box_tests, tri_tests;
} orc_counters;

/* diffuse_rgba8: W*H*4 bytes, normal_matid/worldpos: W*H*4 floats.
 * albedo8[mat][3] are the bytes the colour target holds for each material. */
void orc_albedo_bytes(const float kd[3], int srgb_quirk, uint8_t out[3]);
void orc_primary(const orc_scene *s, const orc_camera *cam, int W, int H,
                 const uint8_t *albedo8 /* [16][4] */,
                 uint8_t *diffuse_rgba8, float *normal_matid, float *worldpos,
                 int y0, int y1);

/* ---- one frame of ray_tracing.hlsl:CSMain --------------------------------- */
/* literal: out = pow(pow(old,2.2)*f/(f+1) + c/(f+1), 1/2.2), plus unorm8 copy.
 * rows [y0,y1) only (for threading). radiance (optional) receives this frame's
 * linear sample, 3 floats per pixel. */
void orc_trace_frame(const orc_scene *s, int W, int H,
                     const uint8_t *diffuse_rgba8, const float *normal_matid, const float *worldpos,
                     uint32_t framecount, uint32_t seed,
                     const float *old_rgba, float *out_rgba, uint8_t *out_unorm,
                     float *radiance, orc_counters *cnt, int y0, int y1);
/* linear sum: acc.rgb += sample ; acc.a += 1 */
void orc_trace_frame_linear(const orc_scene *s, int W, int H,
                            const uint8_t *diffuse_rgba8, const float *normal_matid, const float *worldpos,
                            uint32_t framecount, uint32_t seed,
                            float *acc_rgba, orc_counters *cnt, int y0, int y1);
/* out = pow(acc.rgb / frames, 1/2.2), alpha 1; plus unorm8 */
void orc_resolve(int W, int H, const float *acc_rgba, uint32_t frames,
                 float *out_rgba, uint8_t *out_unorm);

/* single closest-hit query, for unit tests: returns 1 on hit */
int orc_intersect_one(const orc_scene *s, const float origin[3], const float dir[3],
                      float t_min, float t_max, int *prim, float bary[3], float *t,
                      orc_counters *cnt);

/* elementary functions of include/ptamd_math.h, exported for accuracy tests */
float orc_sinf(float x);
float orc_cosf(float x);
float orc_exp2f(float x);
float orc_log2f(float x);
float orc_powf(float x, float y);

#ifdef __cplusplus
}
#endif
#endif
