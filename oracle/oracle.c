/*
 * TEST INFRASTRUCTURE -- CPU restatement of the reference hot path (see
 * oracle.h for scope and for what is and is not pinned).  Plain C99; build
 * with -O2 -ffp-contract=off and no fast-math (oracle/Makefile).
 *
 * Every function names the reference lines it follows.  HLSL constructs with
 * implementation-defined precision (sin, cos, pow, normalize, reflect,
 * min/max) take the meaning fixed in include/ptamd_math.h.
 */
#include "oracle.h"
#include "../include/ptamd_math.h"

#include <stdlib.h>
#include <string.h>

#define ORC_MAX_STACK 128

/* ======================================================================= */
/* scene flattening                                                        */
/* ======================================================================= */

/* scene.cpp:29-54 */
static void calc_normal(float N[3], const float v0[3], const float v1[3], const float v2[3])
{
    float v10[3], v20[3];
    v10[0] = v1[0] - v0[0]; v10[1] = v1[1] - v0[1]; v10[2] = v1[2] - v0[2];
    v20[0] = v2[0] - v0[0]; v20[1] = v2[1] - v0[1]; v20[2] = v2[2] - v0[2];
    N[0] = v20[1] * v10[2] - v20[2] * v10[1];
    N[1] = v20[2] * v10[0] - v20[0] * v10[2];
    N[2] = v20[0] * v10[1] - v20[1] * v10[0];
    float len2 = N[0] * N[0] + N[1] * N[1] + N[2] * N[2];
    if (len2 > 0.0f) {
        float len = sqrtf(len2);
        N[0] /= len; N[1] /= len; N[2] /= len;
    }
}

/* scene.cpp:134-290 (per-shape material groups in order of first use,
 * normals: file normals, else smoothing-group normals, else geometric;
 * x of position and normal negated) followed by ray_tracer.cpp:27-48
 * (shape -> material group -> face order, global index, pad = -1). */
int orc_flatten(const orc_mesh *m, pt_PrimitiveRender *out)
{
    int total = 0;
    for (int s = 0; s < m->n_shapes; s++) {
        int f0 = m->shape_face_offset[s], f1 = m->shape_face_offset[s + 1];
        int nf = f1 - f0;
        /* material -> group, in order of first appearance (scene.cpp:149-165) */
        int *group_mat = (int *)malloc(sizeof(int) * (size_t)(nf > 0 ? nf : 1));
        int *face_group = (int *)malloc(sizeof(int) * (size_t)(nf > 0 ? nf : 1));
        int n_groups = 0;
        for (int f = 0; f < nf; f++) {
            int mat = m->face_mat[f0 + f];
            if (mat < 0 || mat >= m->n_materials) mat = m->n_materials - 1;
            int g = -1;
            for (int k = 0; k < n_groups; k++) if (group_mat[k] == mat) { g = k; break; }
            if (g < 0) { g = n_groups; group_mat[n_groups++] = mat; }
            face_group[f] = g;
        }
        /* smoothing normals (scene.cpp:17-27, 70-131) */
        int has_smooth = 0;
        for (int f = 0; f < nf; f++) if (m->face_smooth[f0 + f] > 0) { has_smooth = 1; break; }
        float *sn = NULL; unsigned char *sn_set = NULL;
        if (has_smooth) {
            sn = (float *)calloc((size_t)m->n_vertices * 3, sizeof(float));
            sn_set = (unsigned char *)calloc((size_t)m->n_vertices, 1);
            for (int f = 0; f < nf; f++) {
                const int *vi = &m->face_v[3 * (f0 + f)];
                float nrm[3];
                calc_normal(nrm, &m->vertices[3 * vi[0]], &m->vertices[3 * vi[1]], &m->vertices[3 * vi[2]]);
                for (int i = 0; i < 3; i++) {
                    float *d = &sn[3 * vi[i]];
                    if (sn_set[vi[i]]) { d[0] += nrm[0]; d[1] += nrm[1]; d[2] += nrm[2]; }
                    else { d[0] = nrm[0]; d[1] = nrm[1]; d[2] = nrm[2]; sn_set[vi[i]] = 1; }
                }
            }
            for (int v = 0; v < m->n_vertices; v++) if (sn_set[v]) {
                float *d = &sn[3 * v];
                float len2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
                if (len2 > 0.0f) { float len = sqrtf(len2); d[0] /= len; d[1] /= len; d[2] /= len; }
            }
        }
        /* emit group by group, faces in file order inside a group */
        for (int g = 0; g < n_groups; g++) {
            for (int f = 0; f < nf; f++) {
                if (face_group[f] != g) continue;
                const int *vi = &m->face_v[3 * (f0 + f)];
                const int *ni = &m->face_vn[3 * (f0 + f)];
                float pos[3][3], nrm[3][3];
                for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) pos[i][k] = m->vertices[3 * vi[i] + k];
                int invalid = 0;
                if (m->n_normals > 0) {
                    if (ni[0] < 0 || ni[1] < 0 || ni[2] < 0) invalid = 1;
                    else for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) nrm[i][k] = m->normals[3 * ni[i] + k];
                } else invalid = 1;
                if (invalid && has_smooth) {
                    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++)
                        nrm[i][k] = sn_set[vi[i]] ? sn[3 * vi[i] + k] : 0.0f;
                    invalid = 0;
                }
                if (invalid) {
                    calc_normal(nrm[0], pos[0], pos[1], pos[2]);
                    memcpy(nrm[1], nrm[0], sizeof(nrm[0]));
                    memcpy(nrm[2], nrm[0], sizeof(nrm[0]));
                }
                pt_PrimitiveRender *p = &out[total++];
                for (int i = 0; i < 3; i++) {
                    p->vertices[i].pos.v[0] = -pos[i][0];
                    p->vertices[i].pos.v[1] = pos[i][1];
                    p->vertices[i].pos.v[2] = pos[i][2];
                    p->vertices[i].normal.v[0] = -nrm[i][0];
                    p->vertices[i].normal.v[1] = nrm[i][1];
                    p->vertices[i].normal.v[2] = nrm[i][2];
                }
                p->matid = group_mat[g];
                p->pad = -1;
            }
        }
        free(group_mat); free(face_group); free(sn); free(sn_set);
    }
    return total;
}

/* ======================================================================= */
/* build                                                                   */
/* ======================================================================= */

/* the Windows min/max macros used by math3d.h / structs.hpp */
static float mmin(float a, float b) { return (a < b) ? a : b; }
static float mmax(float a, float b) { return (a > b) ? a : b; }

/* ray_tracer.cpp:41-44, math3d.h:13-43: max(x, max(y, z)) per component */
void orc_bounds(const pt_PrimitiveRender *prims, int n, pt_PrimitiveRay *out)
{
    for (int i = 0; i < n; i++) {
        const pt_PrimitiveRender *p = &prims[i];
        memset(&out[i], 0, sizeof(out[i]));
        out[i].index = i;
        for (int k = 0; k < 3; k++) {
            float x = p->vertices[0].pos.v[k], y = p->vertices[1].pos.v[k], z = p->vertices[2].pos.v[k];
            out[i].bound.max.v[k] = mmax(x, mmax(y, z));
            out[i].bound.min.v[k] = mmin(x, mmin(y, z));
        }
    }
}

/* ray_tracer.cpp:50-54, structs.hpp:107-116 */
void orc_scene_bound(const pt_PrimitiveRay *rays, int n, pt_Bound *bb)
{
    *bb = rays[0].bound;
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            bb->min.v[k] = mmin(bb->min.v[k], rays[i].bound.min.v[k]);
            bb->max.v[k] = mmax(bb->max.v[k], rays[i].bound.max.v[k]);
        }
}

/* structs.hpp:81-105 -- strict IEEE evaluation of the source expression */
uint64_t orc_morton(const pt_Bound *sb, const pt_Bound *bb)
{
    uint64_t code = 0;
    float max_x = bb->max.v[0], min_x = bb->min.v[0];
    float max_y = bb->max.v[1], min_y = bb->min.v[1];
    float max_z = bb->max.v[2], min_z = bb->min.v[2];
    float midx = (sb->min.v[0] + sb->max.v[0]) / 2;
    float midy = (sb->min.v[1] + sb->max.v[1]) / 2;
    float midz = (sb->min.v[2] + sb->max.v[2]) / 2;
    const int L = PT_MORTON_CODE_LENGTH_EACH;
    /* An axis on which the scene box is flat gives 0/0 = NaN, and converting NaN to an integer is
     * undefined in C.  x86 (the reference's platform) converts it to 0x8000000000000000, of which the
     * interleave below uses bits 21..0: zero.  That value is made explicit here. */
    float qa = ((midx - min_x) / (max_x - min_x)) * L;
    float qb = ((midy - min_y) / (max_y - min_y)) * L;
    float qc = ((midz - min_z) / (max_z - min_z)) * L;
    uint64_t a = (qa != qa) ? 0 : (uint64_t)qa;
    uint64_t b = (qb != qb) ? 0 : (uint64_t)qb;
    uint64_t c = (qc != qc) ? 0 : (uint64_t)qc;
    for (int j = 0; j < L; j++) {
        code |= (((a >> (L - 1 - j)) & 1) << ((L - j) * 3 - 1)) |
                (((b >> (L - 1 - j)) & 1) << ((L - j) * 3 - 2)) |
                (((c >> (L - 1 - j)) & 1) << ((L - j) * 3 - 3));
    }
    return code;
}

void orc_morton_all(pt_PrimitiveRay *rays, int n, const pt_Bound *bb)
{
    for (int i = 0; i < n; i++) rays[i].morton_code = orc_morton(&rays[i].bound, bb);
}

/* ray_tracer.cpp:61 sorts ascending by code with an unstable sort; the order
 * of equal codes is unspecified there.  Contract here: ties keep ascending
 * original index. */
static int cmp_ray(const void *pa, const void *pb)
{
    const pt_PrimitiveRay *a = (const pt_PrimitiveRay *)pa, *b = (const pt_PrimitiveRay *)pb;
    if (a->morton_code != b->morton_code) return a->morton_code < b->morton_code ? -1 : 1;
    return (a->index > b->index) - (a->index < b->index);
}
void orc_sort(pt_PrimitiveRay *rays, int n) { qsort(rays, (size_t)n, sizeof(*rays), cmp_ray); }

/* ray_tracer.cpp:62-66 */
void orc_strip(const pt_PrimitiveRay *rays, int n, pt_PrimitiveRayCS *out)
{
    for (int i = 0; i < n; i++) { out[i].bound = rays[i].bound; out[i].index = rays[i].index; }
}

/* radix_build_cs.hlsl:18-25: keyed on the leaf INDEX, range [0, len) */
static int lcp(int i, int j, int len)
{
    if (0 <= j && j < len) return __builtin_clz((unsigned)(i ^ j));   /* 31 - firstbithigh */
    return -1;
}

/* radix_build_cs.hlsl:74-92: child links and parent back-links; index 2n is
 * outside the reference's 2n-element buffer (writes dropped): slot 2n of our
 * 2n+1 array is kept all-zero. */
static void link_node(pt_TreeNode *tree, int n, int i, int j, int gamma)
{
    int lo = i < j ? i : j, hi = i < j ? j : i;
    int left  = (lo == gamma)     ? gamma + n     : gamma;
    int right = (hi == gamma + 1) ? n + gamma + 1 : gamma + 1;
    tree[i].left = left;
    tree[i].right = right;
    if (left < 2 * n) tree[left].parent = i;
    if (right < 2 * n) tree[right].parent = i;
}

/* radix_build_cs.hlsl:30-96 with the search loops in their usual form */
void orc_radix_tree(int n, pt_TreeNode *tree)
{
    const int len = n + 1;
    for (int i = 0; i < n; i++) {
        int d = (lcp(i, i + 1, len) - lcp(i, i - 1, len) > 0) ? 1 : -1;
        int sig_min = lcp(i, i - d, len);
        int lmax = 2;
        while (lcp(i, i + lmax * d, len) > sig_min) lmax *= 2;
        int l = 0;
        for (int t = lmax / 2; t >= 1; t /= 2)
            if (lcp(i, i + (l + t) * d, len) > sig_min) l += t;
        int j = i + l * d;
        int sig_node = lcp(i, j, len);
        int s = 0, t = l;
        do {
            t = (t + 1) >> 1;
            if (lcp(i, i + (s + t) * d, len) > sig_node) s += t;
        } while (t > 1);
        int gamma = i + s * d + (d < 0 ? d : 0);
        link_node(tree, n, i, j, gamma);
    }
}

/* NON-PARITY option, no counterpart in the reference (whose tree is keyed on the leaf index, above): the binary
 * radix tree over the SORTED MORTON CODES themselves, equal codes told apart by their position (Karras 2012,
 * "Maximizing parallelism in the construction of BVHs, octrees, and k-d trees", section 4).  n keys give n - 1
 * internal nodes 0..n-2 with root 0; key k is leaf n + k; slots n-1 and 2n stay zero.  Same search loops and the
 * same naming of children by the split position as orc_radix_tree. */
static int lcp_codes(const uint64_t *codes, int i, int j, int n)
{
    if (j < 0 || j >= n) return -1;
    uint64_t x = codes[i] ^ codes[j];
    if (x) return __builtin_clzll(x);
    return 64 + __builtin_clz((unsigned)(i ^ j));      /* i != j here */
}
void orc_radix_tree_codes(int n, const uint64_t *codes, pt_TreeNode *tree)
{
    for (int i = 0; i < n - 1; i++) {
        int d = (lcp_codes(codes, i, i + 1, n) - lcp_codes(codes, i, i - 1, n) > 0) ? 1 : -1;
        int sig_min = lcp_codes(codes, i, i - d, n);
        int lmax = 2;
        while (lcp_codes(codes, i, i + lmax * d, n) > sig_min) lmax *= 2;
        int l = 0;
        for (int t = lmax / 2; t >= 1; t /= 2)
            if (lcp_codes(codes, i, i + (l + t) * d, n) > sig_min) l += t;
        int j = i + l * d;
        int sig_node = lcp_codes(codes, i, j, n);
        int s = 0, t = l;
        do {
            t = (t + 1) >> 1;
            if (lcp_codes(codes, i, i + (s + t) * d, n) > sig_node) s += t;
        } while (t > 1);
        int gamma = i + s * d + (d < 0 ? d : 0);
        link_node(tree, n, i, j, gamma);
    }
}

/* The same kernel with its loops as written (radix_build_cs.hlsl:51-72):
 * `for (t = X/divider; t >= 1; divider *= 2) { ...; t = X/divider; }` repeats
 * the first step, and the split search idles at t = 1 until `divider` wraps
 * through int32.  Integer ops wrap; x/0 yields -1 here (loop then ends). */
static int32_t wrap_add(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static int32_t wrap_mul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static int32_t hl_div(int32_t a, int32_t b)
{
    if (b == 0) return -1;
    if (a == INT32_MIN && b == -1) return INT32_MIN;
    return a / b;
}
void orc_radix_tree_literal(int n, pt_TreeNode *tree)
{
    const int len = n + 1;
    for (int i = 0; i < n; i++) {
        int d = (lcp(i, i + 1, len) - lcp(i, i - 1, len) > 0) ? 1 : -1;
        int sig_min = lcp(i, i - d, len);
        int32_t lmax = 2;
        while (lcp(i, wrap_add(i, wrap_mul(lmax, d)), len) > sig_min) lmax = wrap_mul(lmax, 2);
        int32_t l = 0, divider = 2;
        for (int32_t t = hl_div(lmax, divider); t >= 1; divider = wrap_mul(divider, 2)) {
            if (lcp(i, wrap_add(i, wrap_mul(wrap_add(l, t), d)), len) > sig_min) l = wrap_add(l, t);
            t = hl_div(lmax, divider);
        }
        int j = i + l * d;
        int sig_node = lcp(i, j, len);
        int32_t s = 0;
        divider = 2;
        for (int32_t t = hl_div(wrap_add(l, wrap_add(divider, -1)), divider); t >= 1; divider = wrap_mul(divider, 2)) {
            if (lcp(i, wrap_add(i, wrap_mul(wrap_add(s, t), d)), len) > sig_node) s = wrap_add(s, t);
            t = hl_div(wrap_add(l, wrap_add(divider, -1)), divider);
        }
        int gamma = i + s * d + (d < 0 ? d : 0);
        link_node(tree, n, i, j, gamma);
    }
}

/* bvh_build_cs.hlsl:17-55.  The shader's lock protocol is racy and mis-addressed;
 * the meaning pinned here is its intended fixpoint: every leaf holds its sorted
 * primitive, every internal node the union (HLSL min/max) of its two children,
 * and the out-of-range child 2n reads as the all-zero node. */
static void refit_node(int n, pt_TreeNode *tree, int i)
{
    int ch[2] = { tree[i].left, tree[i].right };
    for (int c = 0; c < 2; c++)
        if (ch[c] < n) refit_node(n, tree, ch[c]);
    const pt_Bound *L = &tree[ch[0]].bound, *R = &tree[ch[1]].bound;
    for (int k = 0; k < 3; k++) {
        tree[i].bound.min.v[k] = pt_min(L->min.v[k], R->min.v[k]);
        tree[i].bound.max.v[k] = pt_max(L->max.v[k], R->max.v[k]);
    }
}
void orc_refit(int n, const pt_PrimitiveRayCS *sorted, pt_TreeNode *tree)
{
    for (int i = 0; i < n; i++) {
        tree[n + i].index = sorted[i].index;
        tree[n + i].bound = sorted[i].bound;
    }
    memset(&tree[2 * n].bound, 0, sizeof(pt_Bound));
    tree[2 * n].index = 0;
    if (n > 0) refit_node(n, tree, 0);
}

/* ======================================================================= */
/* RNG                                                                     */
/* ======================================================================= */

/* helpers.hlsli:5-19 */
uint32_t orc_rand_init(uint32_t val0, uint32_t val1)
{
    uint32_t v0 = val0, v1 = val1, s0 = 0;
    for (uint32_t r = 0; r < 16; r++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}

/* helpers.hlsli:26-32 */
float orc_rand_next(uint32_t *s)
{
    const uint32_t LCG_A = 1664525u, LCG_C = 1013904223u;
    *s = LCG_A * (*s) + LCG_C;
    return (float)(*s & 0x00FFFFFFu) / (float)0x01000000;
}

/* MT19937 (the generator std::mt19937 specifies), raw 32-bit outputs */
typedef struct { uint32_t mt[624]; int idx; } mt_state;
static void mt_seed(mt_state *s, uint32_t seed)
{
    s->mt[0] = seed;
    for (int i = 1; i < 624; i++) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = 624;
}
static uint32_t mt_next(mt_state *s)
{
    if (s->idx >= 624) {
        for (int i = 0; i < 624; i++) {
            uint32_t y = (s->mt[i] & 0x80000000u) | (s->mt[(i + 1) % 624] & 0x7fffffffu);
            s->mt[i] = s->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        s->idx = 0;
    }
    uint32_t y = s->mt[s->idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}
void orc_frame_seeds(uint32_t base_seed, uint32_t first_frame, uint32_t count, uint32_t *out)
{
    mt_state st; mt_seed(&st, base_seed);
    uint64_t skip = (uint64_t)first_frame * 16u;
    for (uint64_t i = 0; i < skip; i++) (void)mt_next(&st);
    for (uint32_t f = 0; f < count; f++) {
        out[f] = mt_next(&st);
        for (int k = 1; k < 16; k++) (void)mt_next(&st);
    }
}

/* ======================================================================= */
/* intersection                                                            */
/* ======================================================================= */

typedef struct { pt_f3 origin; float t_min; pt_f3 direction; float t_max; } orc_ray;

static pt_f3 ld3(const pt_vec3 *v) { return pt_mk3(v->v[0], v->v[1], v->v[2]); }

/* ray_tracing.hlsl:46-86 */
static int intersection_bound(const orc_ray *ray, const pt_Bound *bound, float current_t, float *tout)
{
    float t_min, t_max, t_xmin, t_xmax, t_ymin, t_ymax, t_zmin, t_zmax;
    float x_a = 1.0f / ray->direction.x, y_a = 1.0f / ray->direction.y, z_a = 1.0f / ray->direction.z;
    float x_e = ray->origin.x, y_e = ray->origin.y, z_e = ray->origin.z;
    if (x_a >= 0) { t_xmin = (bound->min.v[0] - x_e) * x_a; t_xmax = (bound->max.v[0] - x_e) * x_a; }
    else          { t_xmin = (bound->max.v[0] - x_e) * x_a; t_xmax = (bound->min.v[0] - x_e) * x_a; }
    if (y_a >= 0) { t_ymin = (bound->min.v[1] - y_e) * y_a; t_ymax = (bound->max.v[1] - y_e) * y_a; }
    else          { t_ymin = (bound->max.v[1] - y_e) * y_a; t_ymax = (bound->min.v[1] - y_e) * y_a; }
    if (z_a >= 0) { t_zmin = (bound->min.v[2] - z_e) * z_a; t_zmax = (bound->max.v[2] - z_e) * z_a; }
    else          { t_zmin = (bound->max.v[2] - z_e) * z_a; t_zmax = (bound->min.v[2] - z_e) * z_a; }
    t_min = pt_max(t_xmin, pt_max(t_ymin, t_zmin));
    t_max = pt_min(t_xmax, pt_min(t_ymax, t_zmax));
    *tout = t_min;
    return (t_min <= t_max && t_min <= current_t);
}

/* ray_tracing.hlsl:167-189 */
static int calc_barycentrics3(const pt_PrimitiveRender *prims, int index, const orc_ray *ray, float *t, pt_f3 *bary)
{
    pt_f3 a = ld3(&prims[index].vertices[0].pos);
    pt_f3 b = ld3(&prims[index].vertices[1].pos);
    pt_f3 c = ld3(&prims[index].vertices[2].pos);
    pt_f3 e_1 = pt_sub3(b, a), e_2 = pt_sub3(c, a);
    pt_f3 q = pt_cross3(ray->direction, e_2);
    float a1 = pt_dot3(e_1, q);
    if (fabsf(a1) <= PT_EPSILON) { *t = -1; return 0; }
    pt_f3 oa = pt_sub3(ray->origin, a);
    pt_f3 s = pt_mk3(oa.x / a1, oa.y / a1, oa.z / a1);
    pt_f3 r = pt_cross3(s, e_1);
    *t = pt_dot3(e_2, r);
    bary->x = pt_dot3(s, q);
    bary->y = pt_dot3(r, ray->direction);
    bary->z = 1 - bary->x - bary->y;
    if (bary->x >= -PT_EPSILON && bary->y >= -PT_EPSILON && bary->z >= -PT_EPSILON)
        return *t >= PT_EPSILON;
    return 0;
}

/* ray_tracing.hlsl:191-243.  stack_size 15 is the reference; the early return
 * on overflow (":235-238") is kept for any size. */
static int intersect(const orc_scene *sc, const orc_ray *ray, int *primitive_index, pt_f3 *barycentrics,
                     float *current_t, orc_counters *cnt)
{
    int stack_mem[ORC_MAX_STACK + 1];
    int *stack = stack_mem + 1;                    /* stack[-1] is writable, as the overflow path writes it */
    const int SS = sc->stack_size;
    int stack_top = SS;
    stack[--stack_top] = 0;
    int intersected = 0;
    *current_t = ray->t_max;
    if (cnt) cnt->rays++;
    while (stack_top != SS) {
        int current = stack[stack_top++];
        float t;
        if (cnt) { cnt->node_visits++; cnt->box_tests++; }
        if (intersection_bound(ray, &sc->tree[current].bound, *current_t, &t)) {
            if (current >= sc->n) {
                pt_f3 tmp;
                if (cnt) cnt->tri_tests++;
                if (calc_barycentrics3(sc->prims, sc->tree[current].index, ray, &t, &tmp)) {
                    if (*current_t > t && t > ray->t_min) {
                        *barycentrics = tmp;
                        *current_t = t;
                        *primitive_index = sc->tree[current].index;
                        intersected = 1;
                    }
                }
            } else {
                int first = sc->tree[current].left, second = sc->tree[current].right;
                if (sc->near_first) {
                    /* NON-PARITY option: when both children pass their box test now, the one the ray
                     * enters first is visited first (ties: left, the reference's order) */
                    float tl, tr;
                    int hl = intersection_bound(ray, &sc->tree[first].bound, *current_t, &tl);
                    int hr = intersection_bound(ray, &sc->tree[second].bound, *current_t, &tr);
                    if (hl && hr && tr < tl) { int tmp = first; first = second; second = tmp; }
                }
                stack[--stack_top] = second;
                stack[--stack_top] = first;
                if (stack_top < 0) return intersected;
            }
        }
    }
    return intersected;
}

int orc_intersect_one(const orc_scene *s, const float origin[3], const float dir[3],
                      float t_min, float t_max, int *prim, float bary[3], float *t, orc_counters *cnt)
{
    orc_ray r;
    r.origin = pt_mk3(origin[0], origin[1], origin[2]);
    r.direction = pt_mk3(dir[0], dir[1], dir[2]);
    r.t_min = t_min; r.t_max = t_max;
    pt_f3 b = pt_mk3(0, 0, 0);
    *prim = -1;
    int hit = intersect(s, &r, prim, &b, t, cnt);
    bary[0] = b.x; bary[1] = b.y; bary[2] = b.z;
    return hit;
}

/* ======================================================================= */
/* shading                                                                 */
/* ======================================================================= */

/* mats[id] of the constant buffer.  An index outside the 16 entries reads zeros, as an out-of-range
 * constant-buffer read does in Direct3D 11 (the reference declares `Material mats[16]`, structs.hlsli). */
static const pt_Material *material_of(const orc_scene *sc, int id)
{
    static const pt_Material zero_material;
    return ((unsigned)id < (unsigned)PT_MAX_MATERIALS) ? &sc->cb->mats[id] : &zero_material;
}

/* ray_tracing.hlsl:249-346 */
static int get_bounce(const orc_scene *sc, uint32_t framecount, pt_f3 view_dir, pt_f3 normal, int id,
                      pt_f3 *multi, pt_f3 *direction, uint32_t *seed)
{
    *direction = pt_mk3(0, 0, 0);
    const pt_Material *mat = material_of(sc, id);
    float rand_mat = orc_rand_next(seed);
    float base = mat->tr + mat->ks.v[0] + mat->kd.v[0];
    if (mat->ke > 0) {
        *multi = pt_mul3(*multi, pt_mk3(mat->ke, mat->ke, mat->ke));
        return 0;
    }
    if (mat->tr > 0) {
        float cos = pt_dot3(normal, view_dir);
        float nc = 1.f;
        float nt = mat->ni;
        float nnt = nc / nt;
        if (cos > 0) nnt = nt / nc;
        float ddn = fabsf(cos);
        float cos2t = 1.f - nnt * nnt * (1.f - ddn * ddn);
        if (cos2t < 0.0f) {
            *direction = pt_reflect3(view_dir, normal);
            return 1;
        }
        if (cos < 0) {
            float k = cos * nnt + sqrtf(cos2t);
            *direction = pt_normalize3(pt_sub3(pt_scale3(view_dir, nnt), pt_scale3(normal, k)));
        } else {
            float k = -cos * nnt + sqrtf(cos2t);
            *direction = pt_normalize3(pt_add3(pt_scale3(view_dir, nnt), pt_scale3(normal, k)));
        }
        float a = nt - nc;
        float b = nt + nc;
        float R0 = a * a / (b * b);
        float c = 1 - ddn;
        float Re = R0 + (1 - R0) * c * c * c * c * c;
        float Tr = 1 - Re;
        float P = .25f + .5f * Re;
        float RP = Re / P;
        float TP = Tr / (1.f - P);
        if (orc_rand_next(seed) < P) {
            *multi = pt_scale3(*multi, RP);
            *direction = pt_reflect3(view_dir, normal);
            return 1;
        } else {
            *multi = pt_scale3(*multi, TP);
            return 1;
        }
    } else if (mat->ns > 1 && rand_mat < mat->ks.v[0] / base) {
        *direction = pt_reflect3(view_dir, normal);
        float r1 = 2 * PT_MY_PI * ((framecount & 1) / 2.0f + orc_rand_next(seed) / 2);
        float height = pt_powf((((framecount >> 1) & 3) / 4.0f + orc_rand_next(seed) / 4), 1 / (mat->ns));
        float r2s = sqrtf(1 - height * height);
        pt_f3 w = *direction;
        pt_f3 u = pt_cross3(w, view_dir);
        u = pt_normalize3(u);
        pt_f3 v = pt_cross3(w, u);
        pt_f3 d = pt_add3(pt_add3(pt_scale3(pt_scale3(u, pt_cosf(r1)), r2s),
                                  pt_scale3(pt_scale3(v, pt_sinf(r1)), r2s)),
                          pt_scale3(w, height));
        *direction = pt_normalize3(d);
        *multi = pt_mul3(*multi, pt_mk3(mat->ks.v[0], mat->ks.v[1], mat->ks.v[2]));
    } else {
        *multi = pt_mul3(*multi, pt_mk3(mat->kd.v[0], mat->kd.v[1], mat->kd.v[2]));
        float r1 = 2 * PT_MY_PI * (((framecount >> 1) & 3) / 4.0f + orc_rand_next(seed) / 4.0f);
        float r2 = (((framecount >> 3) & 3) / 4.0f + orc_rand_next(seed) / 4.0f);
        float r2s = sqrtf(r2);
        pt_f3 w = normal;
        pt_f3 u;
        if (fabsf(w.x) > 0.1f) u = pt_cross3(w, pt_mk3(0, 1, 0));
        else u = pt_cross3(w, pt_mk3(1, 0, 0));
        u = pt_normalize3(u);
        pt_f3 v = pt_cross3(w, u);
        pt_f3 d = pt_add3(pt_add3(pt_scale3(pt_scale3(u, pt_cosf(r1)), r2s),
                                  pt_scale3(pt_scale3(v, pt_sinf(r1)), r2s)),
                          pt_scale3(w, sqrtf(1 - r2)));
        *direction = pt_normalize3(d);
    }
    return 1;
}

#define ORC_MAX_ITR 5     /* ray_tracing.hlsl:245-247 */

/* ray_tracing.hlsl:348-397 */
static pt_f3 second_tracing(const orc_scene *sc, uint32_t framecount,
                            const uint8_t *diffuse_px, const float *normal_px, const float *world_px,
                            uint32_t seed, orc_counters *cnt)
{
    const pt_f3 zero = pt_mk3(0, 0, 0);
    pt_f3 diffuse = pt_mk3(diffuse_px[0] / 255.0f, diffuse_px[1] / 255.0f, diffuse_px[2] / 255.0f);
    pt_f3 normal = pt_mk3(normal_px[0], normal_px[1], normal_px[2]);
    pt_f3 hitPos = pt_mk3(world_px[0], world_px[1], world_px[2]);
    int itr = 1;
    int primitive_index = 0;
    pt_f3 barycentrics = zero;
    float t;
    orc_ray this_ray;
    this_ray.t_min = PT_EPSILON;
    this_ray.t_max = 10000;
    pt_f3 multi = diffuse;
    int matid = (int)pt_f2u(normal_px[3]);               /* asint(normal.w) */
    if (matid <= 0) return zero;                         /* 0 = background; negative words are treated alike */
    matid -= 1;
    if (material_of(sc, matid)->ns > 1) multi = pt_mk3(1, 1, 1);
    pt_f3 direction;
    pt_f3 g_pos = ld3(&sc->cb->pos);
    if (!get_bounce(sc, framecount, pt_normalize3(pt_sub3(hitPos, g_pos)), pt_normalize3(normal), matid,
                    &multi, &direction, &seed))
        return multi;
    this_ray.origin = hitPos;
    this_ray.direction = direction;
    while (intersect(sc, &this_ray, &primitive_index, &barycentrics, &t, cnt)) {
        this_ray.origin = pt_add3(this_ray.origin, pt_scale3(this_ray.direction, t));
        itr++;
        matid = sc->prims[primitive_index].matid;
        if (itr > ORC_MAX_ITR) {
            float p = pt_max(pt_max(multi.x, multi.y), multi.z);
            if (p > orc_rand_next(&seed)) multi = pt_scale3(multi, 1 / p);
            else return zero;
        }
        if (itr >= (sc->max_segments > 0 ? sc->max_segments + 1 : ORC_MAX_ITR * 3)) return zero;
        pt_f3 A = ld3(&sc->prims[primitive_index].vertices[0].normal);
        pt_f3 B = ld3(&sc->prims[primitive_index].vertices[1].normal);
        pt_f3 C = ld3(&sc->prims[primitive_index].vertices[2].normal);
        pt_f3 new_normal = pt_add3(pt_add3(pt_scale3(A, barycentrics.z), pt_scale3(B, barycentrics.x)),
                                   pt_scale3(C, barycentrics.y));
        if (!get_bounce(sc, framecount, this_ray.direction, new_normal, matid, &multi, &direction, &seed))
            return multi;
        this_ray.direction = direction;
    }
    return zero;
}

/* ray_tracing.hlsl:410: uint(launchIndex.x + launchIndex.y * viewportDims.x) in float */
static uint32_t pixel_seed(int x, int y, int W, uint32_t frame_seed)
{
    float f = (float)(uint32_t)x + (float)(uint32_t)y * (float)W;
    return orc_rand_init((uint32_t)f, frame_seed);
}

/* float -> UNORM8 store conversion of the display target: NaN -> 0, clamp to
 * [0,1], scale by 255, add 0.5, truncate (Direct3D 11 functional spec, FLOAT ->
 * UNORM; not defined by any file of the reference: parity unpinned). */
static uint8_t to_unorm8(float c)
{
    if (!(c == c)) return 0;
    if (c < 0.0f) c = 0.0f;
    if (c > 1.0f) c = 1.0f;
    return (uint8_t)(int)(c * 255.0f + 0.5f);
}

/* ray_tracing.hlsl:399-420 */
void orc_trace_frame(const orc_scene *s, int W, int H,
                     const uint8_t *diffuse_rgba8, const float *normal_matid, const float *worldpos,
                     uint32_t framecount, uint32_t seed,
                     const float *old_rgba, float *out_rgba, uint8_t *out_unorm,
                     float *radiance, orc_counters *cnt, int y0, int y1)
{
    (void)H;
    const float inv_gamma = 1 / 2.2f;
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < W; x++) {
            size_t p = (size_t)y * (size_t)W + (size_t)x;
            uint32_t ps = pixel_seed(x, y, W, seed);
            if (cnt) cnt->paths++;
            pt_f3 c = second_tracing(s, framecount, &diffuse_rgba8[4 * p], &normal_matid[4 * p], &worldpos[4 * p], ps, cnt);
            if (radiance) { radiance[3 * p] = c.x; radiance[3 * p + 1] = c.y; radiance[3 * p + 2] = c.z; }
            float this_color[4] = { c.x, c.y, c.z, 1.0f };
            float old_factor = 1.0f * (float)framecount / (float)(framecount + 1);
            float new_factor = 1.0f / (float)(framecount + 1);
            for (int k = 0; k < 4; k++) {
                float acc = pt_powf(pt_powf(old_rgba[4 * p + k], 2.2f) * old_factor + this_color[k] * new_factor, inv_gamma);
                out_rgba[4 * p + k] = acc;
                if (out_unorm) out_unorm[4 * p + k] = to_unorm8(acc);
            }
        }
}

void orc_trace_frame_linear(const orc_scene *s, int W, int H,
                            const uint8_t *diffuse_rgba8, const float *normal_matid, const float *worldpos,
                            uint32_t framecount, uint32_t seed,
                            float *acc_rgba, orc_counters *cnt, int y0, int y1)
{
    (void)H;
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < W; x++) {
            size_t p = (size_t)y * (size_t)W + (size_t)x;
            uint32_t ps = pixel_seed(x, y, W, seed);
            if (cnt) cnt->paths++;
            pt_f3 c = second_tracing(s, framecount, &diffuse_rgba8[4 * p], &normal_matid[4 * p], &worldpos[4 * p], ps, cnt);
            acc_rgba[4 * p + 0] += c.x;
            acc_rgba[4 * p + 1] += c.y;
            acc_rgba[4 * p + 2] += c.z;
            acc_rgba[4 * p + 3] += 1.0f;
        }
}

void orc_resolve(int W, int H, const float *acc_rgba, uint32_t frames, float *out_rgba, uint8_t *out_unorm)
{
    const float inv_gamma = 1 / 2.2f;
    const float inv_n = 1.0f / (float)frames;
    for (size_t p = 0; p < (size_t)W * (size_t)H; p++) {
        for (int k = 0; k < 3; k++) {
            float v = pt_powf(acc_rgba[4 * p + k] * inv_n, inv_gamma);
            out_rgba[4 * p + k] = v;
            if (out_unorm) out_unorm[4 * p + k] = to_unorm8(v);
        }
        out_rgba[4 * p + 3] = 1.0f;
        if (out_unorm) out_unorm[4 * p + 3] = 255;
    }
}

/* ======================================================================= */
/* G-buffer by ray casting                                                 */
/* ======================================================================= */

/* Bytes the colour render target holds for a material colour.  The reference
 * draws Kd into the sRGB back buffer and copies the raw bytes into a UNORM
 * texture (CS.cpp:302,357-359,373; fixed_pipeline.hlsl:62), so the shader sees
 * the sRGB-ENCODED value; srgb_quirk = 0 stores plain UNORM8 instead.  The
 * conversion itself is Direct3D's, not the reference's: parity unpinned. */
void orc_albedo_bytes(const float kd[3], int srgb_quirk, uint8_t out[3])
{
    for (int k = 0; k < 3; k++) {
        double c = kd[k];
        if (!(c == c) || c < 0.0) c = 0.0;
        if (c > 1.0) c = 1.0;
        if (srgb_quirk) c = (c <= 0.0031308) ? 12.92 * c : 1.055 * pow(c, 1.0 / 2.4) - 0.055;
        out[k] = (uint8_t)(int)(c * 255.0 + 0.5);
    }
}

/* Pixel-centre rays through the left-handed look-at / perspective camera of
 * CS.cpp:200-201,318-329; closest hit with view depth in [znear, zfar], no
 * culling (CS.cpp:175); outputs as fixed_pipeline.hlsl:59-67 writes them:
 * material colour, normalised interpolated normal + asfloat(matid+1),
 * interpolated world position.  The direction is scaled so that the ray
 * parameter equals view-space depth. */
void orc_primary(const orc_scene *s, const orc_camera *cam, int W, int H,
                 const uint8_t *albedo8,
                 uint8_t *diffuse_rgba8, float *normal_matid, float *worldpos, int y0, int y1)
{
    pt_f3 eye = pt_mk3(cam->eye[0], cam->eye[1], cam->eye[2]);
    pt_f3 at = pt_mk3(cam->at[0], cam->at[1], cam->at[2]);
    pt_f3 up = pt_mk3(cam->up[0], cam->up[1], cam->up[2]);
    pt_f3 zaxis = pt_normalize3(pt_sub3(at, eye));
    pt_f3 xaxis = pt_normalize3(pt_cross3(up, zaxis));
    pt_f3 yaxis = pt_cross3(zaxis, xaxis);
    float tan_half = tanf(cam->fovy * 0.5f);
    float aspect = (float)W / (float)H;
    float sx = tan_half * aspect, sy = tan_half;
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < W; x++) {
            size_t p = (size_t)y * (size_t)W + (size_t)x;
            float ndc_x = ((float)x + 0.5f) / (float)W * 2.0f - 1.0f;
            float ndc_y = 1.0f - ((float)y + 0.5f) / (float)H * 2.0f;
            float vx = ndc_x * sx, vy = ndc_y * sy;
            orc_ray r;
            r.origin = eye;
            r.direction = pt_add3(pt_add3(pt_scale3(xaxis, vx), pt_scale3(yaxis, vy)), zaxis);
            r.t_min = cam->znear;
            r.t_max = cam->zfar;
            int prim = 0; pt_f3 bary = pt_mk3(0, 0, 0); float t;
            memset(&diffuse_rgba8[4 * p], 0, 4);
            memset(&normal_matid[4 * p], 0, 16);
            memset(&worldpos[4 * p], 0, 16);
            if (!intersect(s, &r, &prim, &bary, &t, NULL)) continue;
            const pt_PrimitiveRender *pr = &s->prims[prim];
            pt_f3 nA = ld3(&pr->vertices[0].normal), nB = ld3(&pr->vertices[1].normal), nC = ld3(&pr->vertices[2].normal);
            pt_f3 pA = ld3(&pr->vertices[0].pos), pB = ld3(&pr->vertices[1].pos), pC = ld3(&pr->vertices[2].pos);
            pt_f3 nn = pt_normalize3(pt_add3(pt_add3(pt_scale3(nA, bary.z), pt_scale3(nB, bary.x)), pt_scale3(nC, bary.y)));
            pt_f3 pp = pt_add3(pt_add3(pt_scale3(pA, bary.z), pt_scale3(pB, bary.x)), pt_scale3(pC, bary.y));
            int matid = pr->matid;
            if ((unsigned)matid < (unsigned)PT_MAX_MATERIALS) {      /* a 16-entry table; other ids: black */
                diffuse_rgba8[4 * p + 0] = albedo8[4 * matid + 0];
                diffuse_rgba8[4 * p + 1] = albedo8[4 * matid + 1];
                diffuse_rgba8[4 * p + 2] = albedo8[4 * matid + 2];
            }
            diffuse_rgba8[4 * p + 3] = 255;
            normal_matid[4 * p + 0] = nn.x; normal_matid[4 * p + 1] = nn.y; normal_matid[4 * p + 2] = nn.z;
            normal_matid[4 * p + 3] = pt_u2f((uint32_t)(matid + 1));
            worldpos[4 * p + 0] = pp.x; worldpos[4 * p + 1] = pp.y; worldpos[4 * p + 2] = pp.z;
            worldpos[4 * p + 3] = 1.0f;
        }
}

float orc_sinf(float x) { return pt_sinf(x); }
float orc_cosf(float x) { return pt_cosf(x); }
float orc_exp2f(float x) { return pt_exp2f(x); }
float orc_log2f(float x) { return pt_log2f(x); }
float orc_powf(float x, float y) { return pt_powf(x, y); }
