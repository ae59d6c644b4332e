// Path-trace kernels on gfx950: BVH traversal, triangle tests, BSDF sampling,
// progressive accumulation, and the ray-cast G-buffer pass.
//
// Follows (reference): ray_tracing.hlsl:46-86 (box test), :167-189 (triangle
// test), :191-243 (traversal), :249-346 (getBounce), :348-397 (second_tracing),
// :399-420 (CSMain / accumulate), helpers.hlsli:5-32 (RNG); the G-buffer pass
// stands in for fixed_pipeline.hlsl:49-67 + CS.cpp:309-393.
//
// Results are defined by strict IEEE evaluation of those expressions with the
// elementary functions of include/ptamd_math.h: compile with -ffp-contract=off
// and no fast-math.  Work is VALU / latency bound on L2-resident trees; there
// is no dense contraction here, so no matrix-core use.
//
// Execution shape.  Every path writes its radiance sample into a [pixel][frame]
// staging buffer and k_accumulate folds the frames of a launch into the images
// in frame order, so which lane traces which path never enters a result.
//   k_trace_persistent  one persistent workgroup per CU, lanes take (pixel, frame
//                chunk) items from a global queue, most expensive pixels first;
//                the node records are held in LDS when the tree fits, else read
//                from global memory.
//   k_wf_*       the same paths bounce by bounce over queues (optional schedule).
// All of them keep the current node in a register and only the pending right
// children on a per-lane LDS stack.
#include "pt_internal.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {

constexpr int TRACE_THREADS = 256;
constexpr int POP = (int)0x80000000;

struct SceneView {
    const WideNode *__restrict__ wide;
    const float4 *__restrict__ pair;    // 8 float4 per internal node: the records of its two children (k_pair_records), or null
    const TriIsect *__restrict__ tri;
    const TriShade *__restrict__ shade;
    const pt_CB_Radix *__restrict__ cb;
    const float *__restrict__ root;     // the root's own box: min.xyz, max.xyz
    int n;
};

struct Ray {
    pt_f3 o; float t_min;
    pt_f3 d; float t_max;
};

// wave64 vote straight from the comparison's lane mask (the int-predicate form first turns
// the mask into 0/1 per lane and compares again: two vector instructions per vote)
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// Staging of one launch, layout [pixel][frame of the launch], one 16-byte slot per path: before the path starts its seed
// (in .w), afterwards its radiance sample.  Seed and sample share the slot on purpose: measured against 12-byte samples with
// the seeds in an array of their own (which moves fewer bytes in the seed and accumulate kernels), the shared slot left the
// trace kernel's own memory traffic smaller (aligned 16-byte stores: one 32-byte sector each; 12-byte stores straddle sectors)
// and, on trees far larger than the caches, 17 % more rays per second: the seed read brings the address translation of the
// slot's page in, so the sample store that follows completes at once instead of holding up the next node fetch's wait.
struct Staging {
    float4 *__restrict__ slot;
};
// The same for the kernel variants that may be handed their seeds in an array of their own, [pixel][frame], computed ahead of
// the launch (trees in LDS).  A type of its own, so that the other variants keep their argument list and their code: the
// variant with child lines runs at the register limit, and its allocation turns with anything that is added to it.
struct StagingAhead {
    float4 *__restrict__ slot;
    const uint32_t *__restrict__ seeds;     // or null: the seed is in the slot's .w
};
template <bool LDSN> struct StagingOf {
    typedef Staging type;
    static const Staging &pick(const Staging &a, const StagingAhead &) { return a; }
};
template <> struct StagingOf<true> {
    typedef StagingAhead type;
    static const StagingAhead &pick(const Staging &, const StagingAhead &b) { return b; }
};
template <typename ST>
__device__ __forceinline__ void put_sample(const ST &st, size_t path, float r, float g, float b)
{
    st.slot[path] = make_float4(r, g, b, 1.0f);
}
__device__ __forceinline__ uint32_t path_seed(const Staging &st, size_t path) { return __float_as_uint(st.slot[path].w); }
__device__ __forceinline__ uint32_t path_seed(const StagingAhead &st, size_t path)
{
    return st.seeds ? st.seeds[path] : __float_as_uint(st.slot[path].w);
}

struct LaneCounters {
    unsigned paths = 0, rays = 0, node_visits = 0, box_tests = 0, tri_tests = 0;
};

// helpers.hlsli:5-19
__device__ __forceinline__ uint32_t rand_init(uint32_t val0, uint32_t val1)
{
    uint32_t v0 = val0, v1 = val1, s0 = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}
// helpers.hlsli:26-32
__device__ __forceinline__ float rand_next(uint32_t &s)
{
    s = 1664525u * s + 1013904223u;
    return (float)(s & 0x00FFFFFFu) / (float)0x01000000;
}

// ray_tracing.hlsl:46-86 with the reciprocal direction hoisted out (it is a pure
// function of the ray).  Returns t_min <= t_max; the caller compares entry
// distance with the current hit distance.  Per axis the (near, far) plane pair
// is selected by the sign test on 1/d exactly as the shader does.  Plain scalar
// f32 arithmetic on purpose: the packed forms (v_pk_add_f32 / v_pk_mul_f32, which
// the SLP vectoriser also produces on its own; the library is built without it)
// made the traversal loop 6.5 % slower.
__device__ __forceinline__ bool slab(const float *mn, const float *mx, const pt_f3 o, const pt_f3 inv, float &t_entry)
{
    float nx, fx, ny, fy, nz, fz;
    if (inv.x >= 0) { nx = mn[0]; fx = mx[0]; } else { nx = mx[0]; fx = mn[0]; }
    if (inv.y >= 0) { ny = mn[1]; fy = mx[1]; } else { ny = mx[1]; fy = mn[1]; }
    if (inv.z >= 0) { nz = mn[2]; fz = mx[2]; } else { nz = mx[2]; fz = mn[2]; }
    float t_min = pt_max((nx - o.x) * inv.x, pt_max((ny - o.y) * inv.y, (nz - o.z) * inv.z));
    float t_max = pt_min((fx - o.x) * inv.x, pt_min((fy - o.y) * inv.y, (fz - o.z) * inv.z));
    t_entry = t_min;
    return t_min <= t_max;
}

// ray_tracing.hlsl:191-243.  Visits nodes in the reference's order (left subtree
// before right) and applies its tests at the same moments: a child's box is
// tested when its parent is expanded, and the comparison of the entry distance
// with the current hit distance is repeated when the child is taken from the
// stack (the hit distance may have shrunk meanwhile).  The result equals the
// reference loop's whenever that loop does not overflow its 15-entry stack.
template <bool COUNT>
__device__ __forceinline__ bool traverse(const SceneView &sc, const Ray &ray, uint2 *__restrict__ stack, int stride,
                                         int &primitive_index, pt_f3 &bary, float &current_t, LaneCounters &cnt)
{
    const pt_f3 inv = pt_mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
    bool intersected = false;
    current_t = ray.t_max;
    cnt.rays++;
    float t_entry;
    if (COUNT) { cnt.node_visits++; cnt.box_tests++; }
    {
        const float rmn[3] = {sc.root[0], sc.root[1], sc.root[2]}, rmx[3] = {sc.root[3], sc.root[4], sc.root[5]};
        if (!(slab(rmn, rmx, ray.o, inv, t_entry) && t_entry <= current_t)) return false;
    }
    int cur = 0;
    int sp = 0;
    for (;;) {
        if (cur >= 0) {
            const float4 *w = reinterpret_cast<const float4 *>(&sc.wide[cur]);
            const float4 w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
            const float lmn[3] = {w0.x, w0.y, w0.z}, lmx[3] = {w0.w, w1.x, w1.y};
            const float rmn[3] = {w1.z, w1.w, w2.x}, rmx[3] = {w2.y, w2.z, w2.w};
            const int left = __float_as_int(w3.x), right = __float_as_int(w3.y);
            float tl, tr;
            const bool hl = slab(lmn, lmx, ray.o, inv, tl) && tl <= current_t;
            const bool hr = slab(rmn, rmx, ray.o, inv, tr) && tr <= current_t;
            if (COUNT) { cnt.node_visits += 2; cnt.box_tests += 2; }
            if (hl) {
                if (hr) { stack[sp * stride] = make_uint2((unsigned)right, __float_as_uint(tr)); sp++; }
                cur = left;
            } else {
                cur = hr ? right : POP;
            }
        } else if (cur != POP) {
            // leaves are ~key; the record in leaf order carries the primitive id in its spare word
            const float4 *tp = reinterpret_cast<const float4 *>(&sc.tri[~cur]);
            const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
            const int prim = __float_as_int(t0.w);
            if (COUNT) cnt.tri_tests++;
            // ray_tracing.hlsl:167-189
            const pt_f3 a = pt_mk3(t0.x, t0.y, t0.z), e_1 = pt_mk3(t1.x, t1.y, t1.z), e_2 = pt_mk3(t2.x, t2.y, t2.z);
            const pt_f3 q = pt_cross3(ray.d, e_2);
            const float a1 = pt_dot3(e_1, q);
            if (!(fabsf(a1) <= PT_EPSILON)) {
                const pt_f3 oa = pt_sub3(ray.o, a);
                const pt_f3 s = pt_mk3(oa.x / a1, oa.y / a1, oa.z / a1);
                const pt_f3 r = pt_cross3(s, e_1);
                const float t = pt_dot3(e_2, r);
                const float bx = pt_dot3(s, q);
                const float by = pt_dot3(r, ray.d);
                const float bz = 1 - bx - by;
                if (bx >= -PT_EPSILON && by >= -PT_EPSILON && bz >= -PT_EPSILON && t >= PT_EPSILON) {
                    if (current_t > t && t > ray.t_min) {
                        bary = pt_mk3(bx, by, bz);
                        current_t = t;
                        primitive_index = prim;
                        intersected = true;
                    }
                }
            }
            cur = POP;
        }
        if (cur == POP) {
            // take pending children until one still passes the distance test
            for (;;) {
                if (sp == 0) return intersected;
                sp--;
                const uint2 e = stack[sp * stride];
                if (__uint_as_float(e.y) <= current_t) { cur = (int)e.x; break; }
            }
        }
    }
}

// A node's own record (the boxes of its two children) from the records of those children: a child's box is the union
// of ITS children's boxes, taken with the refit's operations in the refit's operand order (left, right), so the bits
// are the ones the build stored.  a = the left child's record, b = the right child's (a leaf child: its box twice).
__device__ __forceinline__ void record_from_children(const float4 a0, const float4 a1, const float4 a2,
                                                     const float4 b0, const float4 b1, const float4 b2,
                                                     float4 &w0, float4 &w1, float4 &w2)
{
    w0 = make_float4(fminf(a0.x, a1.z), fminf(a0.y, a1.w), fminf(a0.z, a2.x), fmaxf(a0.w, a2.y));
    w1 = make_float4(fmaxf(a1.x, a2.z), fmaxf(a1.y, a2.w), fminf(b0.x, b1.z), fminf(b0.y, b1.w));
    w2 = make_float4(fminf(b0.z, b2.x), fmaxf(b0.w, b2.y), fmaxf(b1.x, b2.z), fmaxf(b1.y, b2.w));
}

// Material tables have PT_MAX_MATERIALS entries plus one all-zero entry: an id outside 0..15 (a
// primitive or G-buffer texel the caller filled with something else) reads the zero material, as an
// out-of-range constant-buffer read does in Direct3D 11, never memory beyond the table.
constexpr int MAT_SLOTS = PT_MAX_MATERIALS + 1;
__device__ __forceinline__ int mat_slot(int id) { return (int)min((unsigned)id, (unsigned)PT_MAX_MATERIALS); }

// The four quotients of getBounce that depend on the material alone (ray_tracing.hlsl:263 nc/nt,
// :284 R0, :303 ks.x/(tr+ks.x+kd.x), :307 1/ns), evaluated with the shader's own float expressions.
__device__ __forceinline__ float4 material_quotients(const float4 m0, const float4 m1, const float4 m2)
{
    const float ns = m0.y, tr = m0.z, ni = m0.w;
    const float base = tr + m2.x + m1.x;
    const float a = ni - 1.f, b = ni + 1.f;
    return make_float4(1 / ns, m2.x / base, 1.f / ni, a * a / (b * b));
}

// ray_tracing.hlsl:249-346.  `mats` = 3 float4 per material (pt_Material); `matq` = one
// material_quotients record per material, or null to evaluate them here.
__device__ __forceinline__ bool get_bounce(const float4 *mats, const float4 *matq, uint32_t framecount,
                                           const pt_f3 view_dir, const pt_f3 normal, int id,
                                           pt_f3 &multi, pt_f3 &direction, uint32_t &seed)
{
    direction = pt_mk3(0, 0, 0);
    id = mat_slot(id);
    const float4 *mp = mats + 3 * id;
    const float4 m0 = mp[0], m1 = mp[1], m2 = mp[2];
    const float ke = m0.x, ns = m0.y, tr = m0.z, ni = m0.w;
    const float rand_mat = rand_next(seed);
    if (ke > 0) {
        multi = pt_mul3(multi, pt_mk3(ke, ke, ke));
        return false;
    }
    if (tr > 0) {
        const float4 mq = matq ? matq[id] : material_quotients(m0, m1, m2);
        const float cosv = pt_dot3(normal, view_dir);
        const float nt = ni;
        float nnt = mq.z;                    // nc / nt, nc = 1
        if (cosv > 0) nnt = nt;              // nt / nc
        const float ddn = fabsf(cosv);
        const float cos2t = 1.f - nnt * nnt * (1.f - ddn * ddn);
        if (cos2t < 0.0f) {
            direction = pt_reflect3(view_dir, normal);
            return true;
        }
        // view*nnt - normal*k (entering) or view*nnt + normal*k (leaving), ray_tracing.hlsl:272-281:
        // a - b is a + (-b) bit for bit, so one normalize serves both
        const float k = (cosv < 0 ? cosv : -cosv) * nnt + sqrtf(cos2t);
        pt_f3 nk = pt_scale3(normal, k);
        if (cosv < 0) nk = pt_mk3(-nk.x, -nk.y, -nk.z);
        direction = pt_normalize3(pt_add3(pt_scale3(view_dir, nnt), nk));
        const float R0 = mq.w;               // (nt - nc)^2 / (nt + nc)^2
        const float c = 1 - ddn;
        const float Re = R0 + (1 - R0) * c * c * c * c * c;
        const float Tr = 1 - Re;
        const float P = .25f + .5f * Re;
        const float RP = Re / P;
        const float TP = Tr / (1.f - P);
        if (rand_next(seed) < P) {
            multi = pt_scale3(multi, RP);
            direction = pt_reflect3(view_dir, normal);
        } else {
            multi = pt_scale3(multi, TP);
        }
        return true;
    }
    const float4 mq = matq ? matq[id] : material_quotients(m0, m1, m2);
    // Phong lobe around the mirror direction (ray_tracing.hlsl:303-321) or cosine lobe around the
    // normal (:322-341).  Both build a frame (u, v, w) and combine it with cos / sin of r1; the
    // branches only choose the operands, the shared tail evaluates the same expressions once for
    // whichever lanes took either branch.
    pt_f3 w, u_raw;
    float r1, r2s, height;
    if (ns > 1 && rand_mat < mq.y) {
        w = pt_reflect3(view_dir, normal);
        r1 = 2 * PT_MY_PI * ((framecount & 1) / 2.0f + rand_next(seed) / 2);
        height = pt_powf((((framecount >> 1) & 3) / 4.0f + rand_next(seed) / 4), mq.x);
        r2s = sqrtf(1 - height * height);
        u_raw = pt_cross3(w, view_dir);
        multi = pt_mul3(multi, pt_mk3(m2.x, m2.y, m2.z));
    } else {
        multi = pt_mul3(multi, pt_mk3(m1.x, m1.y, m1.z));
        r1 = 2 * PT_MY_PI * (((framecount >> 1) & 3) / 4.0f + rand_next(seed) / 4.0f);
        const float r2 = (((framecount >> 3) & 3) / 4.0f + rand_next(seed) / 4.0f);
        r2s = sqrtf(r2);
        w = normal;
        const bool y_axis = fabsf(w.x) > 0.1f;
        u_raw = pt_cross3(w, pt_mk3(y_axis ? 0.0f : 1.0f, y_axis ? 1.0f : 0.0f, 0.0f));
        height = sqrtf(1 - r2);
    }
    const pt_f3 u = pt_normalize3(u_raw);
    const pt_f3 v = pt_cross3(w, u);
    const pt_f3 d = pt_add3(pt_add3(pt_scale3(pt_scale3(u, pt_cosf(r1)), r2s),
                                    pt_scale3(pt_scale3(v, pt_sinf(r1)), r2s)),
                            pt_scale3(w, height));
    direction = pt_normalize3(d);
    return true;
}

// FLOAT -> UNORM8 store of the display target: NaN -> 0, clamp, *255, +0.5, truncate
__device__ __forceinline__ unsigned to_unorm8(float c)
{
    if (!(c == c)) return 0u;
    c = (c < 0.0f) ? 0.0f : c;
    c = (c > 1.0f) ? 1.0f : c;
    return (unsigned)(int)(c * 255.0f + 0.5f);
}

__device__ __forceinline__ void flush_counters(const LaneCounters &c, DevCounters *__restrict__ out, bool detailed)
{
    unsigned p = c.paths, r = c.rays;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { p += __shfl_xor(p, o, 64); r += __shfl_xor(r, o, 64); }
    unsigned long long nv = c.node_visits, bt = c.box_tests, tt = c.tri_tests;
    if (detailed) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            nv += __shfl_xor(nv, o, 64); bt += __shfl_xor(bt, o, 64); tt += __shfl_xor(tt, o, 64);
        }
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out->paths, (unsigned long long)p);
        atomicAdd(&out->rays, (unsigned long long)r);
        if (detailed) {
            atomicAdd(&out->node_visits, nv);
            atomicAdd(&out->box_tests, bt);
            atomicAdd(&out->tri_tests, tt);
        }
    }
}

// lane -> pixel: a 256-thread workgroup covers a 16x16 tile, each wave an 8x8 tile
__device__ __forceinline__ void lane_pixel(int &x, int &y)
{
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    x = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    y = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
}

constexpr int MAX_ITR = 5;   // ray_tracing.hlsl:245-247

// ---------------------------------------------------------------------------
// ray_tracing.hlsl:348-420 as a wave-level state machine.
//
// A lane is either TRAVERSING (it owns a ray and a current node) or IDLE (its
// ray is finished: it needs shading, a new ray, or the next path).  Paths of a
// wave's 64 lanes have very different lengths, and a traversal step is one of
// three different things (expand an internal node, test a triangle, pop).  Run
// naively, each lane drags the whole wave through its own control flow.
// Instead the wave alternates between
//   * a refill phase, entered when at least REFILL_T lanes are idle (or nobody
//     traverses): idle lanes shade their hit, continue or end their path, start
//     the next frame's path, and set up the next ray;
//   * traversal iterations in which every traversing lane does the same thing:
//     expand its internal node; triangle tests are postponed until at least
//     LEAF_T lanes wait at a leaf (or no lane has an internal node left); one
//     pop attempt per iteration.
// Per lane the order of node visits, tests and random draws is exactly that of
// the sequential loop, so results do not change.
// ---------------------------------------------------------------------------
// fine_frames: the last frames of a launch that k_trace_dual hands out in items of fine_chunk frames instead of `chunk`
// (0: none), so that the launch ends on short items
struct Tuning { int refill_t, leaf_t; int itr_limit; int swap_t, shade_t; int fine_frames, fine_chunk; };
// The upper levels of a tree that does not fit in LDS as a whole (index-keyed trees: the node at level k, block b has
// the compact index off[k] + b): levels k0 .. top are read from LDS, the deeper ones from global memory.
// k0 > top: nothing in LDS.
struct TopLevels { int k0, nodes; int off[33]; };    // itr_limit: a path ends when itr reaches it (reference: MAX_ITR * 3)

// ---------------------------------------------------------------------------
// Path seeds of one launch: seed(pixel, frame) = rand_init(pixel id, g_random[0] of the frame)
// (ray_tracing.hlsl:410, helpers.hlsli:5-19), written into the w component of the path's staging slot.
// The 16-round hash costs ~290 instructions; inside the persistent kernel a wave would run it in
// nearly every refill round for the few lanes that start a path, here every lane does useful work.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_path_seeds(int W, unsigned npix, uint32_t count, const uint32_t *__restrict__ seeds,
                                                    Staging samples, int frame_lanes_log2, uint32_t *__restrict__ array)
{
    // A workgroup covers 256 >> frame_lanes_log2 pixels; the lanes of a pixel stride over its frames, so a
    // pixel's slots are written side by side and no index needs a division by `count`.
    const unsigned fl0 = threadIdx.x & ((1u << frame_lanes_log2) - 1u), frame_lanes = 1u << frame_lanes_log2;
    const unsigned px_per_block = 256u >> frame_lanes_log2;
    for (unsigned px = blockIdx.x * px_per_block + (threadIdx.x >> frame_lanes_log2); px < npix; px += gridDim.x * px_per_block) {
        const unsigned y = px / (unsigned)W, x = px - y * (unsigned)W;
        // ray_tracing.hlsl:410: uint(launchIndex.x + launchIndex.y * viewportDims.x), in float
        const uint32_t pixel_id = (uint32_t)((float)x + (float)y * (float)W);
        for (unsigned fl = fl0; fl < count; fl += frame_lanes) {
            const uint32_t seed = rand_init(pixel_id, seeds[fl]);
            if (array) array[(size_t)px * count + fl] = seed;
            else samples.slot[(size_t)px * count + fl] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(seed));
        }
    }
}

// ---------------------------------------------------------------------------
// k_trace_persistent: that state machine as one persistent workgroup per CU.
//
// LDSN = true, for scenes whose node records fit (48 bytes per internal node: the
// two child boxes): the tree is held in LDS.  Every node visit is a per-lane
// random gather; through the vector L1 that is one tag lookup per lane per 16
// bytes, and the L1 tag pipeline (about one access per clock per CU) becomes the
// limit long before the VALU does.  LDS serves sixteen 16-byte lane reads per
// clock.  LDSN = false: the same records are read from global memory.
//
// Child ids are not stored.  The tree over keys 0..n is the binary trie of the
// key indices, so a node is named by (first key s, level k): its left child is
// (s, k-1) with record id s + 2^(k-1) - 1 (a left child's id is its last key),
// its right child starts at s + 2^(k-1), ends at e = min(s + 2^k - 1, n), has
// level bit_length(start ^ e) and record id = its first key; level 0 is a leaf
// whose key indexes the triangle records in sorted order.  Pending right
// children need no ids either: a per-lane bit mask records at which ancestor
// levels one is pending, and the ancestor's first key is the current key with
// its low bits cleared.  The LDS stack therefore holds only the entry distance,
// one float per level.
//
// One persistent workgroup per CU; lanes take work items from a global counter
// (64 consecutive items per wave request).  An item is `chunk` consecutive frames
// of one pixel; the pixels come in the order of `order` (most expensive first, from
// the previous launches' per-pixel iteration counts), so the launch ends on cheap
// items.  Items only decide who computes a sample, never its value.
// ---------------------------------------------------------------------------
// lane states below the tree levels (cur_k >= 1: at an internal node of that level, 0: at a leaf):
// K_POP: take a pending child; K_IDLE: holds an item, its ray is finished (or not started);
// K_NOITEM: holds no item.  One register and single comparisons, so that the wave votes of the
// loop control are the comparison masks themselves.
constexpr int K_POP = -1, K_IDLE = -2, K_NOITEM = -3;
#ifndef PT_EXPANSIONS
#define PT_EXPANSIONS 4
#endif
constexpr int EXPANSIONS = PT_EXPANSIONS;      // node expansions per pass of the traversal loop

// GEN: the tree has explicit child ids (PT_BUILD_TOPOLOGY_MORTON: no closed form) and an explicit per-lane stack
// of (node, entry distance) pairs: `levels` then is the number of stack entries held in LDS, deeper entries go to
// `gstack` (entry e of lane L of workgroup B at gstack[(e - levels) * lanes_total + B * blockDim + L]).
// One wave's share of k_trace_persistent (below), after the workgroup has filled its LDS.  XW (trees in LDS only): this
// wave keeps its per-level stack entries in global memory (`xst`: this lane's column of `levels` rows of 64 floats), as in
// dual_paths: the waves that the LDS beside the tree has no stack room for.
// (The structures arrive BY VALUE, as the kernel itself receives them: handed over by reference, the variant with child
// lines, which runs at the 128-register limit, got an allocation that waits at the top of every expansion for the refill
// round's sample stores before reusing their registers -- 17 % slower on the 100 M-triangle soup, where a store sits behind
// address-translation misses, and invisible at 10 M.)
template <bool COUNT, bool LDSN, bool NEAR, bool PAIR, bool GEN, bool XW>
__device__ __forceinline__ void persistent_paths(const SceneView sc, const TriIsect *__restrict__ tri_sorted, const int levels,
                                                 uint2 *__restrict__ gstack,
                                                 const unsigned n_slots,
                                                 const uint8_t *__restrict__ diffuse_tex,
                                                 const float4 *__restrict__ normal_tex,
                                                 const float4 *__restrict__ world_tex,
                                                 const uint32_t first_frame, const uint32_t count,
                                                 const uint32_t *__restrict__ seeds,
                                                 const typename StagingOf<LDSN>::type samples, DevCounters *__restrict__ counters,
                                                 unsigned *__restrict__ work_counter,
                                                 const uint32_t *__restrict__ order, uint32_t *__restrict__ pixel_work,
                                                 const uint32_t chunk, const Tuning tune, const TopLevels top,
                                                 const float4 *lds_nodes, const float4 *lds_mats, const float4 *lds_matq, const int *lds_off,
                                                 float *stack, const int nthreads, float *__restrict__ xst)
{
    const int n = sc.n;
    const bool TOP = !LDSN && !GEN && !PAIR;
    const int REFILL_T = tune.refill_t, LEAF_T = tune.leaf_t, ITR_LIMIT = tune.itr_limit;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const unsigned chunks = (count + chunk - 1) / chunk;     // items per pixel
    const unsigned total_items = n_slots * chunks;           // n_slots pixels: the whole image, or the listed region
    const float4 *mats = lds_mats;
    const pt_f3 g_pos = pt_mk3(sc.cb->pos.v[0], sc.cb->pos.v[1], sc.cb->pos.v[2]);
    const float root_mn[3] = {sc.root[0], sc.root[1], sc.root[2]}, root_mx[3] = {sc.root[3], sc.root[4], sc.root[5]};
    LaneCounters cnt;
    unsigned long long sched[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    // wave-uniform item queue
    unsigned q_next = 0, q_end = 0;
    bool exhausted = false;

    // per-item state
    unsigned px = 0;
    int matid0 = 0;
    pt_f3 hit_pos = pt_mk3(0, 0, 0), multi0 = pt_mk3(0, 0, 0), view0 = pt_mk3(0, 0, 0), normal0 = pt_mk3(0, 0, 0);
    uint32_t f = 0, f_end = 0;          // next local frame of the item, one past its last
    uint32_t work_start = 0;            // value of `iterations` when the lane took its item
    uint32_t iterations = 0;            // wave-uniform: traversal iterations of this wave so far
    // per-path / per-ray state
    bool path_active = false;
    int cur_k = K_NOITEM, cur_s = 0, cur_id = 0;
    unsigned pending = 0;
    unsigned pend_left = 0;             // NEAR only: levels whose pending child is the LEFT one
    uint32_t seed = 0;
    int itr = 1;
    pt_f3 multi = pt_mk3(0, 0, 0);
    Ray ray;
    ray.o = hit_pos; ray.d = pt_mk3(0, 0, 1); ray.t_min = PT_EPSILON; ray.t_max = 10000;
    pt_f3 inv = pt_mk3(1, 1, 1);
    float current_t = 0;
    int hit_leaf = -1;                  // leaf key of the nearest hit so far, -1: none (see dual_paths)
    float hit_bx = 0, hit_by = 0, hit_bz = 0;
    // global-memory variant with child lines (PAIR): the 128-byte line of a node holds the records of BOTH its children
    // (the node's own record follows from them), so the node AND whichever child is visited next are expanded from one
    // request.  next0..2 = the record of the child the lane went to; `next_valid` says that the current node is that child
    float4 next0 = make_float4(0, 0, 0, 0), next1 = next0, next2 = next0;
    bool next_valid = false;
    int next_l = 0, next_r = 0;         // GEN + PAIR: that child's own child ids
    // GEN: `pending` counts the stack entries; the deeper part of the stack of this lane
    uint2 *const gst = GEN ? gstack + ((size_t)blockIdx.x * nthreads + threadIdx.x) : nullptr;
    const size_t gstride = GEN ? (size_t)gridDim.x * nthreads : 0;
    float *const stack_t = stack + (GEN ? (size_t)levels * nthreads : 0);       // GEN: ids first, then the distances

    // the radiance sample of local frame fl (ray_tracing.hlsl:396-411: `multi` or black)
    auto emit = [&](const pt_f3 &color, uint32_t fl) {
        put_sample(samples, (size_t)px * count + fl, color.x, color.y, color.z);
    };

    // lanes to serve in a refill: K_IDLE always, K_NOITEM while the queue has items
    // ((cur_k - K_NOITEM) in [lo, 1] as one unsigned comparison)
    auto need_vote = [&]() {
        const unsigned lo = exhausted ? 1u : 0u;
        return ballot64((unsigned)(cur_k - K_NOITEM) - lo < 2u - lo);
    };

    for (;;) {
        // ------------------------------------------------------------ refill
        const unsigned long long need_mask = need_vote();
        const unsigned long long trav_mask = ballot64(cur_k >= K_POP);
        if (need_mask == 0 && trav_mask == 0) break;
        if (need_mask != 0 && (__popcll(need_mask) >= REFILL_T || trav_mask == 0)) {
            if (COUNT) sched[5]++;
            const long long t_refill = COUNT ? (long long)__builtin_readcyclecounter() : 0;
            for (int pass = 0; pass < 2; pass++) {
                // ---- take new pixels
                bool want = cur_k == K_NOITEM && !exhausted;
                unsigned long long wm = ballot64(want);
                while (wm != 0) {
                    if (q_next == q_end) {
                        unsigned base = 0;
                        if (lane == 0) base = atomicAdd(work_counter, 64u);
                        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                        if (base >= total_items) { exhausted = true; break; }
                        q_next = base;
                        q_end = min(base + 64u, total_items);
                    }
                    const unsigned rank = (unsigned)__popcll(wm & lt_mask);
                    const unsigned avail = q_end - q_next;
                    if (want && rank < avail) {
                        const unsigned item = q_next + rank, slot = item / chunks;
                        px = order ? order[slot] : slot;
                        f = (item - slot * chunks) * chunk;
                        f_end = min(f + chunk, count);
                        work_start = iterations;
                        const uchar4 dif8 = reinterpret_cast<const uchar4 *>(diffuse_tex)[px];
                        const float4 nrm4 = normal_tex[px];
                        const float4 pos4 = world_tex[px];
                        matid0 = mat_slot(__float_as_int(nrm4.w) - 1);  // asint(normal.w) - 1
                        if (__float_as_int(nrm4.w) <= 0) {
                            // ray_tracing.hlsl:361-362: black, but still accumulated every frame
                            for (uint32_t k = f; k < f_end; k++) { cnt.paths++; emit(pt_mk3(0, 0, 0), k); }
                        } else {
                            hit_pos = pt_mk3(pos4.x, pos4.y, pos4.z);
                            multi0 = pt_mk3(dif8.x / 255.0f, dif8.y / 255.0f, dif8.z / 255.0f);
                            if (mats[3 * matid0].y > 1) multi0 = pt_mk3(1, 1, 1);
                            view0 = pt_normalize3(pt_sub3(hit_pos, g_pos));
                            normal0 = pt_normalize3(pt_mk3(nrm4.x, nrm4.y, nrm4.z));
                            path_active = false;
                            cur_k = K_IDLE;
                            want = false;
                        }
                    }
                    q_next += min((unsigned)__popcll(wm), avail);
                    wm = ballot64(want);
                }
                // ---- shade finished rays, continue / end / start paths: ONE pass, so that the
                // expensive parts (first-bounce set-up, get_bounce, ray set-up) run once per round for
                // all idle lanes together; a lane left without a ray is served by the next round
                if (cur_k == K_IDLE) {
                    pt_f3 direction;
                    bool cont = false, bounce = false;
                    pt_f3 b_view = view0, b_normal = normal0;
                    int b_mat = matid0;
                    if (path_active) {
                        // back from intersect() (ray_tracing.hlsl:372-395)
                        bool ended = true;
                        if (hit_leaf >= 0) {
                            ray.o = pt_add3(ray.o, pt_scale3(ray.d, current_t));
                            itr++;
                            const float4 *shp = reinterpret_cast<const float4 *>(&sc.shade[hit_leaf]);
                            const float4 s0 = shp[0], s1 = shp[1], s2 = shp[2];
                            ended = false;
                            if (itr > MAX_ITR) {
                                const float pr = pt_max(pt_max(multi.x, multi.y), multi.z);
                                if (pr > rand_next(seed)) multi = pt_scale3(multi, 1 / pr);
                                else ended = true;
                            }
                            if (!ended && itr >= ITR_LIMIT) ended = true;
                            if (!ended) {
                                const pt_f3 A = pt_mk3(s0.x, s0.y, s0.z), B = pt_mk3(s1.x, s1.y, s1.z), C = pt_mk3(s2.x, s2.y, s2.z);
                                b_normal = pt_add3(pt_add3(pt_scale3(A, hit_bz), pt_scale3(B, hit_bx)), pt_scale3(C, hit_by));
                                b_view = ray.d;
                                b_mat = __float_as_int(s0.w);
                                bounce = true;
                            }
                        }
                        if (ended) {
                            emit(pt_mk3(0, 0, 0), f);
                            path_active = false;
                            f++;
                        }
                    }
                    if (!path_active) {
                        if (f < f_end) {
                            // new path of frame f (ray_tracing.hlsl:348-370, 410)
                            cnt.paths++;
                            // rand_init(pixel, seed of the frame), left in the path's staging slot by k_path_seeds
                            seed = path_seed(samples, (size_t)px * count + f);
                            multi = multi0;
                            itr = 1;
                            ray.o = hit_pos;
                            path_active = true;
                            bounce = true;
                        } else {
                            // cost of the item = traversal iterations its lane was occupied
                            if (pixel_work && iterations != work_start) atomicAdd(&pixel_work[px], iterations - work_start);
                            cur_k = K_NOITEM;
                        }
                    }
                    if (bounce) {
                        const uint32_t framecount = first_frame + f;
                        if (get_bounce(mats, lds_matq, framecount, b_view, b_normal, b_mat, multi, direction, seed)) {
                            cont = true;
                        } else {
                            emit(multi, f);
                            path_active = false;
                            f++;
                        }
                    }
                    if (cont) {
                        // start intersect() (ray_tracing.hlsl:191-214) with the root's own box test;
                        // a ray that misses the root stays idle and is finished as a miss next round
                        ray.d = direction;
                        inv = pt_mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
                        current_t = ray.t_max;
                        hit_leaf = -1;
                        pending = 0;
                        pend_left = 0;
                        cnt.rays++;
                        if (COUNT) { cnt.node_visits++; cnt.box_tests++; }
                        float t_entry;
                        if (slab(root_mn, root_mx, ray.o, inv, t_entry) && t_entry <= current_t) {
                            cur_k = GEN ? 1 : levels; cur_s = 0; cur_id = 0;
                            next_valid = false;
                        }
                    }
                }
                if (exhausted || ballot64(cur_k == K_NOITEM) == 0) break;
            }
            if (COUNT) sched[8] += (unsigned long long)((long long)__builtin_readcyclecounter() - t_refill);
        }
        // waves in the traversal loop issue ahead of waves in a (long, VALU-dense) refill round: 0.8 % per launch
        __builtin_amdgcn_s_setprio(1);
        const long long t_trav = COUNT ? (long long)__builtin_readcyclecounter() : 0;
        // --------------------------------------------------------- traversal
        // One loop pass = EXPANSIONS node expansions, each followed by a pop attempt, with the loop control
        // and the triangle-test decision once per pass (they cost as much as a fifth of an expansion).
        auto expand = [&]() {
                if (COUNT) {
                    sched[0]++;
                    sched[1] += __popcll(ballot64(cur_k >= 1));
                    sched[4] += __popcll(ballot64(cur_k == 0));                     // waiting at a leaf
                    sched[6] += __popcll(ballot64(cur_k == K_IDLE));                       // waiting for a refill
                    sched[7] += __popcll(ballot64(cur_k == K_NOITEM));                     // no item (queue empty)
                }
                if (GEN) {
                    if (cur_k >= 1) {
                        float4 w0, w1, w2;
                        float4 a0 = next0, a1 = next1, a2 = next2, b0 = next0, b1 = next1, b2 = next2;
                        int left, right, ll = 0, lr = 0, rl = 0, rr = 0;
                        if (!PAIR) {
                            const float4 *w = reinterpret_cast<const float4 *>(sc.wide + cur_id);
                            w0 = w[0]; w1 = w[1]; w2 = w[2];
                            const float4 w3 = w[3];
                            left = __float_as_int(w3.x); right = __float_as_int(w3.y);
                        } else if (next_valid) {
                            w0 = next0; w1 = next1; w2 = next2; left = next_l; right = next_r;
                        } else {
                            // one 128-byte line: the records of both children and the child ids of all three nodes
                            const float4 *w = sc.pair + (size_t)cur_id * 8;
                            a0 = w[0]; a1 = w[1]; a2 = w[2];
                            b0 = w[3]; b1 = w[4]; b2 = w[5];
                            const float4 ids = w[6], idr = w[7];
                            left = __float_as_int(ids.x); right = __float_as_int(ids.y);
                            ll = __float_as_int(ids.z); lr = __float_as_int(ids.w);
                            rl = __float_as_int(idr.x); rr = __float_as_int(idr.y);
                            record_from_children(a0, a1, a2, b0, b1, b2, w0, w1, w2);
                        }
                        const bool fetched_pair = PAIR && !next_valid;
                        const float lmn[3] = {w0.x, w0.y, w0.z}, lmx[3] = {w0.w, w1.x, w1.y};
                        const float rmn[3] = {w1.z, w1.w, w2.x}, rmx[3] = {w2.y, w2.z, w2.w};
                        float tl, tr;
                        const bool hl = slab(lmn, lmx, ray.o, inv, tl) && tl <= current_t;
                        const bool hr = slab(rmn, rmx, ray.o, inv, tr) && tr <= current_t;
                        if (COUNT) { cnt.node_visits += 2; cnt.box_tests += 2; }
                        const bool right_first = NEAR && hl && hr && tr < tl;
                        if (hl && hr) {
                            const int later = right_first ? left : right;
                            const float t_later = right_first ? tl : tr;
                            if ((int)pending < levels) {
                                stack[__umul24(pending, (unsigned)nthreads)] = __int_as_float(later);
                                stack_t[__umul24(pending, (unsigned)nthreads)] = t_later;
                            } else {
                                gst[(size_t)((int)pending - levels) * gstride] = make_uint2((unsigned)later, __float_as_uint(t_later));
                            }
                            pending++;
                        }
                        const bool go_left = hl && !right_first;
                        const int nxt = go_left ? left : right;
                        if (!(hl || hr)) cur_k = K_POP;
                        else if (nxt < 0) { cur_k = 0; cur_s = ~nxt; }
                        else cur_id = nxt;                  // cur_k stays 1
                        if (PAIR) {
                            // the record of the child the lane moved to came with this node's line
                            next0 = go_left ? a0 : b0; next1 = go_left ? a1 : b1; next2 = go_left ? a2 : b2;
                            next_l = go_left ? ll : rl; next_r = go_left ? lr : rr;
                            next_valid = fetched_pair && (hl || hr) && nxt >= 0;
                        }
                    }
                } else
                if (cur_k >= 1) {                       // only traversing lanes have cur_k >= K_POP
                    float4 w0, w1, w2;
                    float4 a0 = next0, a1 = next1, a2 = next2, b0 = next0, b1 = next1, b2 = next2;
                    bool from_top = false;
                    if (LDSN) {
                        const float4 *w = lds_nodes + __umul24((unsigned)cur_id, 3u);      // 24-bit multiply: full rate
                        w0 = w[0]; w1 = w[1]; w2 = w[2];
                    } else if (TOP && cur_k >= top.k0) {
                        // an upper level of the tree: its records are in LDS under (level offset + block)
                        const float4 *w = lds_nodes + __umul24((unsigned)(lds_off[cur_k] + (cur_s >> cur_k)), 3u);
                        w0 = w[0]; w1 = w[1]; w2 = w[2];
                        from_top = true;
                    } else if (!PAIR) {
                        const float4 *w = reinterpret_cast<const float4 *>(sc.wide + cur_id);
                        w0 = w[0]; w1 = w[1]; w2 = w[2];
                    } else if (next_valid) {
                        w0 = next0; w1 = next1; w2 = next2;        // came with the parent's line: no fetch
                    } else {
                        // one 128-byte line: the records of both children.  The memory system serves a random 128-byte
                        // line at the rate of a random 64-byte one, so the step from this node to either child costs
                        // no request of its own.
                        const float4 *w = sc.pair + (size_t)cur_id * 8;
                        a0 = w[0]; a1 = w[1]; a2 = w[2];
                        b0 = w[3]; b1 = w[4]; b2 = w[5];
                        record_from_children(a0, a1, a2, b0, b1, b2, w0, w1, w2);
                    }
                    const bool fetched_pair = PAIR && !next_valid && !from_top;
                    // the right child (first key sr, level kr) is worked out whether or not it is taken: selects
                    // instead of nested branches shorten the wave's instruction stream.  For trees in LDS it is worked out
                    // while the record is on its way (see dual_paths: the empty asm statement keeps it there).
                    if (LDSN) __builtin_amdgcn_sched_barrier(0);
                    const int k = cur_k, half = 1 << (k - 1);
                    int sr = cur_s + half;
                    const int e = min(sr + half - 1, n);
                    int kr = 32 - __clz(sr ^ e);
                    if (LDSN) {
                        asm volatile("" : "+v"(kr), "+v"(sr));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const float lmn[3] = {w0.x, w0.y, w0.z}, lmx[3] = {w0.w, w1.x, w1.y};
                    const float rmn[3] = {w1.z, w1.w, w2.x}, rmx[3] = {w2.y, w2.z, w2.w};
                    float tl, tr;
                    const bool hl = slab(lmn, lmx, ray.o, inv, tl) && tl <= current_t;
                    const bool hr = slab(rmn, rmx, ray.o, inv, tr) && tr <= current_t;
                    if (COUNT) { cnt.node_visits += 2; cnt.box_tests += 2; }
                    if (NEAR) {
                        // NON-PARITY option: of two children that both pass, the one the ray enters first is
                        // visited first (ties: left, the reference's order); the other waits with ITS distance
                        const bool right_first = hl && hr && tr < tl;
                        if (hl && hr) {
                            if (XW) xst[(k - 1) * 64] = right_first ? tl : tr;
                            else stack[__umul24((unsigned)(k - 1), (unsigned)nthreads)] = right_first ? tl : tr;
                            pending |= 1u << k;
                            if (right_first) pend_left |= 1u << k;
                        }
                        const bool go_left = hl && !right_first;
                        cur_k = go_left ? k - 1 : (hr ? kr : K_POP);
                        cur_id = go_left ? sr - 1 : sr;
                        cur_s = go_left ? cur_s : sr;
                    } else {
                        if (hl && hr) {
                            if (XW) xst[(k - 1) * 64] = tr;
                            else stack[__umul24((unsigned)(k - 1), (unsigned)nthreads)] = tr;
                            pending |= 1u << k;
                        }
                        cur_k = hl ? k - 1 : (hr ? kr : K_POP);
                        cur_id = hl ? sr - 1 : sr;
                        cur_s = hl ? cur_s : sr;          // after a double miss only the bits above the pending levels matter
                    }
                    if (PAIR) {
                        // the lane moved to the left child (level k - 1) or to the right one: that child's record came
                        // with this node's line
                        const bool went_left = cur_id == sr - 1;
                        next0 = went_left ? a0 : b0; next1 = went_left ? a1 : b1; next2 = went_left ? a2 : b2;
                        next_valid = fetched_pair && cur_k >= 1;
                    }
                }
        };
        auto pop = [&]() {
                if (GEN) {
                    if (cur_k == K_POP) {
                        if (pending == 0) {
                            cur_k = K_IDLE;
                        } else {
                            pending--;
                            int id;
                            float t_entry;
                            if ((int)pending < levels) {
                                id = __float_as_int(stack[__umul24(pending, (unsigned)nthreads)]);
                                t_entry = stack_t[__umul24(pending, (unsigned)nthreads)];
                            } else {
                                const uint2 e = gst[(size_t)((int)pending - levels) * gstride];
                                id = (int)e.x; t_entry = __uint_as_float(e.y);
                            }
                            if (t_entry <= current_t) {
                                if (PAIR) next_valid = false;
                                if (id < 0) { cur_k = 0; cur_s = ~id; }
                                else { cur_k = 1; cur_id = id; }
                            }
                        }
                    }
                } else
                if (cur_k == K_POP) {
                    // one pending child per iteration; an entry whose entry distance no longer passes
                    // the distance test is dropped and the lane tries again next iteration
                    if (pending == 0) {
                        cur_k = K_IDLE;
                    } else {
                        const int L = __ffs((int)pending) - 1;              // deepest pending ancestor level
                        pending &= pending - 1;
                        const float t_entry = XW ? xst[(L - 1) * 64] : stack[__umul24((unsigned)(L - 1), (unsigned)nthreads)];
                        const bool left = NEAR && ((pend_left >> L) & 1u);
                        if (NEAR) pend_left &= ~(1u << L);
                        if (t_entry <= current_t) {
                            if (PAIR) next_valid = false;        // a node reached from the stack has its own line
                            const int sa = (cur_s >> L) << L, half = 1 << (L - 1);
                            const int sr = sa + half, e = min(sa + 2 * half - 1, n);
                            // a left child covers its 2^(L-1) keys fully: level L - 1, named by its last key
                            cur_k = left ? L - 1 : 32 - __clz(sr ^ e);
                            cur_s = left ? sa : sr;
                            cur_id = left ? sr - 1 : sr;
                        }
                    }
                }
        };
        for (;;) {
            if (ballot64(cur_k >= K_POP) == 0) break;
            if (__popcll(need_vote()) >= REFILL_T) break;
            iterations += EXPANSIONS;
#pragma unroll
            for (int u = 0; u < EXPANSIONS - 1; u++) { expand(); pop(); }
            expand();
            const bool at_leaf = cur_k == 0;
            const unsigned long long leaf_mask = ballot64(at_leaf);
            if (leaf_mask != 0 && (__popcll(leaf_mask) >= LEAF_T || ballot64(cur_k >= 1) == 0)) {
                if (COUNT) { sched[2]++; sched[3] += __popcll(leaf_mask); }
                if (at_leaf) {
                    const float4 *tp = reinterpret_cast<const float4 *>(&tri_sorted[cur_s]);
                    float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                    asm volatile("" : "+v"(t0.x), "+v"(t0.y), "+v"(t0.z));      // all three loads leave together (see k_trace_dual)
                    if (COUNT) cnt.tri_tests++;
                    // ray_tracing.hlsl:167-189
                    const pt_f3 a = pt_mk3(t0.x, t0.y, t0.z), e_1 = pt_mk3(t1.x, t1.y, t1.z), e_2 = pt_mk3(t2.x, t2.y, t2.z);
                    const pt_f3 q = pt_cross3(ray.d, e_2);
                    const float a1 = pt_dot3(e_1, q);
                    if (!(fabsf(a1) <= PT_EPSILON)) {
                        const pt_f3 oa = pt_sub3(ray.o, a);
                        const pt_f3 s = pt_mk3(oa.x / a1, oa.y / a1, oa.z / a1);
                        const pt_f3 r = pt_cross3(s, e_1);
                        const float t = pt_dot3(e_2, r);
                        const float bx = pt_dot3(s, q);
                        const float by = pt_dot3(r, ray.d);
                        const float bz = 1 - bx - by;
                        if (bx >= -PT_EPSILON && by >= -PT_EPSILON && bz >= -PT_EPSILON && t >= PT_EPSILON) {
                            if (current_t > t && t > ray.t_min) {
                                hit_bx = bx; hit_by = by; hit_bz = bz;
                                current_t = t;
                                hit_leaf = cur_s;          // the shading record is in leaf order too
                            }
                        }
                    }
                    cur_k = K_POP;
                }
            }
            pop();
        }
        __builtin_amdgcn_s_setprio(0);
        if (COUNT) sched[9] += (unsigned long long)((long long)__builtin_readcyclecounter() - t_trav);
    }
    flush_counters(cnt, counters, COUNT);
    if (COUNT && lane == 0) {
        for (int k = 0; k < 10; k++) atomicAdd(&counters->sched[k], sched[k]);
    }
}

template <bool COUNT, bool LDSN, bool NEAR, bool PAIR = false, bool GEN = false>
__global__ __launch_bounds__(1024) void k_trace_persistent(SceneView sc, const TriIsect *__restrict__ tri_sorted, int levels,
                                                    uint2 *__restrict__ gstack,
                                                    unsigned n_slots,
                                                    const uint8_t *__restrict__ diffuse_tex,
                                                    const float4 *__restrict__ normal_tex,
                                                    const float4 *__restrict__ world_tex,
                                                    uint32_t first_frame, uint32_t count,
                                                    const uint32_t *__restrict__ seeds,
                                                    const typename StagingOf<LDSN>::type samples, DevCounters *__restrict__ counters,
                                                    unsigned *__restrict__ work_counter,
                                                    const uint32_t *__restrict__ order, uint32_t *__restrict__ pixel_work,
                                                    uint32_t chunk, Tuning tune, TopLevels top, int lds_threads, float *__restrict__ xstack)
{
    // LDS: node records (3 float4 each; LDSN: all of them, else the levels top.k0 and above of an index-keyed tree), the
    // 16 materials (3 float4 each), their quotient records (1 float4 each), the level offsets of the top part, then the
    // per-lane stacks
    extern __shared__ __attribute__((aligned(16))) float4 lds_nodes[];
    const int n = sc.n;
    const int nthreads = blockDim.x;
    const bool TOP = !LDSN && !GEN && !PAIR;       // (with child lines the extra branch cost 18 % at 100 M triangles)
    float4 *lds_mats = lds_nodes + (LDSN ? 3 * n : (TOP ? 3 * top.nodes : 0));
    float4 *lds_matq = lds_mats + 3 * MAT_SLOTS;
    int *lds_off = reinterpret_cast<int *>(lds_matq + MAT_SLOTS);
    float *stack = reinterpret_cast<float *>(lds_matq + MAT_SLOTS + (TOP ? 9 : 0)) + threadIdx.x;   // level L at stack[(L-1) * nthreads]
    if (LDSN) {
        for (int i = threadIdx.x; i < 3 * n; i += nthreads) {
            const int node = i / 3, part = i - 3 * node;
            lds_nodes[i] = reinterpret_cast<const float4 *>(&sc.wide[node])[part];
        }
    }
    if (TOP) {
        if (threadIdx.x < 33) lds_off[threadIdx.x] = top.off[threadIdx.x];
        for (int i = threadIdx.x; i < top.nodes; i += nthreads) {
            // compact index -> (level, block) -> node id: a left child is named by its last key, others by their first
            int k = levels;
            while (k > top.k0 && i >= top.off[k - 1]) k--;
            const long long b = i - top.off[k];
            const int id = ((b & 1) == 0 && ((b + 1) << k) <= (long long)n) ? (int)(((b + 1) << k) - 1) : (int)(b << k);
            const float4 *src = reinterpret_cast<const float4 *>(&sc.wide[id]);
            lds_nodes[3 * i] = src[0]; lds_nodes[3 * i + 1] = src[1]; lds_nodes[3 * i + 2] = src[2];
        }
    }
    if (threadIdx.x < 3 * MAT_SLOTS)        // the device copy of the constants ends with the zero material
        lds_mats[threadIdx.x] = reinterpret_cast<const float4 *>(sc.cb->mats)[threadIdx.x];
    if (threadIdx.x < MAT_SLOTS) {
        const float4 *mp = reinterpret_cast<const float4 *>(sc.cb->mats) + 3 * threadIdx.x;
        lds_matq[threadIdx.x] = material_quotients(mp[0], mp[1], mp[2]);
    }
    __syncthreads();

    // trees in LDS: waves beyond `lds_threads` keep their stack entries in global memory (see k_trace_dual); every other
    // variant is launched with lds_threads = blockDim.x
    if (LDSN && !PAIR && !GEN && __builtin_amdgcn_readfirstlane((int)threadIdx.x) >= lds_threads) {
        const size_t xwave = (size_t)blockIdx.x * ((blockDim.x - lds_threads) >> 6) + ((threadIdx.x - lds_threads) >> 6);
        persistent_paths<COUNT, LDSN, NEAR, PAIR, GEN, LDSN && !PAIR && !GEN>(
            sc, tri_sorted, levels, gstack, n_slots, diffuse_tex, normal_tex, world_tex, first_frame, count, seeds, samples, counters,
            work_counter, order, pixel_work, chunk, tune, top, lds_nodes, lds_mats, lds_matq, lds_off, stack, lds_threads,
            xstack + xwave * 64 * (size_t)levels + (threadIdx.x & 63));
    } else {
        // (variants without extra waves: the workgroup's own size, whose range the compiler knows)
        persistent_paths<COUNT, LDSN, NEAR, PAIR, GEN, false>(
            sc, tri_sorted, levels, gstack, n_slots, diffuse_tex, normal_tex, world_tex, first_frame, count, seeds, samples, counters,
            work_counter, order, pixel_work, chunk, tune, top, lds_nodes, lds_mats, lds_matq, lds_off, stack,
            (LDSN && !PAIR && !GEN) ? lds_threads : nthreads, nullptr);
    }
}

// ---------------------------------------------------------------------------
// k_trace_dual: the persistent kernel for trees in LDS (reference visit order) with TWO paths per lane, two
// frames of the lane's pixel.  One path owns the traversal slot; the other is PARKED in registers, either as a
// ray that is ready to be traversed or as a finished ray that waits for shading.
//   * When the traversed ray ends, the lane exchanges it for the parked ray (register moves), so a lane idles
//     only when BOTH its paths wait for shading; in k_trace_persistent every finished ray idles until the next
//     refill round.
//   * A refill round shades the parked slots: all lanes that finished a ray since the last round, not only
//     the lanes that happen to be idle, run the (long) shading code together.
// Per path the sequence of random draws, tests and arithmetic is the sequential loop's; the two frames of a
// lane are independent paths with their own seeds and sample slots, so results do not change.
// ---------------------------------------------------------------------------
constexpr int K_EMPTY = -4;                              // holds an item, no path in the traversal slot
// parked slot (even = a refill round has to serve it): P_START: free, the item has frames left; P_READY: a ray past the
// root test; P_DONE: a finished ray; P_NONE: free, no frames left
constexpr int P_START = 0, P_READY = 1, P_DONE = 2, P_NONE = 3;

__device__ __forceinline__ void lane_swap(float &a, float &b) { const float t = a; a = b; b = t; }

// One wave's share of k_trace_dual (below), after the workgroup has filled its LDS.  XW: this wave keeps its per-level stack
// entries in global memory (`xst`: this lane's column of `levels` rows of 64 floats) instead of LDS -- the one wave more than
// the LDS beside the tree has stack room for.  The function is instantiated once per kind of wave, each copy with its own
// local state, so the waves with their stack in LDS run an instruction stream that never tests the wave's kind.
template <bool COUNT, bool LDSN, bool XW>
__device__ __forceinline__ void dual_paths(const SceneView &sc, const TriIsect *__restrict__ tri_sorted, const int levels,
                                           const unsigned n_slots,
                                           const uint8_t *__restrict__ diffuse_tex,
                                           const float4 *__restrict__ normal_tex,
                                           const float4 *__restrict__ world_tex,
                                           const uint32_t first_frame, const uint32_t count,
                                           const typename StagingOf<LDSN>::type samples, DevCounters *__restrict__ counters,
                                           unsigned *__restrict__ work_counter,
                                           const uint32_t *__restrict__ order, uint32_t *__restrict__ pixel_work,
                                           const uint32_t chunk, const Tuning &tune, const TopLevels &top,
                                           const float4 *lds_nodes, const float4 *lds_mats, const float4 *lds_matq, const int *lds_off,
                                           float *stack, const int nthreads, float *__restrict__ xst)
{
    const int n = sc.n;
    // STARVE_T: lanes with nothing to traverse in either slot; SHADE_T: parked slots waiting for a refill round;
    // SWAP_T: lanes whose ray ended while a parked ray is ready
    const int STARVE_T = tune.refill_t, SHADE_T = tune.shade_t, LEAF_T = tune.leaf_t, ITR_LIMIT = tune.itr_limit, SWAP_T = tune.swap_t;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // the stack entry of level L sits (L - 1) rows above the lane's first one: one multiply-add on the byte address
    const unsigned row_bytes = XW ? 64u * 4u : (unsigned)nthreads * 4u;
    char *const stack_b = reinterpret_cast<char *>(XW ? xst : stack);
    // items: first every pixel's coarse chunks of the frames [0, coarse_frames), then every pixel's fine chunks of the rest
    const unsigned coarse_frames = count - (unsigned)tune.fine_frames, fine_chunk = (unsigned)tune.fine_chunk;
    const unsigned chunks = (coarse_frames + chunk - 1) / chunk;
    const unsigned fine_chunks = tune.fine_frames ? ((unsigned)tune.fine_frames + fine_chunk - 1) / fine_chunk : 0u;
    const unsigned coarse_items = n_slots * chunks;
    const unsigned total_items = coarse_items + n_slots * fine_chunks;
    const float4 *mats = lds_mats;
    const pt_f3 g_pos = pt_mk3(sc.cb->pos.v[0], sc.cb->pos.v[1], sc.cb->pos.v[2]);
    const float root_mn[3] = {sc.root[0], sc.root[1], sc.root[2]}, root_mx[3] = {sc.root[3], sc.root[4], sc.root[5]};
    const float T_MIN = PT_EPSILON, T_MAX = 10000;
    LaneCounters cnt;
    unsigned long long sched[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    unsigned q_next = 0, q_end = 0;
    bool exhausted = false;
    unsigned long long dead_mask = 0;   // lanes without an item after the queue ran dry

    // per-item state
    unsigned px = 0;
    int matid0 = 0;
    pt_f3 hit_pos = pt_mk3(0, 0, 0), multi0 = pt_mk3(0, 0, 0), view0 = pt_mk3(0, 0, 0), normal0 = pt_mk3(0, 0, 0);
    uint32_t f = 0, f_end = 0;          // next local frame of the item that has no path yet, one past the item's last
    uint32_t work_start = 0, iterations = 0;
    // the lane's two paths (seed, bounce count, local frame, throughput); path `act` owns the traversal slot, the other
    // one the parked slot.  Only the shading code reads or writes them.
    uint32_t sd0 = 0, sd1 = 0, fl0 = 0, fl1 = 0, px0 = 0, px1 = 0;
    int it0 = 1, it1 = 1, act = 0;
    pt_f3 mu0 = pt_mk3(0, 0, 0), mu1 = mu0;
    // traversal slot: cur_k >= K_POP: traversing; K_IDLE: its ray is finished, the result is still here; K_EMPTY / K_NOITEM: no path
    int cur_k = K_NOITEM, cur_s = 0, cur_id = 0;
    unsigned pending = 0;
    pt_f3 ro = pt_mk3(0, 0, 0), rd = pt_mk3(0, 0, 1), inv = pt_mk3(1, 1, 1);
    float current_t = 0;
    // leaf key of the nearest hit so far, -1: none.  (A flag of its own would be a wave mask in a scalar register pair that the
    // compiler patches at the end of every divergent branch of the loop: folded into the key, 105.2 -> 102.9 ms per 128 frames.)
    int hit_leaf = -1;
    float hit_bx = 0, hit_by = 0;
    // parked slot.  p_q: P_READY: 1 / direction; P_DONE: {leaf key of the hit or -1 (bits), barycentric x, y}
    int p_state = P_NONE;
    pt_f3 p_o = pt_mk3(0, 0, 0), p_d = pt_mk3(0, 0, 1);
    float p_q0 = 0, p_q1 = 0, p_q2 = 0;

    // (a path carries its pixel: the lane may have gone on to its next item while the path is still in flight)
    auto emit = [&](const pt_f3 &color, unsigned pixel, uint32_t fl) {
        put_sample(samples, (size_t)pixel * count + fl, color.x, color.y, color.z);
    };
    // The finished ray of the traversal slot (if any) moves to the parked slot, its origin advanced to the hit point
    // (ray_tracing.hlsl:376); the parked ray (if any) moves to the traversal slot.  For lanes that are not traversing and
    // whose parked slot holds a ray (`take`) or is free.
    auto exchange = [&](bool take) {
        const bool done = cur_k == K_IDLE;
        if (done && hit_leaf >= 0) ro = pt_add3(ro, pt_scale3(rd, current_t));
        inv.x = __int_as_float(hit_leaf); inv.y = hit_bx; inv.z = hit_by;      // the outgoing p_q
        lane_swap(ro.x, p_o.x); lane_swap(ro.y, p_o.y); lane_swap(ro.z, p_o.z);
        lane_swap(rd.x, p_d.x); lane_swap(rd.y, p_d.y); lane_swap(rd.z, p_d.z);
        lane_swap(inv.x, p_q0); lane_swap(inv.y, p_q1); lane_swap(inv.z, p_q2);
        act ^= 1;
        p_state = done ? P_DONE : (f < f_end ? P_START : P_NONE);
        if (take) {
            current_t = T_MAX; hit_leaf = -1; pending = 0;
            cur_k = levels; cur_s = 0; cur_id = 0;
        } else {
            cur_k = K_EMPTY;
        }
    };

    for (;;) {
        {
            const bool idle = cur_k == K_IDLE || cur_k == K_EMPTY, take = p_state == P_READY;
            if (idle && take) exchange(true);
        }
        const unsigned long long trav_mask = ballot64(cur_k >= K_POP);
        const unsigned long long shade_mask = ballot64((p_state & 1) == 0);
        const unsigned long long starved_mask = ~trav_mask & ~dead_mask;      // after the exchange: idle lanes have no ready ray
        if (starved_mask == 0 && trav_mask == 0) break;
        // ------------------------------------------------------------ refill
        // (every starved lane needs the round: an item, its finished ray parked and shaded, or its item closed)
        if ((shade_mask | starved_mask) != 0 &&
            (__popcll(starved_mask) >= STARVE_T || __popcll(shade_mask) >= SHADE_T || trav_mask == 0)) {
            if (COUNT) sched[5]++;
            const long long t_refill = COUNT ? (long long)__builtin_readcyclecounter() : 0;
            for (int pass = 0; pass < 2; pass++) {
                // ---- take new pixels
                // a lane asks for its next item as soon as the current one has no frame left to start, while that item's last
                // paths are still in flight: otherwise an item would end on a lone path, with nothing parked beside it
                bool want = f == f_end && !exhausted;
                unsigned long long wm = ballot64(want);
                while (wm != 0) {
                    if (q_next == q_end) {
                        unsigned base = 0;
                        if (lane == 0) base = atomicAdd(work_counter, 64u);
                        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                        if (base >= total_items) { exhausted = true; break; }
                        q_next = base;
                        q_end = min(base + 64u, total_items);
                    }
                    const unsigned rank = (unsigned)__popcll(wm & lt_mask);
                    const unsigned avail = q_end - q_next;
                    if (want && rank < avail) {
                        const unsigned item = q_next + rank;
                        unsigned slot;
                        if (item < coarse_items) {
                            slot = item / chunks;
                            f = (item - slot * chunks) * chunk;
                            f_end = min(f + chunk, coarse_frames);
                        } else {
                            const unsigned j = item - coarse_items;
                            slot = j / fine_chunks;
                            f = coarse_frames + (j - slot * fine_chunks) * fine_chunk;
                            f_end = min(f + fine_chunk, count);
                        }
                        if (cur_k != K_NOITEM && pixel_work && iterations != work_start) atomicAdd(&pixel_work[px], iterations - work_start);
                        px = order ? order[slot] : slot;
                        work_start = iterations;
                        const uchar4 dif8 = reinterpret_cast<const uchar4 *>(diffuse_tex)[px];
                        const float4 nrm4 = normal_tex[px];
                        const float4 pos4 = world_tex[px];
                        matid0 = mat_slot(__float_as_int(nrm4.w) - 1);  // asint(normal.w) - 1
                        if (__float_as_int(nrm4.w) <= 0) {
                            // ray_tracing.hlsl:361-362: black, but still accumulated every frame
                            for (uint32_t k = f; k < f_end; k++) { cnt.paths++; emit(pt_mk3(0, 0, 0), px, k); }
                            f = f_end;
                        } else {
                            hit_pos = pt_mk3(pos4.x, pos4.y, pos4.z);
                            multi0 = pt_mk3(dif8.x / 255.0f, dif8.y / 255.0f, dif8.z / 255.0f);
                            if (mats[3 * matid0].y > 1) multi0 = pt_mk3(1, 1, 1);
                            view0 = pt_normalize3(pt_sub3(hit_pos, g_pos));
                            normal0 = pt_normalize3(pt_mk3(nrm4.x, nrm4.y, nrm4.z));
                            if (cur_k == K_NOITEM) cur_k = K_EMPTY;
                            if (p_state == P_NONE) p_state = P_START;          // an item has at least one frame
                            want = false;
                        }
                    }
                    q_next += min((unsigned)__popcll(wm), avail);
                    wm = ballot64(want);
                }
                // ---- a finished ray whose lane has a free parked slot moves there to be shaded in this round
                if (cur_k == K_IDLE && (p_state == P_START || p_state == P_NONE)) exchange(false);
                // ---- shade the parked slots: finished rays continue or end their path, free slots start the next
                // frame's path; ONE pass of the shading code for all of them
                if ((p_state & 1) == 0) {
                    // the parked path is the one that does not own the traversal slot
                    uint32_t p_seed = act ? sd0 : sd1, p_fl = act ? fl0 : fl1, p_px = act ? px0 : px1;
                    int p_itr = act ? it0 : it1;
                    pt_f3 p_multi = pt_mk3(act ? mu0.x : mu1.x, act ? mu0.y : mu1.y, act ? mu0.z : mu1.z);
                    pt_f3 direction;
                    bool cont = false, bounce = false;
                    pt_f3 b_view = view0, b_normal = normal0;
                    int b_mat = matid0;
                    if (p_state == P_DONE) {
                        // back from intersect() (ray_tracing.hlsl:372-395); p_o is the hit point already
                        bool ended = true;
                        const int leaf = __float_as_int(p_q0);
                        if (leaf >= 0) {
                            p_itr++;
                            const float4 *shp = reinterpret_cast<const float4 *>(&sc.shade[leaf]);
                            const float4 s0 = shp[0], s1 = shp[1], s2 = shp[2];
                            ended = false;
                            if (p_itr > MAX_ITR) {
                                const float pr = pt_max(pt_max(p_multi.x, p_multi.y), p_multi.z);
                                if (pr > rand_next(p_seed)) p_multi = pt_scale3(p_multi, 1 / pr);
                                else ended = true;
                            }
                            if (!ended && p_itr >= ITR_LIMIT) ended = true;
                            if (!ended) {
                                const float bx = p_q1, by = p_q2, bz = 1 - bx - by;
                                const pt_f3 A = pt_mk3(s0.x, s0.y, s0.z), B = pt_mk3(s1.x, s1.y, s1.z), C = pt_mk3(s2.x, s2.y, s2.z);
                                b_normal = pt_add3(pt_add3(pt_scale3(A, bz), pt_scale3(B, bx)), pt_scale3(C, by));
                                b_view = p_d;
                                b_mat = __float_as_int(s0.w);
                                bounce = true;
                            }
                        }
                        if (ended) {
                            emit(pt_mk3(0, 0, 0), p_px, p_fl);
                            p_state = f < f_end ? P_START : P_NONE;
                        }
                    }
                    if (p_state == P_START) {
                        // new path of frame f (ray_tracing.hlsl:348-370, 410)
                        cnt.paths++;
                        p_seed = path_seed(samples, (size_t)px * count + f);
                        p_multi = multi0;
                        p_itr = 1;
                        p_o = hit_pos;
                        p_fl = f;
                        p_px = px;
                        f++;
                        p_state = P_DONE;           // a path is in the slot (the outcome below decides its state)
                        bounce = true;
                    }
                    if (bounce) {
                        if (get_bounce(mats, lds_matq, first_frame + p_fl, b_view, b_normal, b_mat, p_multi, direction, p_seed)) {
                            cont = true;
                        } else {
                            emit(p_multi, p_px, p_fl);
                            p_state = f < f_end ? P_START : P_NONE;
                        }
                    }
                    if (cont) {
                        // intersect() starts with the root's own box test (ray_tracing.hlsl:191-214); a ray that misses
                        // the root is a finished ray without a hit
                        p_d = direction;
                        const pt_f3 pinv = pt_mk3(1.0f / direction.x, 1.0f / direction.y, 1.0f / direction.z);
                        cnt.rays++;
                        if (COUNT) { cnt.node_visits++; cnt.box_tests++; }
                        float t_entry;
                        if (slab(root_mn, root_mx, p_o, pinv, t_entry) && t_entry <= T_MAX) {
                            p_state = P_READY;
                            p_q0 = pinv.x; p_q1 = pinv.y; p_q2 = pinv.z;
                        } else {
                            p_state = P_DONE;
                            p_q0 = __int_as_float(-1);
                        }
                    }
                    if (act) { sd0 = p_seed; fl0 = p_fl; px0 = p_px; it0 = p_itr; mu0 = p_multi; }
                    else     { sd1 = p_seed; fl1 = p_fl; px1 = p_px; it1 = p_itr; mu1 = p_multi; }
                }
                if ((cur_k == K_IDLE || cur_k == K_EMPTY) && p_state == P_READY) exchange(true);
                // ---- items whose frames are all done
                if (cur_k == K_EMPTY && p_state == P_NONE) {
                    if (pixel_work && iterations != work_start) atomicAdd(&pixel_work[px], iterations - work_start);
                    cur_k = K_NOITEM;
                }
                if (exhausted || ballot64(cur_k == K_NOITEM) == 0) break;
            }
            if (exhausted) dead_mask = ballot64(cur_k == K_NOITEM);
            if (COUNT) sched[8] += (unsigned long long)((long long)__builtin_readcyclecounter() - t_refill);
        }
        __builtin_amdgcn_s_setprio(1);
        const long long t_trav = COUNT ? (long long)__builtin_readcyclecounter() : 0;
        // --------------------------------------------------------- traversal (as k_trace_persistent, tree in LDS)
        // Instruction placement matters here: a wave issues in order, four waves share a SIMD, and the record of a node takes
        // about a hundred clocks to arrive from LDS.  Whatever does not need the record (the right child's first key and level,
        // the stack slot, the pending mask with this level added) is computed between the reads and their first use; the empty
        // asm statement names those values so that the compiler keeps them there instead of sinking them behind the box
        // tests, and the scheduling barriers keep the reads in front.  The pop works out the node that the entry names while the
        // entry is on its way and then takes it or not by selects, without a branch.  (Together: 108.6 -> 105.9 ms per 128 frames.)
        auto expand = [&]() {
                if (COUNT) {
                    sched[0]++;
                    sched[1] += __popcll(ballot64(cur_k >= 1));
                    sched[4] += __popcll(ballot64(cur_k == 0));
                    sched[6] += __popcll(ballot64(cur_k == K_IDLE || cur_k == K_EMPTY));
                    sched[7] += __popcll(ballot64(cur_k == K_NOITEM));
                }
                if (cur_k >= 1) {
                    const float4 *w;
                    if (LDSN) w = lds_nodes + __umul24((unsigned)cur_id, 3u);
                    else if (cur_k >= top.k0) w = lds_nodes + __umul24((unsigned)(lds_off[cur_k] + (cur_s >> cur_k)), 3u);
                    else w = reinterpret_cast<const float4 *>(sc.wide + cur_id);
                    const float4 w0 = w[0], w1 = w[1], w2 = w[2];
                    __builtin_amdgcn_sched_barrier(0);
                    const int k = cur_k, half = 1 << (k - 1);
                    int km1 = k - 1, sr = cur_s + half;
                    const int e = min(sr + half - 1, n);
                    int kr = 32 - __clz(sr ^ e);
                    float *const slot = reinterpret_cast<float *>(stack_b + __umul24((unsigned)(k - 1), row_bytes));
                    unsigned pend_both = pending | (1u << k);
                    asm volatile("" : "+v"(kr), "+v"(sr), "+v"(pend_both), "+v"(km1));
                    __builtin_amdgcn_sched_barrier(0);
                    const float lmn[3] = {w0.x, w0.y, w0.z}, lmx[3] = {w0.w, w1.x, w1.y};
                    const float rmn[3] = {w1.z, w1.w, w2.x}, rmx[3] = {w2.y, w2.z, w2.w};
                    float tl, tr;
                    const bool hl = slab(lmn, lmx, ro, inv, tl) && tl <= current_t;
                    const bool hr = slab(rmn, rmx, ro, inv, tr) && tr <= current_t;
                    if (COUNT) { cnt.node_visits += 2; cnt.box_tests += 2; }
                    // (the entry is written whether or not both children pass: a row whose bit is not set is never read)
                    *slot = tr;
                    pending = (hl && hr) ? pend_both : pending;
                    cur_k = hl ? km1 : (hr ? kr : K_POP);
                    cur_id = hl ? sr - 1 : sr;
                    cur_s = hl ? cur_s : sr;
                }
        };
        auto pop = [&]() {
                if (cur_k == K_POP) {
                    // no inner branch: a lane without a pending entry reads the root level's row (bit `levels` stands in for
                    // a pending level) and only the final selects tell it apart
                    const unsigned pend = pending | (1u << levels);
                    const int L = __ffs((int)pend) - 1;
                    const float t_entry = *reinterpret_cast<const float *>(stack_b + __umul24((unsigned)(L - 1), row_bytes));
                    __builtin_amdgcn_sched_barrier(0);
                    const bool none = pending == 0;
                    pending &= pending - 1;
                    const int sa = (cur_s >> L) << L, half = 1 << (L - 1);
                    int sr = sa + half;
                    const int e = min(sa + 2 * half - 1, n);
                    int kp = 32 - __clz(sr ^ e);
                    asm volatile("" : "+v"(kp), "+v"(sr), "+v"(pending));
                    __builtin_amdgcn_sched_barrier(0);
                    // an entry whose distance no longer passes the test is dropped and the lane tries again next time
                    const bool take = !none && t_entry <= current_t;
                    cur_k = none ? K_IDLE : (take ? kp : K_POP);
                    cur_s = take ? sr : cur_s;
                    cur_id = take ? sr : cur_id;
                }
        };
        for (;;) {
            const unsigned long long tm = ballot64(cur_k >= K_POP);
            if (tm == 0) break;
            const unsigned long long ready = ballot64(p_state == P_READY);
            if (__popcll(~tm & ~ready & ~dead_mask) >= STARVE_T) break;
            if (__popcll(ballot64((p_state & 1) == 0)) >= SHADE_T) break;
            if (__popcll(~tm & ready) >= SWAP_T) {
                if ((cur_k == K_IDLE || cur_k == K_EMPTY) && p_state == P_READY) exchange(true);
            }
            iterations += EXPANSIONS;
#pragma unroll
            for (int u = 0; u < EXPANSIONS - 1; u++) { expand(); pop(); }
            expand();
            const bool at_leaf = cur_k == 0;
            const unsigned long long leaf_mask = ballot64(at_leaf);
            if (leaf_mask != 0 && (__popcll(leaf_mask) >= LEAF_T || ballot64(cur_k >= 1) == 0)) {
                if (COUNT) { sched[2]++; sched[3] += __popcll(leaf_mask); }
                if (at_leaf) {
                    const float4 *tp = reinterpret_cast<const float4 *>(&tri_sorted[cur_s]);
                    float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                    // all three loads leave together: left to itself the compiler sinks the first one behind the determinant test, and a
                    // triangle that passes it waits for memory twice
                    asm volatile("" : "+v"(t0.x), "+v"(t0.y), "+v"(t0.z));
                    if (COUNT) cnt.tri_tests++;
                    // ray_tracing.hlsl:167-189
                    const pt_f3 a = pt_mk3(t0.x, t0.y, t0.z), e_1 = pt_mk3(t1.x, t1.y, t1.z), e_2 = pt_mk3(t2.x, t2.y, t2.z);
                    const pt_f3 q = pt_cross3(rd, e_2);
                    const float a1 = pt_dot3(e_1, q);
                    {
                        // (all lanes run the whole test; a determinant within epsilon only enters the final condition)
                        const pt_f3 oa = pt_sub3(ro, a);
                        const pt_f3 s = pt_mk3(oa.x / a1, oa.y / a1, oa.z / a1);
                        const pt_f3 r = pt_cross3(s, e_1);
                        const float t = pt_dot3(e_2, r);
                        const float bx = pt_dot3(s, q);
                        const float by = pt_dot3(r, rd);
                        const float bz = 1 - bx - by;
                        const bool hit = !(fabsf(a1) <= PT_EPSILON) && bx >= -PT_EPSILON && by >= -PT_EPSILON && bz >= -PT_EPSILON &&
                                         t >= PT_EPSILON && current_t > t && t > T_MIN;
                        hit_bx = hit ? bx : hit_bx; hit_by = hit ? by : hit_by;
                        current_t = hit ? t : current_t;
                        hit_leaf = hit ? cur_s : hit_leaf;
                    }
                    cur_k = K_POP;
                }
            }
            pop();
        }
        __builtin_amdgcn_s_setprio(0);
        if (COUNT) sched[9] += (unsigned long long)((long long)__builtin_readcyclecounter() - t_trav);
    }
    flush_counters(cnt, counters, COUNT);
    if (COUNT && lane == 0) {
        for (int k = 0; k < 10; k++) atomicAdd(&counters->sched[k], sched[k]);
    }
}

// LDSN = false: index-keyed trees read as plain records from global memory, their upper levels from LDS (TopLevels)
template <bool COUNT, bool LDSN>
__global__ __launch_bounds__(1024) void k_trace_dual(SceneView sc, const TriIsect *__restrict__ tri_sorted, int levels,
                                              unsigned n_slots,
                                              const uint8_t *__restrict__ diffuse_tex,
                                              const float4 *__restrict__ normal_tex,
                                              const float4 *__restrict__ world_tex,
                                              uint32_t first_frame, uint32_t count,
                                              const typename StagingOf<LDSN>::type samples, DevCounters *__restrict__ counters,
                                              unsigned *__restrict__ work_counter,
                                              const uint32_t *__restrict__ order, uint32_t *__restrict__ pixel_work,
                                              uint32_t chunk, Tuning tune, TopLevels top, int lds_threads, float *__restrict__ xstack)
{
    // LDS as in k_trace_persistent: node records (all, or the upper levels), materials, material quotients, (the level
    // offsets of the upper levels,) one float per level and lane
    extern __shared__ __attribute__((aligned(16))) float4 lds_nodes[];
    const int n = sc.n;
    const int nthreads = blockDim.x;
    float4 *lds_mats = lds_nodes + (LDSN ? 3 * n : 3 * top.nodes);
    float4 *lds_matq = lds_mats + 3 * MAT_SLOTS;
    int *lds_off = reinterpret_cast<int *>(lds_matq + MAT_SLOTS);
    float *stack = reinterpret_cast<float *>(lds_matq + MAT_SLOTS + (LDSN ? 0 : 9)) + threadIdx.x;   // level L at stack[(L-1) * nthreads]
    if (LDSN) {
        for (int i = threadIdx.x; i < 3 * n; i += nthreads) {
            const int node = i / 3, part = i - 3 * node;
            lds_nodes[i] = reinterpret_cast<const float4 *>(&sc.wide[node])[part];
        }
    } else {
        if (threadIdx.x < 33) lds_off[threadIdx.x] = top.off[threadIdx.x];
        for (int i = threadIdx.x; i < top.nodes; i += nthreads) {
            // compact index -> (level, block) -> node id, as in k_trace_persistent
            int k = levels;
            while (k > top.k0 && i >= top.off[k - 1]) k--;
            const long long bb = i - top.off[k];
            const int id = ((bb & 1) == 0 && ((bb + 1) << k) <= (long long)n) ? (int)(((bb + 1) << k) - 1) : (int)(bb << k);
            const float4 *src = reinterpret_cast<const float4 *>(&sc.wide[id]);
            lds_nodes[3 * i] = src[0]; lds_nodes[3 * i + 1] = src[1]; lds_nodes[3 * i + 2] = src[2];
        }
    }
    if (threadIdx.x < 3 * MAT_SLOTS) lds_mats[threadIdx.x] = reinterpret_cast<const float4 *>(sc.cb->mats)[threadIdx.x];
    if (threadIdx.x < MAT_SLOTS) {
        const float4 *mp = reinterpret_cast<const float4 *>(sc.cb->mats) + 3 * threadIdx.x;
        lds_matq[threadIdx.x] = material_quotients(mp[0], mp[1], mp[2]);
    }
    __syncthreads();

    // waves beyond `lds_threads` keep their stack entries in global memory: `levels` rows of 64 floats per wave, which
    // stay in this CU's L1
    if (__builtin_amdgcn_readfirstlane((int)threadIdx.x) >= lds_threads) {
        const size_t xwave = (size_t)blockIdx.x * ((blockDim.x - lds_threads) >> 6) + ((threadIdx.x - lds_threads) >> 6);
        dual_paths<COUNT, LDSN, true>(sc, tri_sorted, levels, n_slots, diffuse_tex, normal_tex, world_tex, first_frame, count, samples, counters,
                                      work_counter, order, pixel_work, chunk, tune, top, lds_nodes, lds_mats, lds_matq, lds_off, stack, lds_threads,
                                      xstack + xwave * 64 * (size_t)levels + (threadIdx.x & 63));
    }
    else
        dual_paths<COUNT, LDSN, false>(sc, tri_sorted, levels, n_slots, diffuse_tex, normal_tex, world_tex, first_frame, count, samples, counters,
                                       work_counter, order, pixel_work, chunk, tune, top, lds_nodes, lds_mats, lds_matq, lds_off, stack, lds_threads,
                                       nullptr);
}

// ---------------------------------------------------------------------------
// Wavefront form of ray_tracing.hlsl:348-397 for scenes whose tree fits in LDS.
//
// In the single-kernel form a wave pays for the whole shading code (material
// sampling with its sine, cosine, power and square roots, path start with the
// 16-round seed hash) every time a quarter of its lanes finish a ray.  Here the
// paths of one launch are advanced bounce by bounce instead:
//   k_wf_first  : one thread per (frame, pixel): the first bounce off the G-buffer
//                 (ray_tracing.hlsl:348-370); continuing paths are appended to a queue;
//   k_wf_extend : intersect() (ray_tracing.hlsl:191-243) for every queued path, by
//                 persistent workgroups with the tree in LDS; writes one hit record
//                 per path.  A lane that finishes takes the next queued ray, which
//                 costs a 32-byte load and the root test, not a shading pass;
//   k_wf_shade  : one thread per queued path: the loop body of second_tracing
//                 (ray_tracing.hlsl:372-395); continuing paths go to the next queue.
// A path is at most 14 rays long (itr reaches MAX_ITR * 3), so a launch is the
// first-bounce kernel plus 14 extend / shade pairs; queue lengths stay on the
// device.  A path's state (origin, direction, throughput, seed, bounce count)
// travels in its 48-byte queue record, so every kernel reads and writes
// consecutive records.  Per path the sequence of random draws, tests and
// arithmetic is that of the sequential loop; which thread runs it does not
// enter the result.
//
// Queues are split into WF_SEGS segments of `seg_size` records.  The survivors of
// the inputs of segment k are appended to segment k of the next queue (they
// cannot outnumber the inputs), so every queue counter -- append positions and
// the extend kernel's read heads -- exists once per segment, each in its own
// 128-byte line: atomics on one address retire one at a time (about 13 ns each
// here), and a single counter per queue would cost more than the tracing.
// ---------------------------------------------------------------------------
constexpr int WF_SEGS = 64;             // = lanes of a wave: one lane inspects one segment
constexpr int WF_CSTRIDE = 32;          // words between counters (128 bytes)
struct PathRec {                 // 48 bytes
    float ox, oy, oz; uint32_t seed;
    float dx, dy, dz; uint32_t slot;      // slot = pixel * frames of the launch + local frame: the sample to write
    float mx, my, mz; int itr;
};
struct HitRec { float t; int leaf; float bx, by; };   // leaf key of the hit; < 0: no hit


// appends the records of the lanes with `keep` to queue[*count ...] (one atomic per wave)
__device__ __forceinline__ void wf_append(bool keep, const PathRec &rec, PathRec *__restrict__ queue,
                                          unsigned *__restrict__ count)
{
    const unsigned long long m = ballot64(keep);
    if (m == 0) return;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    unsigned base = 0;
    if (lane == __ffsll((long long)m) - 1) base = atomicAdd(count, (unsigned)__popcll(m));
    base = (unsigned)__shfl((int)base, __ffsll((long long)m) - 1, 64);
    if (keep) {
        float4 *dst = reinterpret_cast<float4 *>(&queue[base + (unsigned)__popcll(m & lt)]);
        dst[0] = make_float4(rec.ox, rec.oy, rec.oz, __uint_as_float(rec.seed));
        dst[1] = make_float4(rec.dx, rec.dy, rec.dz, __uint_as_float(rec.slot));
        dst[2] = make_float4(rec.mx, rec.my, rec.mz, __int_as_float(rec.itr));
    }
}

__global__ __launch_bounds__(256) void k_wf_first(SceneView sc, int W, int H,
                                                  const uint8_t *__restrict__ diffuse_tex,
                                                  const float4 *__restrict__ normal_tex,
                                                  const float4 *__restrict__ world_tex,
                                                  uint32_t first_frame, uint32_t count,
                                                  const uint32_t *__restrict__ seeds,
                                                  const Staging samples,
                                                  PathRec *__restrict__ queue, unsigned *__restrict__ qcount,
                                                  unsigned seg_size, DevCounters *__restrict__ counters)
{
    const unsigned npix = (unsigned)W * (unsigned)H;
    const unsigned total = npix * count;
    const float4 *mats = reinterpret_cast<const float4 *>(sc.cb->mats);
    const pt_f3 g_pos = pt_mk3(sc.cb->pos.v[0], sc.cb->pos.v[1], sc.cb->pos.v[2]);
    LaneCounters cnt;
    // whole waves iterate together so that wf_append sees every lane
    for (unsigned base = blockIdx.x * 256u; base < total; base += gridDim.x * 256u) {
        const unsigned slot = base + threadIdx.x;
        bool cont = false;
        PathRec rec = {};
        if (slot < total) {
            const unsigned px = slot / count, fl = slot - px * count;
            const float4 nrm4 = normal_tex[px];
            const int matid0 = mat_slot(__float_as_int(nrm4.w) - 1);    // asint(normal.w) - 1
            cnt.paths++;
            pt_f3 color = pt_mk3(0, 0, 0);                              // ray_tracing.hlsl:361-362
            if (__float_as_int(nrm4.w) > 0) {
                const unsigned y = px / (unsigned)W, x = px - y * (unsigned)W;
                const uchar4 dif8 = reinterpret_cast<const uchar4 *>(diffuse_tex)[px];
                const float4 pos4 = world_tex[px];
                const pt_f3 hit_pos = pt_mk3(pos4.x, pos4.y, pos4.z);
                pt_f3 multi = pt_mk3(dif8.x / 255.0f, dif8.y / 255.0f, dif8.z / 255.0f);
                if (mats[3 * matid0].y > 1) multi = pt_mk3(1, 1, 1);
                const pt_f3 view0 = pt_normalize3(pt_sub3(hit_pos, g_pos));
                const pt_f3 normal0 = pt_normalize3(pt_mk3(nrm4.x, nrm4.y, nrm4.z));
                // ray_tracing.hlsl:410: uint(launchIndex.x + launchIndex.y * viewportDims.x), in float
                const uint32_t pixel_id = (uint32_t)((float)x + (float)y * (float)W);
                uint32_t seed = rand_init(pixel_id, seeds[fl]);
                pt_f3 direction;
                if (get_bounce(mats, nullptr, first_frame + fl, view0, normal0, matid0, multi, direction, seed)) {
                    cont = true;
                    rec.ox = hit_pos.x; rec.oy = hit_pos.y; rec.oz = hit_pos.z; rec.seed = seed;
                    rec.dx = direction.x; rec.dy = direction.y; rec.dz = direction.z; rec.slot = slot;
                    rec.mx = multi.x; rec.my = multi.y; rec.mz = multi.z; rec.itr = 1;
                } else {
                    color = multi;
                }
            }
            if (!cont) put_sample(samples, slot, color.x, color.y, color.z);
        }
        const unsigned seg = base / seg_size;          // seg_size is a multiple of the block size
        wf_append(cont, rec, queue + (size_t)seg * seg_size, qcount + seg * WF_CSTRIDE);
    }
    flush_counters(cnt, counters, false);
}

__global__ __launch_bounds__(256) void k_wf_shade(SceneView sc, const PathRec *__restrict__ queue_in,
                                                  const HitRec *__restrict__ hits,
                                                  const unsigned *__restrict__ count_in,
                                                  PathRec *__restrict__ queue_out, unsigned *__restrict__ count_out,
                                                  unsigned seg_size,
                                                  const Staging samples, uint32_t first_frame, uint32_t count,
                                                  int itr_limit)
{
    // blockIdx.y = segment; the blocks of a segment stride over its records
    const unsigned seg = blockIdx.y;
    const unsigned total = count_in[seg * WF_CSTRIDE];
    const size_t seg_base = (size_t)seg * seg_size;
    const float4 *mats = reinterpret_cast<const float4 *>(sc.cb->mats);
    for (unsigned base = blockIdx.x * 256u; base < total; base += gridDim.x * 256u) {
        const unsigned i = base + threadIdx.x;
        bool cont = false;
        PathRec rec = {};
        if (i < total) {
            const float4 *src = reinterpret_cast<const float4 *>(&queue_in[seg_base + i]);
            const float4 r0 = src[0], r1 = src[1], r2 = src[2];
            const float4 h = reinterpret_cast<const float4 *>(hits)[seg_base + i];
            const int leaf = __float_as_int(h.y);                 // leaf key of the hit, -1 for a miss
            const uint32_t slot = __float_as_uint(r1.w);
            uint32_t seed = __float_as_uint(r0.w);
            pt_f3 multi = pt_mk3(r2.x, r2.y, r2.z);
            int itr = __float_as_int(r2.w);
            pt_f3 color = pt_mk3(0, 0, 0);
            if (leaf >= 0) {
                // back from intersect() with a hit (ray_tracing.hlsl:372-395)
                const pt_f3 d = pt_mk3(r1.x, r1.y, r1.z);
                const pt_f3 o = pt_add3(pt_mk3(r0.x, r0.y, r0.z), pt_scale3(d, h.x));
                itr++;
                bool ended = false;
                if (itr > MAX_ITR) {
                    const float pr = pt_max(pt_max(multi.x, multi.y), multi.z);
                    if (pr > rand_next(seed)) multi = pt_scale3(multi, 1 / pr);
                    else ended = true;
                }
                if (!ended && itr >= itr_limit) ended = true;
                if (!ended) {
                    const float4 *shp = reinterpret_cast<const float4 *>(&sc.shade[leaf]);
                    const float4 s0 = shp[0], s1 = shp[1], s2 = shp[2];
                    const float bx = h.z, by = h.w, bz = 1 - bx - by;
                    const pt_f3 A = pt_mk3(s0.x, s0.y, s0.z), B = pt_mk3(s1.x, s1.y, s1.z), C = pt_mk3(s2.x, s2.y, s2.z);
                    const pt_f3 normal = pt_add3(pt_add3(pt_scale3(A, bz), pt_scale3(B, bx)), pt_scale3(C, by));
                    const uint32_t framecount = first_frame + slot % count;
                    pt_f3 direction;
                    if (get_bounce(mats, nullptr, framecount, d, normal, __float_as_int(s0.w), multi, direction, seed)) {
                        cont = true;
                        rec.ox = o.x; rec.oy = o.y; rec.oz = o.z; rec.seed = seed;
                        rec.dx = direction.x; rec.dy = direction.y; rec.dz = direction.z; rec.slot = slot;
                        rec.mx = multi.x; rec.my = multi.y; rec.mz = multi.z; rec.itr = itr;
                    } else {
                        color = multi;
                    }
                }
            }
            if (!cont) put_sample(samples, slot, color.x, color.y, color.z);
        }
        wf_append(cont, rec, queue_out + seg_base, count_out + seg * WF_CSTRIDE);
    }
}

template <bool COUNT>
__global__ __launch_bounds__(1024) void k_wf_extend(SceneView sc, const TriIsect *__restrict__ tri_sorted, int levels,
                                                    const PathRec *__restrict__ queue,
                                                    const unsigned *__restrict__ count_in, unsigned seg_size,
                                                    HitRec *__restrict__ hits, unsigned *__restrict__ heads,
                                                    DevCounters *__restrict__ counters, Tuning tune)
{
    extern __shared__ __attribute__((aligned(16))) float4 lds_nodes[];
    const int lane = threadIdx.x & 63;
    // lane k looks at segment k; an empty queue ends the launch before the tree is loaded
    const unsigned my_seg_count = count_in[lane * WF_CSTRIDE];
    if (ballot64(my_seg_count != 0) == 0) return;
    const int n = sc.n;
    const int nthreads = blockDim.x;
    float *stack = reinterpret_cast<float *>(lds_nodes + 3 * n) + threadIdx.x;   // level L at stack[(L-1) * nthreads]
    for (int i = threadIdx.x; i < 3 * n; i += nthreads) {
        const int node = i / 3, part = i - 3 * node;
        lds_nodes[i] = reinterpret_cast<const float4 *>(&sc.wide[node])[part];
    }
    __syncthreads();

    const int REFILL_T = tune.refill_t, LEAF_T = tune.leaf_t;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const float root_mn[3] = {sc.root[0], sc.root[1], sc.root[2]}, root_mx[3] = {sc.root[3], sc.root[4], sc.root[5]};
    LaneCounters cnt;

    // wave-uniform: the segment this wave reads from and its current share of it (record indices)
    int seg = (int)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % WF_SEGS);
    unsigned q_next = 0, q_end = 0;
    bool exhausted = false;
    bool have = false;                     // a finished ray whose hit record is not written yet
    unsigned ridx = 0;
    int cur_k = K_IDLE, cur_s = 0, cur_id = 0;
    unsigned pending = 0;
    Ray ray;
    ray.o = pt_mk3(0, 0, 0); ray.d = pt_mk3(0, 0, 1); ray.t_min = PT_EPSILON; ray.t_max = 10000;
    pt_f3 inv = pt_mk3(1, 1, 1);
    float current_t = 0;
    int hit_leaf = -1;
    float hit_bx = 0, hit_by = 0;

    for (;;) {
        // ------------------------------------------------------------ refill
        const bool need = cur_k == K_IDLE && (have || !exhausted);
        const unsigned long long need_mask = ballot64(need);
        const unsigned long long trav_mask = ballot64(cur_k >= K_POP);
        if (need_mask == 0 && trav_mask == 0) break;
        if (need_mask != 0 && (__popcll(need_mask) >= REFILL_T || trav_mask == 0)) {
            if (cur_k == K_IDLE && have) {
                reinterpret_cast<float4 *>(hits)[ridx] =
                    make_float4(current_t, __int_as_float(hit_leaf), hit_bx, hit_by);
                have = false;
            }
            bool want = cur_k == K_IDLE && !exhausted;
            unsigned long long wm = ballot64(want);
            while (wm != 0) {
                if (q_next == q_end) {
                    // 64 records of the current segment; when it has none left, move to the next
                    // segment (cyclically) that still shows unread records
                    for (;;) {
                        unsigned base = 0;
                        if (lane == 0) base = atomicAdd(&heads[seg * WF_CSTRIDE], 64u);
                        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                        const unsigned seg_total = (unsigned)__shfl((int)my_seg_count, seg, 64);
                        if (base < seg_total) {
                            q_next = (unsigned)seg * seg_size + base;
                            q_end = (unsigned)seg * seg_size + min(base + 64u, seg_total);
                            break;
                        }
                        const unsigned h = __hip_atomic_load(&heads[lane * WF_CSTRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long open = ballot64(h < my_seg_count);
                        if (open == 0) { exhausted = true; break; }
                        const unsigned long long above = (seg == 63) ? 0ull : (open & (~0ull << (seg + 1)));
                        seg = __ffsll((long long)(above ? above : open)) - 1;
                    }
                    if (exhausted) break;
                }
                const unsigned rank = (unsigned)__popcll(wm & lt_mask);
                const unsigned avail = q_end - q_next;
                if (want && rank < avail) {
                    ridx = q_next + rank;
                    const float4 *src = reinterpret_cast<const float4 *>(&queue[ridx]);
                    const float4 r0 = src[0], r1 = src[1];
                    // start intersect() (ray_tracing.hlsl:191-214) with the root's own box test
                    ray.o = pt_mk3(r0.x, r0.y, r0.z);
                    ray.d = pt_mk3(r1.x, r1.y, r1.z);
                    inv = pt_mk3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
                    current_t = ray.t_max;
                    hit_leaf = -1;
                    pending = 0;
                    have = true;
                    cnt.rays++;
                    if (COUNT) { cnt.node_visits++; cnt.box_tests++; }
                    float t_entry;
                    if (slab(root_mn, root_mx, ray.o, inv, t_entry) && t_entry <= current_t) {
                        cur_k = levels; cur_s = 0; cur_id = 0;
                    }
                    want = false;
                }
                q_next += min((unsigned)__popcll(wm), avail);
                wm = ballot64(want);
            }
        }
        // --------------------------------------------------------- traversal
        // as in k_trace_persistent: EXPANSIONS node expansions per loop pass, each followed by a pop
        // attempt; loop control and the triangle-test decision once per pass
        auto expand = [&]() {
            if (cur_k >= 1) {                       // only traversing lanes have cur_k >= K_POP
                const float4 *w = lds_nodes + __umul24((unsigned)cur_id, 3u);   // 24-bit multiply: full rate
                const float4 w0 = w[0], w1 = w[1], w2 = w[2];
                const float lmn[3] = {w0.x, w0.y, w0.z}, lmx[3] = {w0.w, w1.x, w1.y};
                const float rmn[3] = {w1.z, w1.w, w2.x}, rmx[3] = {w2.y, w2.z, w2.w};
                float tl, tr;
                const bool hl = slab(lmn, lmx, ray.o, inv, tl) && tl <= current_t;
                const bool hr = slab(rmn, rmx, ray.o, inv, tr) && tr <= current_t;
                if (COUNT) { cnt.node_visits += 2; cnt.box_tests += 2; }
                const int k = cur_k, half = 1 << (k - 1);
                const int sr = cur_s + half, e = min(sr + half - 1, n);
                const int kr = 32 - __clz(sr ^ e);
                if (hl && hr) { stack[__umul24((unsigned)(k - 1), (unsigned)nthreads)] = tr; pending |= 1u << k; }
                cur_k = hl ? k - 1 : (hr ? kr : K_POP);
                cur_id = hl ? sr - 1 : sr;
                cur_s = hl ? cur_s : sr;
            }
        };
        auto pop = [&]() {
            if (cur_k == K_POP) {
                // one pending child per iteration; an entry whose entry distance no longer passes
                // the distance test is dropped and the lane tries again next iteration
                if (pending == 0) {
                    cur_k = K_IDLE;
                } else {
                    const int L = __ffs((int)pending) - 1;              // deepest pending ancestor level
                    pending &= pending - 1;
                    const float t_entry = stack[__umul24((unsigned)(L - 1), (unsigned)nthreads)];
                    if (t_entry <= current_t) {
                        const int sa = (cur_s >> L) << L, half = 1 << (L - 1);
                        const int sr = sa + half, e = min(sa + 2 * half - 1, n);
                        cur_k = 32 - __clz(sr ^ e);
                        cur_s = sr;
                        cur_id = sr;
                    }
                }
            }
        };
        for (;;) {
            if (ballot64(cur_k >= K_POP) == 0) break;
            if (__popcll(ballot64(cur_k == K_IDLE && (have || !exhausted))) >= REFILL_T) break;
#pragma unroll
            for (int u = 0; u < EXPANSIONS - 1; u++) { expand(); pop(); }
            expand();
            const bool at_leaf = cur_k == 0;
            const unsigned long long leaf_mask = ballot64(at_leaf);
            if (leaf_mask != 0 && (__popcll(leaf_mask) >= LEAF_T || ballot64(cur_k >= 1) == 0)) {
                if (at_leaf) {
                    const float4 *tp = reinterpret_cast<const float4 *>(&tri_sorted[cur_s]);
                    float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                    asm volatile("" : "+v"(t0.x), "+v"(t0.y), "+v"(t0.z));      // all three loads leave together (see k_trace_dual)
                    if (COUNT) cnt.tri_tests++;
                    // ray_tracing.hlsl:167-189
                    const pt_f3 a = pt_mk3(t0.x, t0.y, t0.z), e_1 = pt_mk3(t1.x, t1.y, t1.z), e_2 = pt_mk3(t2.x, t2.y, t2.z);
                    const pt_f3 q = pt_cross3(ray.d, e_2);
                    const float a1 = pt_dot3(e_1, q);
                    if (!(fabsf(a1) <= PT_EPSILON)) {
                        const pt_f3 oa = pt_sub3(ray.o, a);
                        const pt_f3 s = pt_mk3(oa.x / a1, oa.y / a1, oa.z / a1);
                        const pt_f3 r = pt_cross3(s, e_1);
                        const float t = pt_dot3(e_2, r);
                        const float bx = pt_dot3(s, q);
                        const float by = pt_dot3(r, ray.d);
                        const float bz = 1 - bx - by;
                        if (bx >= -PT_EPSILON && by >= -PT_EPSILON && bz >= -PT_EPSILON && t >= PT_EPSILON) {
                            if (current_t > t && t > ray.t_min) {
                                hit_bx = bx; hit_by = by;
                                current_t = t;
                                hit_leaf = cur_s;          // the shading record is in leaf order too
                            }
                        }
                    }
                    cur_k = K_POP;
                }
            }
            pop();
        }
    }
    flush_counters(cnt, counters, COUNT);
}

// Folds the samples of `count` frames (layout [pixel][frame]) into the images, frame by frame
// in order.  LITERAL: ray_tracing.hlsl:412-419 on `output` / `old` plus the UNORM8 copy;
// LINEAR: running float sum, alpha counts samples.
// A lane owns one pixel (the recurrence is sequential in the frame index), but a pixel's frames
// are 16 bytes apart while neighbouring pixels are `count`*16 bytes apart, so the wave loads
// tiles of 64 pixels x 8 frames with 8 lanes per pixel (128 contiguous bytes each) and hands
// them to the owning lanes through LDS.
constexpr int ACC_FRAMES = 8;

template <bool LINEAR>
__global__ __launch_bounds__(256) void k_accumulate(const float4 *__restrict__ samples, size_t npix,
                                                    uint32_t first_frame, uint32_t count,
                                                    float4 *__restrict__ output, uint8_t *__restrict__ display,
                                                    float4 *__restrict__ linear, int W, int4 region)
{
    __shared__ float4 tile[4][64][ACC_FRAMES + 1];        // +1: lanes of a wave land in different banks
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t p0 = (size_t)blockIdx.x * 256 + (size_t)wave * 64;       // the wave's first pixel
    const size_t p = p0 + lane;
    bool owner = p < npix;
    if (region.x >= 0) {
        // a launch over a rectangle of the image (inclusive bounds, ray_tracing.hlsl:402-406): pixels
        // outside it keep their values
        const int y = (int)(p / (size_t)W), x = (int)(p - (size_t)y * (size_t)W);
        owner = owner && x >= region.x && x <= region.z && y >= region.y && y <= region.w;
    }
    float4 acc = make_float4(0, 0, 0, 0);
    if (owner) acc = LINEAR ? linear[p] : output[p];
    const float inv_gamma = 1 / 2.2f;
    for (uint32_t k0 = 0; k0 < count; k0 += ACC_FRAMES) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int pl = (lane >> 3) + 8 * j;                           // pixel within the wave's 64
            const uint32_t k = k0 + (lane & 7);
            if (p0 + pl < npix && k < count) tile[wave][pl][lane & 7] = samples[(p0 + pl) * count + k];
        }
        __builtin_amdgcn_wave_barrier();                                   // the tile is private to the wave
        const uint32_t kn = min((uint32_t)ACC_FRAMES, count - k0);
        if (owner) {
            for (uint32_t kk = 0; kk < kn; kk++) {
                const float4 c = tile[wave][lane][kk];
                if (LINEAR) {
                    acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += 1.0f;
                } else {
                    const uint32_t framecount = first_frame + k0 + kk;
                    const float this_color[4] = {c.x, c.y, c.z, 1.0f};
                    const float old_factor = 1.0f * (float)framecount / (float)(framecount + 1);
                    const float new_factor = 1.0f / (float)(framecount + 1);
                    float o[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        o[j] = pt_powf(pt_powf(o[j], 2.2f) * old_factor + this_color[j] * new_factor, inv_gamma);
                    acc = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (!owner) return;
    if (LINEAR) {
        linear[p] = acc;
    } else {
        output[p] = acc;
        reinterpret_cast<uchar4 *>(display)[p] =
            make_uchar4((unsigned char)to_unorm8(acc.x), (unsigned char)to_unorm8(acc.y),
                        (unsigned char)to_unorm8(acc.z), (unsigned char)to_unorm8(acc.w));
    }
}

// pixel ids of the rectangle [x0, x0 + rw) x [y0, y0 + rh), row by row: the work list of a region launch
__global__ __launch_bounds__(256) void k_region_list(int W, int x0, int y0, int rw, unsigned count, uint32_t *__restrict__ out)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const unsigned ry = i / (unsigned)rw, rx = i - ry * (unsigned)rw;
    out[i] = (unsigned)(y0 + (int)ry) * (unsigned)W + (unsigned)(x0 + (int)rx);
}

// out = pow(sum.rgb / sum.a, 1/2.2), alpha 1
__global__ __launch_bounds__(256) void k_resolve(const float4 *__restrict__ linear, size_t npix,
                                                 float4 *__restrict__ output, uint8_t *__restrict__ display)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    const float4 s = linear[p];
    const float inv_n = 1.0f / s.w;
    const float inv_gamma = 1 / 2.2f;
    const float r = pt_powf(s.x * inv_n, inv_gamma), g = pt_powf(s.y * inv_n, inv_gamma), b = pt_powf(s.z * inv_n, inv_gamma);
    output[p] = make_float4(r, g, b, 1.0f);
    reinterpret_cast<uchar4 *>(display)[p] =
        make_uchar4((unsigned char)to_unorm8(r), (unsigned char)to_unorm8(g), (unsigned char)to_unorm8(b), 255);
}

struct CamBasis { float eye[3], xaxis[3], yaxis[3], zaxis[3]; float sx, sy, znear, zfar; };

// One ray per pixel centre; the direction is scaled so the ray parameter is
// view-space depth.  Writes colour bytes, normalised interpolated normal +
// asfloat(matid+1), interpolated world position (fixed_pipeline.hlsl:59-67).
__global__ __launch_bounds__(TRACE_THREADS) void k_primary(SceneView sc, CamBasis cam, int W, int H,
                                                           const uchar4 *__restrict__ albedo,
                                                           const pt_PrimitiveRender *__restrict__ prims,
                                                           uchar4 *__restrict__ diffuse_tex, float4 *__restrict__ normal_tex,
                                                           float4 *__restrict__ world_tex)
{
    extern __shared__ __attribute__((aligned(16))) uint2 lds_stack[];
    uint2 *stack = lds_stack + threadIdx.x;
    int x, y;
    lane_pixel(x, y);
    if (x >= W || y >= H) return;
    const size_t p = (size_t)y * (size_t)W + (size_t)x;
    const float ndc_x = ((float)x + 0.5f) / (float)W * 2.0f - 1.0f;
    const float ndc_y = 1.0f - ((float)y + 0.5f) / (float)H * 2.0f;
    const float vx = ndc_x * cam.sx, vy = ndc_y * cam.sy;
    const pt_f3 xa = pt_mk3(cam.xaxis[0], cam.xaxis[1], cam.xaxis[2]);
    const pt_f3 ya = pt_mk3(cam.yaxis[0], cam.yaxis[1], cam.yaxis[2]);
    const pt_f3 za = pt_mk3(cam.zaxis[0], cam.zaxis[1], cam.zaxis[2]);
    Ray ray;
    ray.o = pt_mk3(cam.eye[0], cam.eye[1], cam.eye[2]);
    ray.d = pt_add3(pt_add3(pt_scale3(xa, vx), pt_scale3(ya, vy)), za);
    ray.t_min = cam.znear;
    ray.t_max = cam.zfar;
    int prim = 0;
    pt_f3 bary = pt_mk3(0, 0, 0);
    float t;
    LaneCounters cnt;
    uchar4 dif = make_uchar4(0, 0, 0, 0);
    float4 nrm = make_float4(0, 0, 0, 0), pos = make_float4(0, 0, 0, 0);
    if (traverse<false>(sc, ray, stack, TRACE_THREADS, prim, bary, t, cnt)) {
        const pt_PrimitiveRender *pr = &prims[prim];
        const pt_f3 nA = pt_mk3(pr->vertices[0].normal.v[0], pr->vertices[0].normal.v[1], pr->vertices[0].normal.v[2]);
        const pt_f3 nB = pt_mk3(pr->vertices[1].normal.v[0], pr->vertices[1].normal.v[1], pr->vertices[1].normal.v[2]);
        const pt_f3 nC = pt_mk3(pr->vertices[2].normal.v[0], pr->vertices[2].normal.v[1], pr->vertices[2].normal.v[2]);
        const pt_f3 pA = pt_mk3(pr->vertices[0].pos.v[0], pr->vertices[0].pos.v[1], pr->vertices[0].pos.v[2]);
        const pt_f3 pB = pt_mk3(pr->vertices[1].pos.v[0], pr->vertices[1].pos.v[1], pr->vertices[1].pos.v[2]);
        const pt_f3 pC = pt_mk3(pr->vertices[2].pos.v[0], pr->vertices[2].pos.v[1], pr->vertices[2].pos.v[2]);
        const pt_f3 nn = pt_normalize3(pt_add3(pt_add3(pt_scale3(nA, bary.z), pt_scale3(nB, bary.x)), pt_scale3(nC, bary.y)));
        const pt_f3 pp = pt_add3(pt_add3(pt_scale3(pA, bary.z), pt_scale3(pB, bary.x)), pt_scale3(pC, bary.y));
        const int matid = pr->matid;
        if ((unsigned)matid < (unsigned)PT_MAX_MATERIALS) dif = albedo[matid];    // a 16-entry table
        dif.w = 255;
        nrm = make_float4(nn.x, nn.y, nn.z, __int_as_float(matid + 1));
        pos = make_float4(pp.x, pp.y, pp.z, 1.0f);
    }
    diffuse_tex[p] = dif;
    normal_tex[p] = nrm;
    world_tex[p] = pos;
}

__global__ __launch_bounds__(TRACE_THREADS) void k_intersect(SceneView sc, const float4 *__restrict__ rays, int n_rays,
                                                             pt_hit *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) uint2 lds_stack[];
    uint2 *stack = lds_stack + threadIdx.x;
    const int i = blockIdx.x * TRACE_THREADS + threadIdx.x;
    if (i >= n_rays) return;
    const float4 r0 = rays[2 * i], r1 = rays[2 * i + 1];
    Ray ray;
    ray.o = pt_mk3(r0.x, r0.y, r0.z); ray.t_min = r0.w;
    ray.d = pt_mk3(r1.x, r1.y, r1.z); ray.t_max = r1.w;
    int prim = -1;
    pt_f3 bary = pt_mk3(0, 0, 0);
    float t;
    LaneCounters cnt;
    const bool hit = traverse<false>(sc, ray, stack, TRACE_THREADS, prim, bary, t, cnt);
    pt_hit h;
    h.hit = hit ? 1 : 0;
    h.primitive = prim;
    h.t = t;
    h.bary[0] = bary.x; h.bary[1] = bary.y; h.bary[2] = bary.z;
    out[i] = h;
}

// level of internal node `id` of the tree over keys 0..n (closed form: a left child is named by its last key,
// a right child by its first key, the root is node 0)
__device__ __forceinline__ int node_level(int id, int n, int levels)
{
    if (id == 0) return levels;
    if (id & 1) return __ffs(id + 1) - 1;
    const int e = min(id + (id & -id) - 1, n);
    return 32 - __clz(id ^ e);
}

// the record slot of a LEAF child in a child line: its box twice, so that the union the traversal takes is the box itself
__device__ __forceinline__ void leaf_slot(float4 *__restrict__ d, const float *mn, const float *mx)
{
    d[0] = make_float4(mn[0], mn[1], mn[2], mx[0]);
    d[1] = make_float4(mx[1], mx[2], mn[0], mn[1]);
    d[2] = make_float4(mn[2], mx[0], mx[1], mx[2]);
}

// One 128-byte line per internal node for the traversal of trees that live in global memory: the records of the
// node's LEFT and RIGHT child (48 bytes each; a leaf child: its box twice).  The node's own record is not stored: the
// boxes of its children are the unions of the boxes in these two records.  8 float4 per node; 6..7 are unused here.
__global__ __launch_bounds__(256) void k_pair_records(const WideNode *__restrict__ wide, int n, int levels, float4 *__restrict__ pair)
{
    // two threads per node: thread 0 writes the left child's slot, thread 1 the right child's
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int id = (int)(t >> 1), half = (int)(t & 1);
    if (id >= n) return;
    const int k = node_level(id, n, levels);
    const int first = (id == 0 || !(id & 1)) ? id : id - (1 << k) + 1;          // first key of the node's range
    const int sr = first + (1 << (k - 1)), e = min(first + (1 << k) - 1, n);   // the right child's first key, the node's last
    // closed form: the left child is a leaf at level 1 and else named by its last key; the right child is a leaf when
    // its range is one key and else named by its first key
    const bool leaf = half ? (sr == e) : (k == 1);
    const int child = half ? sr : sr - 1;
    float4 *d = pair + (size_t)id * 8 + half * 3;
    if (leaf) {
        const WideNode &w = wide[id];
        if (half) leaf_slot(d, w.rmin, w.rmax); else leaf_slot(d, w.lmin, w.lmax);
    } else {
        const float4 *s = reinterpret_cast<const float4 *>(&wide[child]);
        d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
    }
}

// The same line for a tree with explicit child ids (PT_BUILD_TOPOLOGY_MORTON, n_nodes = n - 1): float4 6 holds the
// node's child ids and its left child's, float4 7 the right child's.
__global__ __launch_bounds__(256) void k_pair_records_ids(const WideNode *__restrict__ wide, int n_nodes, float4 *__restrict__ pair)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int id = (int)(t >> 1), half = (int)(t & 1);
    if (id >= n_nodes) return;
    const WideNode &w = wide[id];
    const int child = half ? w.right : w.left;
    float4 *d = pair + (size_t)id * 8;
    float4 child_ids = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (child < 0) {
        if (half) leaf_slot(d + 3, w.rmin, w.rmax); else leaf_slot(d, w.lmin, w.lmax);
    } else {
        const float4 *s = reinterpret_cast<const float4 *>(&wide[child]);
        d[half * 3 + 0] = s[0]; d[half * 3 + 1] = s[1]; d[half * 3 + 2] = s[2];
        child_ids = s[3];
    }
    if (half) d[7] = make_float4(child_ids.x, child_ids.y, 0.0f, 0.0f);
    else      d[6] = make_float4(__int_as_float(w.left), __int_as_float(w.right), child_ids.x, child_ids.y);
}

__global__ void k_root_box(const WideNode *__restrict__ wide, float *__restrict__ out6)
{   // the root's own box = union of its children's boxes (HLSL min/max)
    if (threadIdx.x < 3) {
        const WideNode &w = wide[0];
        out6[threadIdx.x] = pt_min(w.lmin[threadIdx.x], w.rmin[threadIdx.x]);
        out6[3 + threadIdx.x] = pt_max(w.lmax[threadIdx.x], w.rmax[threadIdx.x]);
    }
}

}  // namespace

static int make_view(pt_context *ctx, SceneView &sv)
{
    sv.wide = ctx->d_wide;
    sv.pair = ctx->d_pair;
    sv.tri = ctx->d_tri_sorted;
    sv.shade = ctx->d_shade;
    sv.cb = ctx->d_cb;
    sv.root = ctx->d_root;
    sv.n = ctx->n;
    return PT_OK;
}

int ptk_root_box(pt_context *ctx)
{
    hipLaunchKernelGGL(k_root_box, dim3(1), dim3(64), 0, ctx->stream, ctx->d_wide, ctx->d_root);
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

// The per-lane stack holds at most one pending right child per tree level, so
// bit_length(n) + 1 entries can never overflow.
static size_t stack_bytes(const pt_context *ctx)
{
    // a Morton-keyed tree is as high as its codes make it: one entry per internal node on the longest path
    if (ctx->build_flags & PT_BUILD_TOPOLOGY_MORTON)
        return (size_t)TRACE_THREADS * sizeof(uint2) * (size_t)(ctx->tree_height <= 16 ? 16 : (ctx->tree_height + 7) / 8 * 8);
    const int levels = 32 - __builtin_clz((unsigned)ctx->n | 1u);
    return (size_t)TRACE_THREADS * sizeof(uint2) * (size_t)(levels + 1 <= 16 ? 16 : 32);
}

// Largest workgroup (multiple of 64, at most 1024 threads) whose per-level stack fits
// beside the node records in the CU's 160 KiB of LDS; 0 when the tree does not fit.
// LDS residency pays while the stacks of at least five waves fit beside the tree: the workgroup is filled to 1024 threads with
// waves whose stacks are in global memory (3 KB per wave, which must stay in the CU's 32 KB L1 beside the triangle records).
// scene04 padded to n triangles, ms per 128 frames with the tree in LDS / read from global memory at 16 waves: n = 2800 (9 LDS
// stacks) 157 / 178, 3000 (6) 166 / 184, 3100 (4) 181 / 181, 3250 (2) 235 / 192; one-frame calls 2.41 / 3.18, 2.53 / 3.33,
// 2.53 / 3.28 ms.  (Without the extra waves the limit was eight waves: a 3000-triangle tree ran 25 % slower at 5 waves than
// from global memory at 16.)
static int lds_tree_min_threads(const pt_context *ctx) { return ctx->knobs.extra_wave > 0 ? 320 : 512; }

static int lds_resident_threads(const pt_context *ctx, int *levels_out, size_t *bytes_out)
{
    const size_t lds_total = 160 * 1024;
    const int levels = 32 - __builtin_clz((unsigned)ctx->n | 1u);
    const size_t nodes = (size_t)ctx->n * 48 + MAT_SLOTS * 64;             // + materials and their quotients
    if (nodes + (size_t)levels * 4 * 256 > lds_total) return 0;
    size_t t = (lds_total - nodes) / ((size_t)levels * 4);
    t = t > 1024 ? 1024 : (t / 64) * 64;
    *levels_out = levels;
    *bytes_out = nodes + (size_t)levels * 4 * t;
    return (int)t;
}

// Frames per kernel launch: the samples of one launch are staged as [pixel][frame] float4 (a lane writes the frames of its item side by side) and
// folded into the images by k_accumulate.  The staging buffer is capped (below), at 1 GiB (2^26 paths) for
// the wavefront schedule, whose queues take another 112 bytes per path.
constexpr size_t STAGING_BYTES_PER_PATH = sizeof(float4);

static uint32_t frames_per_launch(const pt_context *ctx, uint32_t count, bool wavefront)
{
    const size_t per_frame = (size_t)ctx->W * ctx->H * STAGING_BYTES_PER_PATH;
    // persistent schedule: up to an eighth of the device's memory (36 GiB of 288), at least 4 GiB -- at
    // 3840x2160 a 128-frame launch (17 GB of samples) is 6 % faster than four 32-frame launches
    size_t bytes = wavefront ? ((size_t)1 << 30) : (ctx->samples_cap_bytes > ((size_t)4 << 30) ? ctx->samples_cap_bytes : ((size_t)4 << 30));
    if (ctx->knobs.samples_gib > 0) bytes = (size_t)ctx->knobs.samples_gib << 30;
    size_t cap = bytes / per_frame;
    if (cap < 1) cap = 1;
    return (uint32_t)(count < cap ? count : cap);
}

// Plain-record trees in global memory (index-keyed, too small for child lines): like trees in LDS they need no buffer that
// a launch creates or resizes (child lines, deep stack entries), so short calls on them may overlap on the internal streams.
static bool uses_child_lines(const pt_context *ctx)
{
    const Knobs &kn = ctx->knobs;
    const size_t pair_max_nodes = kn.pair_max_mib > 0 ? ((size_t)kn.pair_max_mib << 20) / 128 : 0;
    return (size_t)ctx->n <= pair_max_nodes && (size_t)ctx->n * 64 >= ((size_t)kn.pair_min_mib << 20);
}

bool ptk_short_calls_may_overlap(const pt_context *ctx)
{
    if (ctx->build_flags & PT_BUILD_TOPOLOGY_MORTON) return false;
    return ptk_tree_in_lds(ctx) || !uses_child_lines(ctx);
}

// A call whose path seeds may be computed ahead (pt_context::SeedsAhead): persistent schedule, tree in LDS (trees in global
// memory keep the seed in the sample slot: the seed read brings the slot's address translation in), one launch for the call.
bool ptk_seeds_ahead_ok(const pt_context *ctx, uint32_t count)
{
    return ctx->knobs.seeds_ahead && ctx->schedule == PT_SCHEDULE_PERSISTENT && ptk_tree_in_lds(ctx) &&
           frames_per_launch(ctx, count, false) >= count;
}

int ptk_path_seeds_ahead(pt_context *ctx, hipStream_t stream, uint32_t count, const uint32_t *d_frame_seeds, uint32_t *d_path_seeds)
{
    const size_t npix = (size_t)ctx->W * ctx->H;
    int frame_lanes_log2 = 0;
    while (frame_lanes_log2 < 8 && (2u << frame_lanes_log2) <= count) frame_lanes_log2++;
    const Staging none = {nullptr};
    hipLaunchKernelGGL(k_path_seeds, dim3((unsigned)ctx->num_cus * 16u), dim3(256), 0, stream, ctx->W, (unsigned)npix, count,
                       d_frame_seeds, none, frame_lanes_log2, d_path_seeds);
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

bool ptk_tree_in_lds(const pt_context *ctx)
{
    int levels = 0;
    size_t bytes = 0;
    return !(ctx->build_flags & PT_BUILD_TOPOLOGY_MORTON) && !ctx->knobs.no_lds_tree && lds_resident_threads(ctx, &levels, &bytes) >= lds_tree_min_threads(ctx);
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel variant and context (again only
// when a later launch needs more)
template <typename F>
static int ensure_lds_limit(pt_context *ctx, int slot, F kernel, size_t lds_bytes)
{
    if (ctx->lds_attr[slot] >= (int)lds_bytes) return PT_OK;
    PT_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    ctx->lds_attr[slot] = (int)lds_bytes;
    return PT_OK;
}

static int timing_event(pt_context *ctx, hipEvent_t *ev)
{
    if (!ctx->event_pool.empty()) { *ev = ctx->event_pool.back(); ctx->event_pool.pop_back(); return PT_OK; }
    PT_HIP(ctx, hipEventCreate(ev));
    return PT_OK;
}

// region: null = the whole image, else {x0, y0, x1, y1} inclusive and inside the image
int ptk_trace_frames(pt_context *ctx, uint32_t first_frame, uint32_t count, int mode, const int *region)
{
    SceneView sv;
    if (int rc = make_view(ctx, sv)) return rc;
    const Knobs &kn = ctx->knobs;
    Tuning tune = {kn.refill_t, kn.leaf_t, ctx->itr_limit, kn.swap_t, kn.shade_t, 0, 0};
    const bool lin = (mode == PT_ACCUM_LINEAR);
    const size_t npix = (size_t)ctx->W * ctx->H;
    int levels = 0;
    size_t lds_bytes = 0;
    int threads = lds_resident_threads(ctx, &levels, &lds_bytes);
    if (kn.no_lds_tree) threads = 0;
    if (threads < lds_tree_min_threads(ctx)) threads = 0;
    if (kn.max_threads) {      // occupancy experiments: fewer waves per CU
        const int cap = (kn.max_threads / 64) * 64;
        if (cap >= 256 && threads > cap) { lds_bytes -= (size_t)levels * 4 * (size_t)(threads - cap); threads = cap; }
    }
    const bool gen = (ctx->build_flags & PT_BUILD_TOPOLOGY_MORTON) != 0;
    if (gen) threads = 0;                           // explicit child ids: the records stay in global memory
    const bool lds_nodes = threads >= 320;          // else: node records stay in global memory
    if (gen) {
        // LDS holds the materials and the first `levels` entries of every lane's (node, distance) stack; a tree higher
        // than that keeps the rest of the stacks in global memory
        threads = 1024;
        if (kn.max_threads >= 256) threads = (kn.max_threads / 64) * 64;
        const int fit = (int)((160 * 1024 - MAT_SLOTS * 64) / (8 * (size_t)threads));
        levels = ctx->tree_height < fit ? ctx->tree_height : fit;
        if (kn.gen_lds_entries >= 1 && kn.gen_lds_entries < levels) levels = kn.gen_lds_entries;
        lds_bytes = MAT_SLOTS * 64 + (size_t)levels * 8 * threads;
        const size_t deep = (size_t)(ctx->tree_height - levels) * ctx->num_cus * threads;
        if (ctx->gstack_cap < deep) {
            PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->d_gstack) (void)hipFree(ctx->d_gstack);
            ctx->d_gstack = nullptr;
            ctx->gstack_cap = 0;
            PT_HIP(ctx, hipMalloc((void **)&ctx->d_gstack, deep * sizeof(uint2)));
            ctx->gstack_cap = deep;
        }
    } else if (!lds_nodes) {
        // LDS holds the materials and one float per tree level and lane
        levels = 32 - __builtin_clz((unsigned)ctx->n | 1u);
        threads = 1024;
        if (kn.max_threads >= 256) threads = (kn.max_threads / 64) * 64;
        lds_bytes = MAT_SLOTS * 64 + (size_t)levels * 4 * threads;
    }
    // child lines take twice the memory of the plain records; measured up to 100 M triangles (12.8 GB of lines) they pay
    // at every size that does not fit in LDS (PTAMD_PAIR_MAX_MIB bounds the array; 0 = plain records)
    // ... once the plain records (64 n bytes) approach the size of the L2s (8 x 4 MiB): below about 400 k triangles the
    // records come from the caches anyway and the lines' extra instructions and registers cost 10-30 % (PTAMD_PAIR_MIN_MIB)
    const bool use_pair = !lds_nodes && uses_child_lines(ctx);
    // Index-keyed trees that do not fit in LDS as a whole and are read as plain records keep their upper levels in LDS: as
    // many whole levels, from the root down, as fit beside the stacks: 3-9 % (child ids follow from level and block, so
    // the LDS copy needs no id table; together with child lines it measured 1-5 % slower).
    TopLevels top{};
    top.k0 = 33;
    if (!lds_nodes && !gen && !use_pair && !kn.no_top_levels) {
        const size_t fixed = MAT_SLOTS * 64 + 144 + (size_t)levels * 4 * threads;
        const size_t max_nodes = fixed < 160 * 1024 ? (160 * 1024 - fixed) / 48 : 0;
        size_t acc = 0;
        int k = levels;
        for (; k >= 1; k--) {
            const long long half = 1ll << (k - 1);
            const size_t cnt = (long long)ctx->n >= half ? (size_t)((((long long)ctx->n - half) >> k) + 1) : 0;
            if (acc + cnt > max_nodes) break;
            top.off[k] = (int)acc;
            acc += cnt;
        }
        for (int j = k; j >= 0; j--) top.off[j] = (int)acc;
        top.k0 = k + 1;
        top.nodes = (int)acc;
        lds_bytes += acc * 48;
    }
    if (!lds_nodes && !gen && !use_pair) lds_bytes += 144;          // the level-offset table of that variant (present also when unused)
    if (use_pair && !ctx->pair_valid) {
        // the 128-byte child lines of the global-memory traversal, made at the first trace after a build
        // (a build alone does not pay for them)
        if (ctx->pair_cap < (size_t)ctx->n) {
            PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->d_pair) (void)hipFree(ctx->d_pair);
            ctx->d_pair = nullptr;
            ctx->pair_cap = 0;
            PT_HIP(ctx, hipMalloc((void **)&ctx->d_pair, (size_t)ctx->n * 128));
            ctx->pair_cap = (size_t)ctx->n;
        }
        const long long threads2 = 2ll * ctx->n;
        if (gen)
            hipLaunchKernelGGL(k_pair_records_ids, dim3((unsigned)((threads2 + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_wide, ctx->n - 1,
                               ctx->d_pair);
        else
            hipLaunchKernelGGL(k_pair_records, dim3((unsigned)((threads2 + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_wide, ctx->n, levels,
                               ctx->d_pair);
        PT_HIP(ctx, hipGetLastError());
        ctx->pair_valid = true;
        sv.pair = ctx->d_pair;
    }
    const bool near_first = ctx->near_first != 0;
    // the wavefront schedule exists for whole-image launches in the reference's visit order
    const bool wavefront = lds_nodes && ctx->schedule == PT_SCHEDULE_WAVEFRONT && !region && !near_first;
    const uint32_t per_launch = frames_per_launch(ctx, count, wavefront);
    if (ctx->samples_frames < per_launch) {
        PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_samples) (void)hipFree(ctx->d_samples);
        ctx->d_samples = nullptr;
        ctx->samples_frames = 0;
        PT_HIP(ctx, hipMalloc((void **)&ctx->d_samples, (size_t)per_launch * npix * STAGING_BYTES_PER_PATH));
        ctx->samples_frames = per_launch;
    }
    const Staging staging = {ctx->d_samples};
    const StagingAhead staging_ahead = {ctx->d_samples, ctx->path_seeds};
    if (wavefront && ctx->wf_paths < (size_t)per_launch * npix) {
        PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int k = 0; k < 2; k++) { if (ctx->d_wf_queue[k]) (void)hipFree(ctx->d_wf_queue[k]); ctx->d_wf_queue[k] = nullptr; }
        if (ctx->d_wf_hits) (void)hipFree(ctx->d_wf_hits);
        ctx->d_wf_hits = nullptr;
        ctx->wf_paths = 0;
        const size_t paths = (size_t)per_launch * npix;
        const size_t recs = paths + (size_t)WF_SEGS * 256;      // segments are rounded up to whole workgroups
        PT_HIP(ctx, hipMalloc((void **)&ctx->d_wf_queue[0], recs * sizeof(PathRec)));
        PT_HIP(ctx, hipMalloc((void **)&ctx->d_wf_queue[1], recs * sizeof(PathRec)));
        PT_HIP(ctx, hipMalloc((void **)&ctx->d_wf_hits, recs * sizeof(HitRec)));
        ctx->wf_paths = paths;
    }
    // a region launch walks a list of the rectangle's pixel ids (d_okeys[1]: scratch of the pixel sort otherwise)
    unsigned n_slots = (unsigned)npix;
    int4 reg = make_int4(-1, -1, -1, -1);
    if (region) {
        const int rw = region[2] - region[0] + 1, rh = region[3] - region[1] + 1;
        n_slots = (unsigned)rw * (unsigned)rh;
        reg = make_int4(region[0], region[1], region[2], region[3]);
        hipLaunchKernelGGL(k_region_list, dim3((n_slots + 255) / 256), dim3(256), 0, ctx->stream, ctx->W, region[0], region[1], rw, n_slots,
                           ctx->d_okeys[1]);
    }

    for (uint32_t off = 0; off < count; off += per_launch) {
        const uint32_t cnt = (count - off < per_launch) ? count - off : per_launch;
        const uint32_t first = first_frame + off;
        const uint32_t *seeds = ctx->d_seeds + off;
        hipEvent_t ev_begin = nullptr, ev_end = nullptr;
        if (ctx->time_trace) {
            // bounded history: a caller that never reads the times keeps the latest 4096 launches
            if (ctx->trace_events.size() >= 4096) {
                ctx->event_pool.push_back(ctx->trace_events.front().first);
                ctx->event_pool.push_back(ctx->trace_events.front().second);
                ctx->trace_events.erase(ctx->trace_events.begin());
            }
            if (int rc = timing_event(ctx, &ev_begin)) return rc;
            if (int rc = timing_event(ctx, &ev_end)) return rc;
            ctx->trace_events.emplace_back(ev_begin, ev_end);
        }
        if (wavefront) {
            // per depth and segment: queue length and the extend kernel's read head
            const size_t per_depth = (size_t)WF_SEGS * WF_CSTRIDE;
            unsigned *qcount = ctx->d_wf_counts, *qhead = ctx->d_wf_counts + 16 * per_depth;
            PT_HIP(ctx, hipMemsetAsync(ctx->d_wf_counts, 0, 32 * per_depth * sizeof(unsigned), ctx->stream));
            // segment size: the launch's paths in WF_SEGS equal parts, whole workgroups of the first kernel
            const unsigned seg_size = (unsigned)(((((size_t)cnt * npix + WF_SEGS - 1) / WF_SEGS + 255) / 256) * 256);
            PathRec *queue[2] = {static_cast<PathRec *>(ctx->d_wf_queue[0]), static_cast<PathRec *>(ctx->d_wf_queue[1])};
            HitRec *hits = static_cast<HitRec *>(ctx->d_wf_hits);
            const dim3 fgrid((unsigned)ctx->num_cus * 8u), sgrid(32, WF_SEGS);
            Tuning wt = {kn.wf_refill_t, kn.wf_leaf_t, ctx->itr_limit, kn.swap_t, kn.shade_t, 0, 0};
            if (int rc = ensure_lds_limit(ctx, 8, &k_wf_extend<false>, lds_bytes)) return rc;
            if (int rc = ensure_lds_limit(ctx, 9, &k_wf_extend<true>, lds_bytes)) return rc;
            if (ev_begin) PT_HIP(ctx, hipEventRecord(ev_begin, ctx->stream));
            hipLaunchKernelGGL(k_wf_first, fgrid, dim3(256), 0, ctx->stream, sv, ctx->W, ctx->H, ctx->d_diffuse, ctx->d_normal,
                               ctx->d_world, first, cnt, seeds, staging, queue[0], qcount + per_depth, seg_size, ctx->d_counters);
            for (int d = 1; d <= ctx->itr_limit - 1; d++) {          // a path is at most itr_limit - 1 rays long
                PathRec *qin = queue[(d - 1) & 1], *qout = queue[d & 1];
                if (ctx->detailed)
                    hipLaunchKernelGGL((k_wf_extend<true>), dim3(ctx->num_cus), dim3(threads), lds_bytes, ctx->stream, sv, ctx->d_tri_sorted,
                                       levels, qin, qcount + d * per_depth, seg_size, hits, qhead + d * per_depth, ctx->d_counters, wt);
                else
                    hipLaunchKernelGGL((k_wf_extend<false>), dim3(ctx->num_cus), dim3(threads), lds_bytes, ctx->stream, sv, ctx->d_tri_sorted,
                                       levels, qin, qcount + d * per_depth, seg_size, hits, qhead + d * per_depth, ctx->d_counters, wt);
                hipLaunchKernelGGL(k_wf_shade, sgrid, dim3(256), 0, ctx->stream, sv, qin, hits, qcount + d * per_depth, qout,
                                   qcount + (d + 1) * per_depth, seg_size, staging, first, cnt, ctx->itr_limit);
            }
            if (ev_end) PT_HIP(ctx, hipEventRecord(ev_end, ctx->stream));
            PT_HIP(ctx, hipGetLastError());
        } else {
            // frames per work item: about 16 items per lane, so that the launch ends on short
            // items, in equal parts of the launch's frames
            // overlapped short calls: several launches are in flight, each may take a part of the CUs
            unsigned grid = (unsigned)ctx->num_cus;
            if (ctx->pipe_active >= 0) {
                int g = kn.short_grid > 0 ? kn.short_grid : (kn.short_grid_div > 0 ? ctx->num_cus / kn.short_grid_div : ctx->num_cus * 7 / 16);
                grid = (unsigned)std::max(1, std::min(g, ctx->num_cus));
            }
            const unsigned long long lanes = (unsigned long long)grid * threads;
            unsigned long long chunk = ((unsigned long long)cnt * n_slots) / (lanes * 16ull);
            if (chunk < 1) chunk = 1;
            if (chunk > cnt) chunk = cnt;
            chunk = (cnt + (cnt + chunk - 1) / chunk - 1) / ((cnt + chunk - 1) / chunk);
            if (kn.chunk > 0) { chunk = (unsigned long long)kn.chunk; if (chunk > cnt) chunk = cnt; }
            // two paths per lane (k_trace_dual) for every launch length: a path carries its pixel, so a lane's two paths may be
            // two frames of one item or frames of two items (kn.dual_min_chunk > 1 keeps the single-path kernel for launches
            // whose sixteen-per-lane items would be shorter than that)
            const bool use_dual = (lds_nodes || (!gen && !use_pair)) && !near_first && kn.dual && chunk >= (unsigned long long)kn.dual_min_chunk;
            // k_trace_dual hands a launch's frames out in two tiers.  An item's end costs (its last path runs without a second
            // one to switch to, and the next item's pixel has to be fetched), so every pixel's first frames come in few long
            // items, about kn.coarse_items per lane, which for an image of several pixels per lane is ONE item per pixel; the
            // launch then ends on short items of the remaining frames, about kn.fine_items per lane (none when the long items are
            // many per lane already).  Long items need the expensive pixels first: a launch without a cost order from its
            // predecessors keeps the sixteen items per lane of the single-path kernel.  Measured on scene04, 32 / 128 frames per
            // launch, against sixteen long items per lane and a quarter of the frames in items of 8: 27.1 -> 25.8 and
            // 98.1 -> 96.3 ms at 1162x1008, 178.1 -> 169.7 ms (128 frames) at 1920x1080.
            unsigned fine_frames = 0, fine_chunk = 0;
            const bool two_tiers = use_dual && !kn.chunk && kn.fine_items > 0;
            // frames per long item, the short tier's frames and item length (fine[2]) for a number of long items per lane; returns the launch's items
            auto tiers = [&](unsigned long long coarse_items, unsigned long long &c_out, unsigned *fine) -> unsigned long long {
                const unsigned fc = kn.fine_chunk > 0 ? (unsigned)kn.fine_chunk : std::min(8u, std::max(2u, cnt / 16u));
                unsigned fr = 0;
                if ((unsigned long long)n_slots < 16ull * lanes) {
                    // short items per pixel so that a lane gets about kn.fine_items of them; at most half of the frames
                    const unsigned long long per_pixel = ((unsigned long long)kn.fine_items * lanes + n_slots - 1) / n_slots;
                    fr = (unsigned)std::min<unsigned long long>(per_pixel * fc, (cnt / 2u) / fc * fc);
                }
                if (fr < fc) fr = 0;
                const unsigned coarse = cnt - fr;
                unsigned long long c = ((unsigned long long)coarse * n_slots) / (lanes * coarse_items);
                c = std::max<unsigned long long>(c, std::max(fc + 1u, (unsigned)kn.dual_min_chunk));
                c = std::min<unsigned long long>(c, coarse);
                const unsigned long long parts = (coarse + c - 1) / c;
                c_out = (coarse + parts - 1) / parts;                // equal parts of the long tier
                fine[0] = fr; fine[1] = fr ? fc : 0u;
                return (unsigned long long)n_slots * (parts + (fr ? fr / fc : 0u));
            };
            // With few items per lane the launch ends ragged unless the expensive pixels go first; with
            // many, image order (neighbouring pixels, similar paths in a wave) is the better order.
            unsigned long long items = (unsigned long long)n_slots * ((cnt + chunk - 1) / chunk);
            unsigned long long long_chunk = chunk;
            unsigned long_fine[2] = {0, 0};
            if (two_tiers) items = tiers((unsigned long long)kn.coarse_items, long_chunk, long_fine);
            // Only for trees in LDS: when the node records come from global memory, neighbouring pixels' rays share cache lines
            // and address translations, and a queue in cost order scatters them (image order against cost order, 1920x1080:
            // 10 K triangles 8.4 against 10.3 ms per 16-frame launch, 100 K 289 against 328 ms per 128 frames, the
            // 100 M-triangle soup at 3840x2160 26.1 against 22.9 Mrays/s).
            // (an overlapped call keeps image order too: the cost keys and their sort belong to the context's stream)
            const bool ordered = lds_nodes && !region && !kn.no_order && items < 32ull * lanes && ctx->pipe_active < 0;
            if (two_tiers) {
                if (ordered && ctx->order_valid) { chunk = long_chunk; fine_frames = long_fine[0]; fine_chunk = long_fine[1]; }
                else { unsigned f2[2]; (void)tiers(16ull, chunk, f2); fine_frames = f2[0]; fine_chunk = f2[1]; }
            }
            if (!ordered && !region) ctx->order_valid = false;
            const uint32_t *order = region ? ctx->d_okeys[1] : (ctx->order_valid ? ctx->d_ovals[0] : (const uint32_t *)nullptr);
            PT_HIP(ctx, hipMemsetAsync(ctx->d_work, 0, sizeof(unsigned), ctx->stream));
            int frame_lanes_log2 = 0;                    // lanes per pixel: the largest power of two <= count, at most 256
            while (frame_lanes_log2 < 8 && (2u << frame_lanes_log2) <= cnt) frame_lanes_log2++;
            if (!ctx->path_seeds)
                hipLaunchKernelGGL(k_path_seeds, dim3((unsigned)ctx->num_cus * 16u), dim3(256), 0, ctx->stream, ctx->W, (unsigned)npix, cnt,
                               seeds, staging, frame_lanes_log2, (uint32_t *)nullptr);
            // A tree in LDS leaves stack room for `threads` lanes; the waves up to 1024 threads (scene04: one, the sixteenth)
            // run too, with their stack entries in global memory: 115 -> 109 ms per 128 frames of scene04.
            int xthreads = (lds_nodes && !kn.max_threads) ? 1024 - threads : 0;       // (PTAMD_MAX_THREADS asks for FEWER waves)
            if (xthreads > 64 * kn.extra_wave) xthreads = 64 * kn.extra_wave;
            // 16 waves x 32 levels x 64 lanes per CU cover every case of ONE launch; launches of overlapped short calls run
            // side by side (workgroup b of the next call starts on a CU while workgroup b of this one is still in its
            // tail), so the context's stream and each of the overlap slots get a region of their own
            const size_t xstack_region = (size_t)ctx->num_cus * 16 * 32 * 64;
            if (xthreads && !ctx->d_xstack)
                PT_HIP(ctx, hipMalloc((void **)&ctx->d_xstack, (PT_PIPE_SLOTS_MAX + 1) * xstack_region * sizeof(float)));
            float *const xstack = ctx->d_xstack ? ctx->d_xstack + (size_t)(ctx->pipe_active + 1) * xstack_region : nullptr;
#define PT_LAUNCH(SLOT, C, L, N) PT_LAUNCH_(SLOT, C, L, N, false, false)
#define PT_LAUNCH_G(SLOT, C, N) do { \
            if (gen) { if (use_pair) PT_LAUNCH_(SLOT + 22, C, false, N, true, true); else PT_LAUNCH_(SLOT + 16, C, false, N, false, true); } \
            else if (use_pair) PT_LAUNCH_(SLOT + 10, C, false, N, true, false); else PT_LAUNCH_(SLOT, C, false, N, false, false); } while (0)
#define PT_LAUNCH_(SLOT, C, L, N, P, G) do { \
            if (int rc_ = ensure_lds_limit(ctx, SLOT, &k_trace_persistent<C, L, N, P, G>, lds_bytes)) return rc_; \
            if (ev_begin) PT_HIP(ctx, hipEventRecord(ev_begin, ctx->stream)); \
            hipLaunchKernelGGL((k_trace_persistent<C, L, N, P, G>), dim3(grid), dim3(threads + ((L) ? xthreads : 0)), lds_bytes, ctx->stream, sv, ctx->d_tri_sorted, levels, \
                ctx->d_gstack, n_slots, ctx->d_diffuse, ctx->d_normal, ctx->d_world, first, cnt, seeds, \
                StagingOf<(L)>::pick(staging, staging_ahead), ctx->d_counters, ctx->d_work, \
                order, ordered ? ctx->d_pixel_work : (uint32_t *)nullptr, \
                (uint32_t)chunk, tune, top, threads, xstack); } while (0)
            // slot of the variant in ctx->lds_attr: COUNT, LDSN, NEAR as bits; + 10 with pair lines, + 16 / + 22 for trees with
            // explicit child ids (without / with pair lines); 8, 9: wavefront; 28, 29: the per-ray kernels
            if (use_dual) {
                // k_trace_dual: slots 30, 31 (tree in LDS), 18, 19 (plain records from global memory)
                Tuning dt = tune;
                dt.fine_frames = (int)fine_frames;
                dt.fine_chunk = (int)fine_chunk;
                dt.refill_t = kn.dual_refill_t;
                dt.leaf_t = kn.dual_leaf_t;
#define PT_LAUNCH_DUAL(SLOT, C, L) do { \
                if (int rc_ = ensure_lds_limit(ctx, SLOT, &k_trace_dual<C, L>, lds_bytes)) return rc_; \
                if (ev_begin) PT_HIP(ctx, hipEventRecord(ev_begin, ctx->stream)); \
                hipLaunchKernelGGL((k_trace_dual<C, L>), dim3(grid), dim3(threads + xthreads), lds_bytes, ctx->stream, sv, ctx->d_tri_sorted, levels, \
                    n_slots, ctx->d_diffuse, ctx->d_normal, ctx->d_world, first, cnt, StagingOf<(L)>::pick(staging, staging_ahead), ctx->d_counters, ctx->d_work, \
                    order, ordered ? ctx->d_pixel_work : (uint32_t *)nullptr, (uint32_t)chunk, dt, top, threads, xstack); } while (0)
                if (lds_nodes) { if (ctx->detailed) PT_LAUNCH_DUAL(31, true, true); else PT_LAUNCH_DUAL(30, false, true); }
                else           { if (ctx->detailed) PT_LAUNCH_DUAL(19, true, false); else PT_LAUNCH_DUAL(18, false, false); }
#undef PT_LAUNCH_DUAL
            } else if (near_first) {
                if (lds_nodes) { if (ctx->detailed) PT_LAUNCH(7, true, true, true); else PT_LAUNCH(6, false, true, true); }
                else           { if (ctx->detailed) PT_LAUNCH_G(5, true, true); else PT_LAUNCH_G(4, false, true); }
            } else {
                if (lds_nodes) { if (ctx->detailed) PT_LAUNCH(3, true, true, false); else PT_LAUNCH(2, false, true, false); }
                else           { if (ctx->detailed) PT_LAUNCH_G(1, true, false); else PT_LAUNCH_G(0, false, false); }
            }
#undef PT_LAUNCH
#undef PT_LAUNCH_G
#undef PT_LAUNCH_
            if (ev_end) PT_HIP(ctx, hipEventRecord(ev_end, ctx->stream));
            PT_HIP(ctx, hipGetLastError());
            // queue order of the next launch from the decaying per-pixel iteration counts (about
            // eight launches of at most 2^12 iterations per frame), reduced to 16 bits
            if (ordered) {
                const unsigned long long top = (unsigned long long)cnt << 15;
                int bits = 0;
                while ((top >> bits) != 0) bits++;
                if (int rc = ptk_order_pixels(ctx, bits > 16 ? bits - 16 : 0)) return rc;
            }
        }
        const dim3 agrid((unsigned)((npix + 255) / 256));
        if (ctx->pipe_active >= 0) {
            // the images are a recurrence over the calls: this call's accumulate follows the other slot's
            pt_context::PipeSlot &other = ctx->pipe[(ctx->pipe_active + kn.pipe_slots - 1) % kn.pipe_slots];
            if (other.used) PT_HIP(ctx, hipStreamWaitEvent(ctx->stream, other.done, 0));
        }
        if (lin) hipLaunchKernelGGL(k_accumulate<true>, agrid, dim3(256), 0, ctx->stream, staging.slot, npix, first, cnt,
                                    ctx->d_output, ctx->d_display, ctx->d_linear, ctx->W, reg);
        else     hipLaunchKernelGGL(k_accumulate<false>, agrid, dim3(256), 0, ctx->stream, staging.slot, npix, first, cnt,
                                    ctx->d_output, ctx->d_display, ctx->d_linear, ctx->W, reg);
        PT_HIP(ctx, hipGetLastError());
    }
    return PT_OK;
}

int ptk_resolve(pt_context *ctx)
{
    const size_t npix = (size_t)ctx->W * ctx->H;
    hipLaunchKernelGGL(k_resolve, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, ctx->stream,
                       ctx->d_linear, npix, ctx->d_output, ctx->d_display);
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

int ptk_primary(pt_context *ctx, const pt_camera *cam, const uint8_t *d_albedo)
{
    SceneView sv;
    if (int rc = make_view(ctx, sv)) return rc;
    // left-handed look-at basis and perspective scales (CS.cpp:200-201,318-329)
    const pt_f3 eye = pt_mk3(cam->eye[0], cam->eye[1], cam->eye[2]);
    const pt_f3 at = pt_mk3(cam->at[0], cam->at[1], cam->at[2]);
    const pt_f3 up = pt_mk3(cam->up[0], cam->up[1], cam->up[2]);
    const pt_f3 zaxis = pt_normalize3(pt_sub3(at, eye));
    const pt_f3 xaxis = pt_normalize3(pt_cross3(up, zaxis));
    const pt_f3 yaxis = pt_cross3(zaxis, xaxis);
    const float tan_half = tanf(cam->fovy * 0.5f);
    const float aspect = (float)ctx->W / (float)ctx->H;
    CamBasis cb;
    cb.eye[0] = eye.x; cb.eye[1] = eye.y; cb.eye[2] = eye.z;
    cb.xaxis[0] = xaxis.x; cb.xaxis[1] = xaxis.y; cb.xaxis[2] = xaxis.z;
    cb.yaxis[0] = yaxis.x; cb.yaxis[1] = yaxis.y; cb.yaxis[2] = yaxis.z;
    cb.zaxis[0] = zaxis.x; cb.zaxis[1] = zaxis.y; cb.zaxis[2] = zaxis.z;
    cb.sx = tan_half * aspect;
    cb.sy = tan_half;
    cb.znear = cam->znear;
    cb.zfar = cam->zfar;
    dim3 grid((ctx->W + 15) / 16, (ctx->H + 15) / 16), block(TRACE_THREADS);
    if (int rc = ensure_lds_limit(ctx, 28, &k_primary, stack_bytes(ctx))) return rc;
    hipLaunchKernelGGL(k_primary, grid, block, stack_bytes(ctx), ctx->stream, sv, cb, ctx->W, ctx->H,
                       reinterpret_cast<const uchar4 *>(d_albedo), ctx->d_prims,
                       reinterpret_cast<uchar4 *>(ctx->d_diffuse), ctx->d_normal, ctx->d_world);
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

int ptk_intersect(pt_context *ctx, const float *d_rays8, int n_rays, pt_hit *d_out)
{
    SceneView sv;
    if (int rc = make_view(ctx, sv)) return rc;
    if (int rc = ensure_lds_limit(ctx, 29, &k_intersect, stack_bytes(ctx))) return rc;
    hipLaunchKernelGGL(k_intersect, dim3((n_rays + TRACE_THREADS - 1) / TRACE_THREADS), dim3(TRACE_THREADS),
                       stack_bytes(ctx), ctx->stream, sv, reinterpret_cast<const float4 *>(d_rays8), n_rays, d_out);
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}
