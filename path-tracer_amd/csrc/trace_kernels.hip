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
// Execution shape: one lane per pixel, a wave covers an 8x8 pixel tile.  A lane
// runs its pixel's frames back to back (a finished path immediately starts the
// next frame's path), so lanes of a wave stay busy although path lengths
// differ.  The traversal keeps the current node in a register and only the
// pending right children (id + entry distance) on a per-lane LDS stack.
#include "pt_internal.hpp"

namespace {

constexpr int TRACE_THREADS = 256;
constexpr int POP = (int)0x80000000;

struct SceneView {
    const WideNode *__restrict__ wide;
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
// distance with the current hit distance.
__device__ __forceinline__ bool slab(const float *mn, const float *mx, const pt_f3 &o, const pt_f3 &inv, float &t_entry)
{
    float t_xmin, t_xmax, t_ymin, t_ymax, t_zmin, t_zmax;
    if (inv.x >= 0) { t_xmin = (mn[0] - o.x) * inv.x; t_xmax = (mx[0] - o.x) * inv.x; }
    else            { t_xmin = (mx[0] - o.x) * inv.x; t_xmax = (mn[0] - o.x) * inv.x; }
    if (inv.y >= 0) { t_ymin = (mn[1] - o.y) * inv.y; t_ymax = (mx[1] - o.y) * inv.y; }
    else            { t_ymin = (mx[1] - o.y) * inv.y; t_ymax = (mn[1] - o.y) * inv.y; }
    if (inv.z >= 0) { t_zmin = (mn[2] - o.z) * inv.z; t_zmax = (mx[2] - o.z) * inv.z; }
    else            { t_zmin = (mx[2] - o.z) * inv.z; t_zmax = (mn[2] - o.z) * inv.z; }
    float t_min = pt_max(t_xmin, pt_max(t_ymin, t_zmin));
    float t_max = pt_min(t_xmax, pt_min(t_ymax, t_zmax));
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
            const int prim = ~cur;
            const float4 *tp = reinterpret_cast<const float4 *>(&sc.tri[prim]);
            const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
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

// ray_tracing.hlsl:249-346
__device__ __forceinline__ bool get_bounce(const pt_Material *__restrict__ mats, uint32_t framecount,
                                           const pt_f3 view_dir, const pt_f3 normal, int id,
                                           pt_f3 &multi, pt_f3 &direction, uint32_t &seed)
{
    direction = pt_mk3(0, 0, 0);
    const float4 *mp = reinterpret_cast<const float4 *>(&mats[id]);
    const float4 m0 = mp[0], m1 = mp[1], m2 = mp[2];
    const float ke = m0.x, ns = m0.y, tr = m0.z, ni = m0.w;
    const float rand_mat = rand_next(seed);
    const float base = tr + m2.x + m1.x;
    if (ke > 0) {
        multi = pt_mul3(multi, pt_mk3(ke, ke, ke));
        return false;
    }
    if (tr > 0) {
        const float cosv = pt_dot3(normal, view_dir);
        const float nc = 1.f, nt = ni;
        float nnt = nc / nt;
        if (cosv > 0) nnt = nt / nc;
        const float ddn = fabsf(cosv);
        const float cos2t = 1.f - nnt * nnt * (1.f - ddn * ddn);
        if (cos2t < 0.0f) {
            direction = pt_reflect3(view_dir, normal);
            return true;
        }
        if (cosv < 0) {
            const float k = cosv * nnt + sqrtf(cos2t);
            direction = pt_normalize3(pt_sub3(pt_scale3(view_dir, nnt), pt_scale3(normal, k)));
        } else {
            const float k = -cosv * nnt + sqrtf(cos2t);
            direction = pt_normalize3(pt_add3(pt_scale3(view_dir, nnt), pt_scale3(normal, k)));
        }
        const float a = nt - nc, b = nt + nc;
        const float R0 = a * a / (b * b);
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
    if (ns > 1 && rand_mat < m2.x / base) {
        direction = pt_reflect3(view_dir, normal);
        const float r1 = 2 * PT_MY_PI * ((framecount & 1) / 2.0f + rand_next(seed) / 2);
        const float height = pt_powf((((framecount >> 1) & 3) / 4.0f + rand_next(seed) / 4), 1 / ns);
        const float r2s = sqrtf(1 - height * height);
        const pt_f3 w = direction;
        const pt_f3 u = pt_normalize3(pt_cross3(w, view_dir));
        const pt_f3 v = pt_cross3(w, u);
        const pt_f3 d = pt_add3(pt_add3(pt_scale3(pt_scale3(u, pt_cosf(r1)), r2s),
                                        pt_scale3(pt_scale3(v, pt_sinf(r1)), r2s)),
                                pt_scale3(w, height));
        direction = pt_normalize3(d);
        multi = pt_mul3(multi, pt_mk3(m2.x, m2.y, m2.z));
    } else {
        multi = pt_mul3(multi, pt_mk3(m1.x, m1.y, m1.z));
        const float r1 = 2 * PT_MY_PI * (((framecount >> 1) & 3) / 4.0f + rand_next(seed) / 4.0f);
        const float r2 = (((framecount >> 3) & 3) / 4.0f + rand_next(seed) / 4.0f);
        const float r2s = sqrtf(r2);
        const pt_f3 w = normal;
        pt_f3 u;
        if (fabsf(w.x) > 0.1f) u = pt_cross3(w, pt_mk3(0, 1, 0));
        else u = pt_cross3(w, pt_mk3(1, 0, 0));
        u = pt_normalize3(u);
        const pt_f3 v = pt_cross3(w, u);
        const pt_f3 d = pt_add3(pt_add3(pt_scale3(pt_scale3(u, pt_cosf(r1)), r2s),
                                        pt_scale3(pt_scale3(v, pt_sinf(r1)), r2s)),
                                pt_scale3(w, sqrtf(1 - r2)));
        direction = pt_normalize3(d);
    }
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

// ray_tracing.hlsl:348-420 for `count` consecutive frames of one pixel.
template <bool COUNT, bool LINEAR>
__global__ __launch_bounds__(TRACE_THREADS) void k_trace(SceneView sc, int W, int H,
                                                         const uint8_t *__restrict__ diffuse_tex,
                                                         const float4 *__restrict__ normal_tex,
                                                         const float4 *__restrict__ world_tex,
                                                         uint32_t first_frame, uint32_t count,
                                                         const uint32_t *__restrict__ seeds,
                                                         float4 *__restrict__ output, uint8_t *__restrict__ display,
                                                         float4 *__restrict__ linear, DevCounters *__restrict__ counters)
{
    extern __shared__ __attribute__((aligned(16))) uint2 lds_stack[];
    uint2 *stack = lds_stack + threadIdx.x;
    const int stride = TRACE_THREADS;

    int x, y;
    lane_pixel(x, y);
    LaneCounters cnt;
    if (x < W && y < H) {
        const size_t p = (size_t)y * (size_t)W + (size_t)x;
        const uchar4 dif8 = reinterpret_cast<const uchar4 *>(diffuse_tex)[p];
        const float4 nrm4 = normal_tex[p];
        const float4 pos4 = world_tex[p];
        const pt_Material *mats = sc.cb->mats;
        const pt_f3 g_pos = pt_mk3(sc.cb->pos.v[0], sc.cb->pos.v[1], sc.cb->pos.v[2]);

        const int matid0 = __float_as_int(nrm4.w) - 1;              // asint(normal.w) - 1
        const bool background = (matid0 < 0);
        const pt_f3 hit_pos = pt_mk3(pos4.x, pos4.y, pos4.z);
        pt_f3 multi0 = pt_mk3(dif8.x / 255.0f, dif8.y / 255.0f, dif8.z / 255.0f);
        pt_f3 view0 = pt_mk3(0, 0, 0), normal0 = pt_mk3(0, 0, 0);
        if (!background) {
            if (mats[matid0].ns > 1) multi0 = pt_mk3(1, 1, 1);
            view0 = pt_normalize3(pt_sub3(hit_pos, g_pos));
            normal0 = pt_normalize3(pt_mk3(nrm4.x, nrm4.y, nrm4.z));
        }
        // ray_tracing.hlsl:410: uint(launchIndex.x + launchIndex.y * viewportDims.x), in float
        const uint32_t pixel_id = (uint32_t)((float)(uint32_t)x + (float)(uint32_t)y * (float)W);

        float4 acc = LINEAR ? linear[p] : output[p];
        const float inv_gamma = 1 / 2.2f;

        for (uint32_t f = 0; f < count; f++) {
            const uint32_t framecount = first_frame + f;
            pt_f3 color = pt_mk3(0, 0, 0);
            cnt.paths++;
            if (!background) {
                uint32_t seed = rand_init(pixel_id, seeds[f]);
                pt_f3 multi = multi0, direction;
                if (!get_bounce(mats, framecount, view0, normal0, matid0, multi, direction, seed)) {
                    color = multi;
                } else {
                    Ray ray;
                    ray.t_min = PT_EPSILON;
                    ray.t_max = 10000;
                    ray.o = hit_pos;
                    ray.d = direction;
                    int itr = 1, prim = 0;
                    pt_f3 bary = pt_mk3(0, 0, 0);
                    float t;
                    while (traverse<COUNT>(sc, ray, stack, stride, prim, bary, t, cnt)) {
                        ray.o = pt_add3(ray.o, pt_scale3(ray.d, t));
                        itr++;
                        const float4 *sp = reinterpret_cast<const float4 *>(&sc.shade[prim]);
                        const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2];
                        const int matid = __float_as_int(s0.w);
                        if (itr > MAX_ITR) {
                            const float pr = pt_max(pt_max(multi.x, multi.y), multi.z);
                            if (pr > rand_next(seed)) multi = pt_scale3(multi, 1 / pr);
                            else break;
                        }
                        if (itr >= MAX_ITR * 3) break;
                        const pt_f3 A = pt_mk3(s0.x, s0.y, s0.z), B = pt_mk3(s1.x, s1.y, s1.z), C = pt_mk3(s2.x, s2.y, s2.z);
                        const pt_f3 new_normal = pt_add3(pt_add3(pt_scale3(A, bary.z), pt_scale3(B, bary.x)), pt_scale3(C, bary.y));
                        if (!get_bounce(mats, framecount, ray.d, new_normal, matid, multi, direction, seed)) {
                            color = multi;
                            break;
                        }
                        ray.d = direction;
                    }
                }
            }
            if (LINEAR) {
                acc.x += color.x; acc.y += color.y; acc.z += color.z; acc.w += 1.0f;
            } else {
                // ray_tracing.hlsl:412-419
                const float this_color[4] = {color.x, color.y, color.z, 1.0f};
                const float old_factor = 1.0f * (float)framecount / (float)(framecount + 1);
                const float new_factor = 1.0f / (float)(framecount + 1);
                float o[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
                for (int k = 0; k < 4; k++)
                    o[k] = pt_powf(pt_powf(o[k], 2.2f) * old_factor + this_color[k] * new_factor, inv_gamma);
                acc = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
        if (LINEAR) {
            linear[p] = acc;
        } else {
            output[p] = acc;
            reinterpret_cast<uchar4 *>(display)[p] =
                make_uchar4((unsigned char)to_unorm8(acc.x), (unsigned char)to_unorm8(acc.y),
                            (unsigned char)to_unorm8(acc.z), (unsigned char)to_unorm8(acc.w));
        }
    }
    flush_counters(cnt, counters, COUNT);
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
        dif = albedo[matid];
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
    sv.tri = ctx->d_tri;
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
    const int levels = 32 - __builtin_clz((unsigned)ctx->n | 1u);
    return (size_t)TRACE_THREADS * sizeof(uint2) * (size_t)(levels + 1 <= 16 ? 16 : 32);
}

int ptk_trace_frames(pt_context *ctx, uint32_t first_frame, uint32_t count, int mode)
{
    SceneView sv;
    if (int rc = make_view(ctx, sv)) return rc;
    dim3 grid((ctx->W + 15) / 16, (ctx->H + 15) / 16), block(TRACE_THREADS);
    const size_t lds = stack_bytes(ctx);
#define PT_LAUNCH_TRACE(C, L) hipLaunchKernelGGL((k_trace<C, L>), grid, block, lds, ctx->stream, sv, ctx->W, ctx->H, \
        ctx->d_diffuse, ctx->d_normal, ctx->d_world, first_frame, count, ctx->d_seeds, \
        ctx->d_output, ctx->d_display, ctx->d_linear, ctx->d_counters)
    const bool lin = (mode == PT_ACCUM_LINEAR);
    if (ctx->detailed) { if (lin) PT_LAUNCH_TRACE(true, true); else PT_LAUNCH_TRACE(true, false); }
    else               { if (lin) PT_LAUNCH_TRACE(false, true); else PT_LAUNCH_TRACE(false, false); }
#undef PT_LAUNCH_TRACE
    PT_HIP(ctx, hipGetLastError());
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
    hipLaunchKernelGGL(k_intersect, dim3((n_rays + TRACE_THREADS - 1) / TRACE_THREADS), dim3(TRACE_THREADS),
                       stack_bytes(ctx), ctx->stream, sv, reinterpret_cast<const float4 *>(d_rays8), n_rays, d_out);
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}
