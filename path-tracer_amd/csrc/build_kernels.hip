// Acceleration-structure build on gfx950: per-triangle boxes + scene box,
// Morton codes, stable LSD radix sort, binary-radix-tree links, bottom-up
// boxes, triangle records for traversal.
//
// Follows (reference): ray_tracer.cpp:41-66 (boxes, scene box, codes, sort,
// strip), structs.hpp:81-116 (genMortonCode, mergeBound), radix_build_cs.hlsl
// (tree links), bvh_build_cs.hlsl (leaf init + box merge).
//
// All stages are HBM-streaming integer / min-max work: no matrix cores.  Wave
// size is 64 throughout.  Compile with -ffp-contract=off (Morton arithmetic
// must not be fused).
#include "pt_internal.hpp"

#include <algorithm>

namespace {

// ---------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned enc_f(float f)
{   // order-preserving float -> uint (for atomicMin/atomicMax)
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_f(unsigned u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

typedef float f4v __attribute__((ext_vector_type(4)));
// streaming store of 16 bytes that nothing in this kernel reads again
__device__ __forceinline__ void nt_store4(float4 *p, float a, float b, float c, float d)
{
    f4v v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<f4v *>(p));
}

__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------------------
// stage 1: per-triangle boxes + scene box          (ray_tracer.cpp:41-54)
// A workgroup stages 256 consecutive 80-byte records through LDS with 16-byte
// coalesced loads, then each lane reads its own triangle.
// ---------------------------------------------------------------------------
constexpr int PRIM_DW = 20;   // dwords per PrimitiveRender

// MIDS: the default build of the index-keyed tree needs of a box only its centre, for the Morton code (its refit fetches
// the primitive itself): 12 bytes per triangle are written in the boxes' place, (min + max) / 2 as structs.hpp:90-92 forms it.
template <bool MIDS>
__global__ __launch_bounds__(256) void k_bounds(const uint4 *__restrict__ prims16, int n,
                                                BoxRec *__restrict__ boxes, float *__restrict__ partial)
{
    __shared__ uint4 stage[256 * PRIM_DW / 4];      // 20480 B
    __shared__ float red[6][4];
    const int t = threadIdx.x;
    const long long first = (long long)blockIdx.x * 256;
    const int cnt = (int)min((long long)256, (long long)n - first);
    const int n16 = cnt * (PRIM_DW / 4);
    const uint4 *src = prims16 + first * (PRIM_DW / 4);
    for (int i = t; i < n16; i += 256) stage[i] = src[i];
    __syncthreads();

    float mn[3], mx[3];
    if (t < cnt) {
        const float *p = reinterpret_cast<const float *>(stage) + t * PRIM_DW;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float x = p[k], y = p[6 + k], z = p[12 + k];
            // math3d.h:13-23: max(x, max(y, z)) / min(x, min(y, z))
            float yz_max = (y > z) ? y : z, yz_min = (y < z) ? y : z;
            mx[k] = (x > yz_max) ? x : yz_max;
            mn[k] = (x < yz_min) ? x : yz_min;
        }
        if (MIDS) {
            float *dst = reinterpret_cast<float *>(boxes) + 3 * (first + t);
#pragma unroll
            for (int k = 0; k < 3; k++) dst[k] = (mn[k] + mx[k]) / 2;
        } else {
            float4 *dst = reinterpret_cast<float4 *>(&boxes[first + t]);
            nt_store4(&dst[0], mn[0], mn[1], mn[2], 0.0f);
            nt_store4(&dst[1], mx[0], mx[1], mx[2], __int_as_float((int32_t)(first + t)));
        }
    } else {
        const float inf = __uint_as_float(0x7f800000u);
#pragma unroll
        for (int k = 0; k < 3; k++) { mn[k] = inf; mx[k] = -inf; }
    }
    const int wave = t >> 6, lane = t & 63;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float a = wave_min(mn[k]), b = wave_max(mx[k]);
        if (lane == 0) { red[k][wave] = a; red[3 + k][wave] = b; }
    }
    __syncthreads();
    // one partial box per workgroup; a second, single-workgroup kernel merges them (tens of
    // thousands of workgroups hammering six atomics would serialise at the memory side)
    if (t < 3) partial[(size_t)blockIdx.x * 6 + t] = fminf(fminf(red[t][0], red[t][1]), fminf(red[t][2], red[t][3]));
    else if (t < 6) partial[(size_t)blockIdx.x * 6 + t] = fmaxf(fmaxf(red[t][0], red[t][1]), fmaxf(red[t][2], red[t][3]));
}

__global__ __launch_bounds__(1024) void k_bbox_reduce(const float *__restrict__ partial, int nparts, unsigned *__restrict__ bbox_enc)
{
    __shared__ float red[6][16];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const float inf = __uint_as_float(0x7f800000u);
    float v[6] = {inf, inf, inf, -inf, -inf, -inf};
    for (int i = t; i < nparts; i += 1024) {
#pragma unroll
        for (int k = 0; k < 3; k++) { v[k] = fminf(v[k], partial[(size_t)i * 6 + k]); v[3 + k] = fmaxf(v[3 + k], partial[(size_t)i * 6 + 3 + k]); }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float a = wave_min(v[k]), b = wave_max(v[3 + k]);
        if (lane == 0) { red[k][wave] = a; red[3 + k][wave] = b; }
    }
    __syncthreads();
    if (t < 6) {
        float r = red[t][0];
        for (int w = 1; w < 16; w++) r = (t < 3) ? fminf(r, red[t][w]) : fmaxf(r, red[t][w]);
        bbox_enc[t] = enc_f(r);
    }
}

// ---------------------------------------------------------------------------
// stage 2: Morton codes                                 (structs.hpp:81-105)
// Strict IEEE evaluation of ((mid - min) / (max - min)) * L per axis, then bit
// interleave with x in the highest of each triple.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t spread3(uint64_t v)
{   // bit i of v (i < 21) -> bit 3*i
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

template <typename K, bool MIDS>
__device__ __forceinline__ K morton_code(const BoxRec *__restrict__ boxes, long long i, const unsigned *__restrict__ bbox_enc, float cells)
{
    float midx, midy, midz;
    if (MIDS) {
        const float *src = reinterpret_cast<const float *>(boxes) + 3 * (size_t)i;
        midx = src[0]; midy = src[1]; midz = src[2];
    } else {
        const float4 *src = reinterpret_cast<const float4 *>(&boxes[i]);
        const float4 lo = src[0], hi = src[1];
        midx = (lo.x + hi.x) / 2;
        midy = (lo.y + hi.y) / 2;
        midz = (lo.z + hi.z) / 2;
    }
    float min_x = dec_f(bbox_enc[0]), min_y = dec_f(bbox_enc[1]), min_z = dec_f(bbox_enc[2]);
    float max_x = dec_f(bbox_enc[3]), max_y = dec_f(bbox_enc[4]), max_z = dec_f(bbox_enc[5]);
    // a flat axis of the scene box gives 0/0 = NaN: cell 0 (what the x86 conversion of the reference
    // leaves in the bits it interleaves)
    const float qa = ((midx - min_x) / (max_x - min_x)) * cells;
    const float qb = ((midy - min_y) / (max_y - min_y)) * cells;
    const float qc = ((midz - min_z) / (max_z - min_z)) * cells;
    uint64_t a = (qa != qa) ? 0 : (uint64_t)qa;
    uint64_t b = (qb != qb) ? 0 : (uint64_t)qb;
    uint64_t c = (qc != qc) ? 0 : (uint64_t)qc;
    return (K)((spread3(a) << 2) | (spread3(b) << 1) | spread3(c));
}

template <typename K, bool MIDS>
__global__ __launch_bounds__(256) void k_morton(const BoxRec *__restrict__ boxes, int n,
                                                const unsigned *__restrict__ bbox_enc, float cells,
                                                K *__restrict__ codes)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) codes[i] = morton_code<K, MIDS>(boxes, i, bbox_enc, cells);
}

// The same codes, one sorting tile (PT_SORT_TILE keys) per workgroup, together with the tile's counts of the lowest digit:
// what k_sort_hist would count for the first radix pass by reading the codes again (block_hist is digit-major, as there).
template <typename K, bool MIDS>
__global__ __launch_bounds__(PT_SORT_THREADS) void k_morton_hist(const BoxRec *__restrict__ boxes, int n,
                                                                 const unsigned *__restrict__ bbox_enc, float cells,
                                                                 K *__restrict__ codes, uint32_t *__restrict__ block_hist, int nblocks)
{
    __shared__ uint32_t h[256];
    const int t = threadIdx.x;
    h[t] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * PT_SORT_TILE;
#pragma unroll 4
    for (int r = 0; r < PT_SORT_ITEMS; r++) {
        const long long i = base + r * PT_SORT_THREADS + t;
        if (i < n) {
            const K code = morton_code<K, MIDS>(boxes, i, bbox_enc, cells);
            codes[i] = code;
            atomicAdd(&h[(unsigned)code & 255u], 1u);
        }
    }
    __syncthreads();
    block_hist[(size_t)t * nblocks + blockIdx.x] = h[t];
}

// ---------------------------------------------------------------------------
// stage 3: stable LSD radix sort, 8-bit digits, (32- or 64-bit key, 32-bit index);
// the last pass leaves only the permutation (nothing reads the sorted keys)
//   per pass: k_sort_hist (per-workgroup digit counts, LDS histogram)
//             k_sort_scan (one workgroup per digit scans its row of counts)
//             k_sort_scatter (wave64 ballot match -> stable ranks, LDS-staged
//                             reorder, coalesced runs out)
// Sorts of more than two passes (full-resolution codes) read their keys once per
// pass instead of twice: k_sort_digit_counts counts every digit of every pass in
// ONE sweep over the unsorted keys (the counts of a digit do not depend on the
// order of the keys), and each pass is a single k_sort_scatter<CHAINED> launch
// whose workgroups find their per-digit offsets by looking back over the tiles
// before them (one status word per tile and digit) -- see chain_lookback.
// ---------------------------------------------------------------------------
template <typename K>
__global__ __launch_bounds__(PT_SORT_THREADS) void k_sort_hist(const K *__restrict__ keys, int n, int shift,
                                                               uint32_t *__restrict__ block_hist, int nblocks)
{
    __shared__ uint32_t h[256];
    const int t = threadIdx.x;
    h[t] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * PT_SORT_TILE;
#pragma unroll 4
    for (int i = t; i < PT_SORT_TILE; i += PT_SORT_THREADS) {
        long long g = base + i;
        if (g < n) atomicAdd(&h[(unsigned)(keys[g] >> shift) & 255u], 1u);
    }
    __syncthreads();
    block_hist[(size_t)t * nblocks + blockIdx.x] = h[t];
}

__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t x, uint32_t *sh /*[5]*/, uint32_t *total)
{   // exclusive scan over the 256 threads of the workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t v = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    __syncthreads();                 // protects sh[] against the previous use
    if (lane == 63) sh[wave] = v;
    __syncthreads();
    uint32_t w0 = sh[0], w1 = sh[1], w2 = sh[2], w3 = sh[3];
    uint32_t off = (wave > 0 ? w0 : 0) + (wave > 1 ? w1 : 0) + (wave > 2 ? w2 : 0);
    *total = w0 + w1 + w2 + w3;
    return off + v - x;
}

// workgroup d turns row d (the counts of digit d per sorting workgroup) into exclusive
// offsets within the digit and leaves the digit's total in digit_total[d]
__global__ __launch_bounds__(256) void k_sort_scan(uint32_t *__restrict__ block_hist,
                                                   uint32_t *__restrict__ digit_total, int nblocks)
{
    __shared__ uint32_t sh[5];
    const int d = blockIdx.x, t = threadIdx.x;
    uint32_t carry = 0;
    uint32_t *row = block_hist + (size_t)d * nblocks;
    for (int c = 0; c < nblocks; c += 256) {
        int i = c + t;
        uint32_t x = (i < nblocks) ? row[i] : 0u;
        uint32_t sum;
        uint32_t ex = block_excl_scan_256(x, sh, &sum);
        if (i < nblocks) row[i] = carry + ex;
        carry += sum;
    }
    if (t == 0) digit_total[d] = carry;
}

// All PASSES digit histograms of the keys in one sweep: per-workgroup counts in LDS, then one
// global add per non-empty (pass, digit).  digit_total[PASSES][256] starts at zero.
template <typename K, int PASSES>
__global__ __launch_bounds__(256) void k_sort_digit_counts(const K *__restrict__ keys, int n, uint32_t *__restrict__ digit_total)
{
    __shared__ uint32_t h[PASSES * 256];
    const int t = threadIdx.x;
    for (int i = t; i < PASSES * 256; i += 256) h[i] = 0;
    __syncthreads();
    const long long stride = (long long)gridDim.x * 1024;
    for (long long base = (long long)blockIdx.x * 1024; base < n; base += stride) {
        K k[4];
        bool valid[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {                   // four loads in flight
            const long long g = base + r * 256 + t;
            valid[r] = g < n;
            k[r] = valid[r] ? keys[g] : (K)0;
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (valid[r]) {
                const unsigned active = (unsigned)__popcll(__ballot(1));
                const bool first_lane = (t & 63) == __builtin_amdgcn_readfirstlane(t & 63);
#pragma unroll
                for (int p = 0; p < PASSES; p++) {
                    const unsigned d = (unsigned)(k[r] >> (8 * p)) & 255u;
                    // a digit that all lanes share (the empty top bytes of short codes) is one add, not 64 serialised ones
                    const unsigned d0 = (unsigned)__builtin_amdgcn_readfirstlane((int)d);
                    if (__all(d == d0)) { if (first_lane) atomicAdd(&h[p * 256 + d0], active); }
                    else atomicAdd(&h[p * 256 + d], 1u);
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PASSES; p++) {
        const uint32_t c = h[p * 256 + t];
        if (c) atomicAdd(&digit_total[p * 256 + t], c);
    }
}

// Status word of (tile, digit) in a chained pass: bits 31..30 = 0 nothing yet, CH_AGG: bits 29..0 hold the tile's own count
// of the digit, CH_PREFIX: they hold the count of this tile and all tiles before it.  One aligned 32-bit word written by ONE
// agent-scope store and polled by agent-scope loads (both bypass the caches that other workgroups' stores do not refresh), so
// the word is its own flag and no fence is needed.  Counts stay below 2^30 (the host checks n).
constexpr uint32_t CH_AGG = 1u << 30, CH_PREFIX = 2u << 30, CH_MASK = CH_AGG - 1;
constexpr unsigned CH_SPIN_LIMIT = 1u << 22;            // polls of one word before a workgroup gives up (seconds)
constexpr int CH_TICKET_WORD = 8 * 256 + 16;            // in digit_total: one tile counter per pass
constexpr int CH_ERROR_WORD = 8 * 256 + 32;             // in digit_total: set when a workgroup gave up
__device__ __forceinline__ uint32_t chain_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void chain_store(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Number of keys with digit t in the tiles before `tile`: sums the counts of the preceding tiles, nearest first, until one
// of them already knows its own prefix.  Tiles are numbered in the order in which workgroups START (a ticket counter), so
// every tile before this one belongs to a workgroup that is running or has finished, and a tile publishes its own count
// before it looks back: the wait always ends.  Four status words are fetched per round trip.  A workgroup that polls one
// word CH_SPIN_LIMIT times sets the error word and goes on with what it has (positions then stay inside the output,
// because a partial sum is never larger than the whole); the host reports the build as failed.
__device__ __forceinline__ uint32_t chain_lookback(uint32_t *status, uint32_t tile, int t, uint32_t *error)
{
    uint32_t excl = 0;
    long long j = (long long)tile - 1;
    unsigned spins = 0;
    while (j >= 0) {
        uint32_t x[4];
#pragma unroll
        for (int k = 0; k < 4; k++) x[k] = (j - k >= 0) ? chain_load(status + (size_t)(j - k) * 256 + t) : CH_PREFIX;
        int used = 0;
        bool done = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t flag = x[k] >> 30;
            if (!done && used == k && flag != 0) { excl += x[k] & CH_MASK; used = k + 1; done = (flag == 2u); }
        }
        if (done) break;
        j -= used;
        if (used == 0) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > CH_SPIN_LIMIT) { atomicOr(error, 1u); break; }
        }
    }
    return excl;
}

// CHAINED: `status` = one word per (tile, digit), all zero at launch; chain_ticket = the pass's tile counter (zero at
// launch); digit_total = the digit counts of this pass (k_sort_digit_counts).  Otherwise block_off / digit_total come from
// k_sort_hist / k_sort_scan and the tile is the workgroup's index.
template <typename K, bool IOTA, bool KEYS_OUT, bool CHAINED>
__global__ __launch_bounds__(PT_SORT_THREADS) void k_sort_scatter(
    const K *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
    K *__restrict__ keys_out, uint32_t *__restrict__ vals_out, int n, int shift,
    const uint32_t *__restrict__ block_off, const uint32_t *__restrict__ digit_total, int nblocks,
    uint32_t *status, uint32_t *chain_ticket, uint32_t *chain_error)
{
    __shared__ uint32_t digit_gbase[256];               // first output slot of each digit
    // 64-bit keys: keys and values go through the SAME 32 KB in turn (39 KB per workgroup, four workgroups per CU
    // instead of two); 32-bit keys: 16 KB each, side by side
    constexpr bool TWO_ROUNDS = sizeof(K) == 8;
    __shared__ K skeys[PT_SORT_TILE];
    __shared__ uint32_t svals_own[TWO_ROUNDS ? 1 : PT_SORT_TILE];
    uint32_t *svals = TWO_ROUNDS ? reinterpret_cast<uint32_t *>(skeys) : svals_own;
    __shared__ uint32_t wave_hist[4][256];              //  4 KB
    __shared__ uint32_t digit_base[256];
    __shared__ uint32_t tile_off[CHAINED ? 256 : 1];    // CHAINED: keys of the digit in the tiles before this one
    __shared__ uint32_t sh[5];
    __shared__ uint32_t s_tile;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (CHAINED && t == 0) s_tile = atomicAdd(chain_ticket, 1u);
    for (int i = t; i < 4 * 256; i += PT_SORT_THREADS) (&wave_hist[0][0])[i] = 0;
    __syncthreads();
    const uint32_t tile = CHAINED ? s_tile : blockIdx.x;

    const long long tile_base = (long long)tile * PT_SORT_TILE;
    const long long wave_base = tile_base + (long long)wave * (PT_SORT_ITEMS * 64);
    const int tile_cnt = (int)min((long long)PT_SORT_TILE, (long long)n - tile_base);

    K key[PT_SORT_ITEMS];
    uint32_t val[PT_SORT_ITEMS];
    uint32_t rank[PT_SORT_ITEMS];
#pragma unroll
    for (int r = 0; r < PT_SORT_ITEMS; r++) {
        long long g = wave_base + r * 64 + lane;
        bool valid = g < n;
        key[r] = valid ? keys_in[g] : (K)~(K)0;
        val[r] = IOTA ? (uint32_t)g : (valid ? vals_in[g] : 0u);
    }
    // (an explicit LDS pointer: volatile accesses through a generic pointer are issued as flat_ instructions)
    typedef __attribute__((address_space(3))) volatile uint32_t lds_counter;
    lds_counter *wh = (lds_counter *)wave_hist[wave];
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < PT_SORT_ITEMS; r++) {
        // out-of-range slots take digit 255 and, being last in input order, end
        // up behind every real key of the tile; they are never written out
        uint32_t d = (uint32_t)(key[r] >> shift) & 255u;
        uint64_t peers = ~0ull;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            bool bit = (d >> b) & 1u;
            uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        uint32_t lower = __popcll(peers & lt_mask);
        uint32_t prev = wh[d];
        rank[r] = prev + lower;
        __builtin_amdgcn_wave_barrier();
        if (lower == 0) wh[d] = prev + __popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {   // per digit: exclusive prefix over the four waves, then over the digits
        uint32_t c0 = wave_hist[0][t], c1 = wave_hist[1][t], c2 = wave_hist[2][t], c3 = wave_hist[3][t];
        uint32_t tot = c0 + c1 + c2 + c3, dummy;
        // (the padding slots of a ragged last tile count as digit 255 here: no tile follows it)
        if (CHAINED) chain_store(status + (size_t)tile * 256 + t, (tile == 0 ? CH_PREFIX : CH_AGG) | tot);
        wave_hist[0][t] = 0; wave_hist[1][t] = c0; wave_hist[2][t] = c0 + c1; wave_hist[3][t] = c0 + c1 + c2;
        digit_base[t] = block_excl_scan_256(tot, sh, &dummy);
        digit_gbase[t] = block_excl_scan_256(digit_total[t], sh, &dummy);
        if (CHAINED) {
            const uint32_t before = chain_lookback(status, tile, t, chain_error);
            if (tile > 0) chain_store(status + (size_t)tile * 256 + t, CH_PREFIX | (before + tot));
            tile_off[t] = before;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PT_SORT_ITEMS; r++) {
        uint32_t d = (uint32_t)(key[r] >> shift) & 255u;
        uint32_t p = digit_base[d] + wave_hist[wave][d] + rank[r];
        rank[r] = p;                                    // the slot inside the tile's sorted order
        skeys[p] = key[r];
        if (!TWO_ROUNDS) svals[p] = val[r];
    }
    __syncthreads();
    if (!TWO_ROUNDS) {
        for (int i = t; i < tile_cnt; i += PT_SORT_THREADS) {
            K k = skeys[i];
            uint32_t d = (uint32_t)(k >> shift) & 255u;
            uint32_t dst = digit_gbase[d] + (CHAINED ? tile_off[d] : block_off[(size_t)d * nblocks + blockIdx.x]) + ((uint32_t)i - digit_base[d]);
            if (KEYS_OUT) keys_out[dst] = k;
            vals_out[dst] = svals[i];
        }
    } else {
        uint32_t dst[PT_SORT_ITEMS];
#pragma unroll
        for (int r = 0; r < PT_SORT_ITEMS; r++) {
            const int i = t + r * PT_SORT_THREADS;
            if (i < tile_cnt) {
                K k = skeys[i];
                uint32_t d = (uint32_t)(k >> shift) & 255u;
                dst[r] = digit_gbase[d] + (CHAINED ? tile_off[d] : block_off[(size_t)d * nblocks + blockIdx.x]) + ((uint32_t)i - digit_base[d]);
                if (KEYS_OUT) keys_out[dst[r]] = k;
            }
        }
        __syncthreads();                                // every key has been read: the buffer takes the values
#pragma unroll
        for (int r = 0; r < PT_SORT_ITEMS; r++) svals[rank[r]] = val[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < PT_SORT_ITEMS; r++) {
            const int i = t + r * PT_SORT_THREADS;
            if (i < tile_cnt) vals_out[dst[r]] = svals[i];
        }
    }
}

// ---------------------------------------------------------------------------
// stage 4: binary radix tree links                    (radix_build_cs.hlsl)
// Keyed on the leaf index over the range [0, n]: delta(i, j) = clz(i ^ j), or -1
// outside the range.  One lane per internal node; writes left/right of the
// node and parent of both children.  Node 2n lies outside the reference's
// buffer: writes to it are dropped.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int lcp(int i, int j, int len)
{
    return (0 <= j && j < len) ? __clz(i ^ j) : -1;
}

__global__ __launch_bounds__(256) void k_radix_tree(int n, pt_TreeNode *__restrict__ tree)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int len = n + 1;
    const int d = (lcp(i, i + 1, len) - lcp(i, i - 1, len) > 0) ? 1 : -1;
    const int sig_min = lcp(i, i - d, len);
    int lmax = 2;
    while (lcp(i, i + lmax * d, len) > sig_min) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (lcp(i, i + (l + t) * d, len) > sig_min) l += t;
    const int j = i + l * d;
    const int sig_node = lcp(i, j, len);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (lcp(i, i + (s + t) * d, len) > sig_node) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int left = (min(i, j) == gamma) ? gamma + n : gamma;
    const int right = (max(i, j) == gamma + 1) ? n + gamma + 1 : gamma + 1;
    tree[i].left = left;
    tree[i].right = right;
    if (i == 0) tree[0].parent = 0;
    tree[left].parent = i;
    if (right < 2 * n) tree[right].parent = i;
}

// ---------------------------------------------------------------------------
// stage 5: leaf init + bottom-up boxes                 (bvh_build_cs.hlsl)
// The tree over keys 0..n is the binary trie of the key indices, so the node
// with a given range is known in closed form: at level k (split on bit k-1) the
// block b covers keys [b*2^k, min((b+1)*2^k - 1, n)] and is a node iff
// b*2^k + 2^(k-1) <= n; its id is the LAST key of the range when it is a left
// child (b even and (b+1)*2^k <= n), else the FIRST key.  Levels are processed
// bottom-up with no inter-workgroup hand-off: levels 1..10 of a 1024-key chunk
// inside one workgroup through LDS, higher levels one launch per level over the
// per-chunk boxes.  Key n is the out-of-range leaf: all-zero box, primitive 0.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int node_id(long long b, int k, int n)
{
    long long s = b << k;
    return ((b & 1) == 0 && ((b + 1) << k) <= (long long)n) ? (int)(((b + 1) << k) - 1) : (int)s;
}

__device__ __forceinline__ void write_wide(WideNode *__restrict__ w, const float *L, const float *R, int left, int right)
{
    float4 *dst = reinterpret_cast<float4 *>(w);
    // (the few nodes of the upper levels; the lower levels stage their records through LDS.)  Ordinary stores:
    // streaming stores of partial lines made the refit three times slower
    dst[0] = make_float4(L[0], L[1], L[2], L[3]);
    dst[1] = make_float4(L[4], L[5], R[0], R[1]);
    dst[2] = make_float4(R[2], R[3], R[4], R[5]);
    dst[3] = make_float4(__int_as_float(left), __int_as_float(right), 0.0f, 0.0f);
}

// parent of the node at (level k, block b): the nearest ancestor block that is a node.  The
// block one level up is a node when we are its right half (b odd) or when its right half is
// not empty; otherwise every ancestor that has us in its left half has an empty right half, and
// the parent is the first ancestor that has us in its right half.
__device__ __forceinline__ int parent_id(long long b, int k, int n, int top)
{
    if (k == top) return 0;                           // the root keeps parent 0
    if ((b & 1) || ((b + 1) << k) <= (long long)n) return node_id(b >> 1, k + 1, n);
    const int m = __ffsll((unsigned long long)b);     // b even, non-zero here
    return node_id(b >> m, k + m, n);
}

// one whole 48-byte record of the reference layout: left, right, parent, index, box, padding
__device__ __forceinline__ void write_tree_node(pt_TreeNode *__restrict__ node, int left, int right, int parent,
                                                int index, const float *U)
{
    float4 *dst = reinterpret_cast<float4 *>(node);
    dst[0] = make_float4(__int_as_float(left), __int_as_float(right), __int_as_float(parent), __int_as_float(index));
    dst[1] = make_float4(U[0], U[1], U[2], 0.0f);
    dst[2] = make_float4(U[3], U[4], U[5], 0.0f);
}

// PACK: the lane fetches its primitive itself (the 80-byte record, by the permutation) instead of the
// 32-byte box, evaluates the box with the expressions of k_bounds, and leaves the two records the trace
// kernels read per leaf, in leaf order: {a, e1 = b - a, e2 = c - a, primitive id} and {normals, material}.
// One random 80-byte fetch serves the refit and both records; a separate pass over the primitives would
// fetch them again.  Key n, the out-of-range leaf, gets the records of primitive 0 (ray_tracing.hlsl:219-226
// reads an all-zero node there: zero box, index 0).
template <bool PACK>
__global__ __launch_bounds__(PT_REFIT_CHUNK) void k_refit_low(const uint32_t *__restrict__ perm, const BoxRec *__restrict__ boxes,
                                                   const uint4 *__restrict__ prims16,
                                                   int n, WideNode *__restrict__ wide, ChunkBox *__restrict__ chunk,
                                                   TriIsect *__restrict__ tri, TriShade *__restrict__ shade)
{
    // 48 KB that stage what the chunk writes, so that store instructions write whole lines: with PACK the per-leaf
    // records first, then (both variants) the boxes of the chunk's node records, 48 bytes per node id of the chunk
    __shared__ float4 lds[3 * PT_REFIT_CHUNK];
    __shared__ float sb[6][PT_REFIT_CHUNK];          // 24 KB: min.xyz, max.xyz per key slot
    __shared__ unsigned short meta[PT_REFIT_CHUNK];  // per node id of the chunk: level << 10 | first key slot; 0 = not written here
    const int t = threadIdx.x;
    const long long base = (long long)blockIdx.x * PT_REFIT_CHUNK;
    {   // one key per lane: every gather of the chunk is in flight at once
        const int j = t;
        long long key = base + j;
        float b[6] = {0, 0, 0, 0, 0, 0};
        if (PACK) {
            float4 r[6] = {};
            if (key <= n) {
                const int p = (key < n) ? (int)perm[key] : 0;
                const float4 *src = reinterpret_cast<const float4 *>(prims16 + (size_t)p * (PRIM_DW / 4));
                const float4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3], v4 = src[4];
                // xyzn x 3: positions at floats 0..2, 6..8, 12..14, normals at 3..5, 9..11, 15..17, material at 18
                const float A[3] = {v0.x, v0.y, v0.z}, B[3] = {v1.z, v1.w, v2.x}, C[3] = {v3.x, v3.y, v3.z};
                if (key < n) {
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        // math3d.h:13-23, as in k_bounds
                        const float x = A[k], y = B[k], z = C[k];
                        const float yz_max = (y > z) ? y : z, yz_min = (y < z) ? y : z;
                        b[3 + k] = (x > yz_max) ? x : yz_max;
                        b[k] = (x < yz_min) ? x : yz_min;
                    }
                }
                r[0] = make_float4(A[0], A[1], A[2], __int_as_float(p));
                r[1] = make_float4(B[0] - A[0], B[1] - A[1], B[2] - A[2], 0.0f);
                r[2] = make_float4(C[0] - A[0], C[1] - A[1], C[2] - A[2], 0.0f);
                r[3] = make_float4(v0.w, v1.x, v1.y, v4.z);           // normal 0, material id (bits)
                r[4] = make_float4(v2.y, v2.z, v2.w, 0.0f);
                r[5] = make_float4(v3.w, v4.x, v4.y, 0.0f);
            }
            // The records of a chunk are 48 KB of consecutive bytes per array, 48 bytes per lane: written from the lanes a
            // store instruction would touch 24 lines with 16 bytes each; through LDS every instruction writes 1 KiB in a row.
            const long long left_keys = (long long)n + 1 - base;
            const int quads = 3 * (int)(left_keys < PT_REFIT_CHUNK ? left_keys : PT_REFIT_CHUNK);
#pragma unroll
            for (int a = 0; a < 2; a++) {
                lds[3 * t] = r[3 * a]; lds[3 * t + 1] = r[3 * a + 1]; lds[3 * t + 2] = r[3 * a + 2];
                __syncthreads();
                float4 *dst = a == 0 ? reinterpret_cast<float4 *>(tri + base) : reinterpret_cast<float4 *>(shade + base);
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const int q = t + i * PT_REFIT_CHUNK;
                    if (q < quads) dst[q] = lds[q];
                }
                __syncthreads();
            }
        } else if (key < n) {
            const int p = (int)perm[key];
            const float4 *src = reinterpret_cast<const float4 *>(&boxes[p]);
            float4 lo = src[0], hi = src[1];
            b[0] = lo.x; b[1] = lo.y; b[2] = lo.z; b[3] = hi.x; b[4] = hi.y; b[5] = hi.z;
        }
#pragma unroll
        for (int c = 0; c < 6; c++) sb[c][j] = b[c];
        meta[j] = 0;
    }
    __syncthreads();
    for (int k = 1; k <= PT_REFIT_CHUNK_LOG2; k++) {
        const int nb = PT_REFIT_CHUNK >> k, half = 1 << (k - 1);
        for (int bl = t; bl < nb; bl += PT_REFIT_CHUNK) {
            const int ls = bl << k;
            const long long s = base + ls;
            if (s + half <= n) {
                const int rs = ls + half;
                float L[6], R[6], U[6];
#pragma unroll
                for (int c = 0; c < 6; c++) { L[c] = sb[c][ls]; R[c] = sb[c][rs]; }
#pragma unroll
                for (int c = 0; c < 3; c++) { U[c] = fminf(L[c], R[c]); U[3 + c] = fmaxf(L[3 + c], R[3 + c]); }
                // the node's id lies in the chunk's key range (node_id: the first or the last key of the node's range)
                const int slot = (int)(node_id(s >> k, k, n) - base);
                lds[3 * slot] = make_float4(L[0], L[1], L[2], L[3]);
                lds[3 * slot + 1] = make_float4(L[4], L[5], R[0], R[1]);
                lds[3 * slot + 2] = make_float4(R[2], R[3], R[4], R[5]);
                meta[slot] = (unsigned short)((k << 10) | ls);
#pragma unroll
                for (int c = 0; c < 6; c++) sb[c][ls] = U[c];
            }
        }
        __syncthreads();
    }
    if (t < 6) (&chunk[blockIdx.x].mn[0])[t] = sb[t][0];
    // The chunk's node records are 64 KB of consecutive bytes (one id of the range belongs to a node of a higher level and
    // is skipped): written record by record as the levels finish, a store instruction touched 64 lines with 16 bytes each.
    float4 *dst = reinterpret_cast<float4 *>(wide + base);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int q = t + i * PT_REFIT_CHUNK, slot = q >> 2, part = q & 3;
        const unsigned m = meta[slot];
        const int k = (int)(m >> 10);
        if (k != 0) {
            const long long s = base + (long long)(m & 1023u);
            const long long gamma = s + (1ll << (k - 1)) - 1;
            const long long e = min(s + (1ll << k) - 1, (long long)n);
            // a leaf child is named ~key (its position in the sorted order)
            const int left = (k == 1) ? ~(int)s : (int)gamma;
            const int right = (gamma + 1 == e) ? ~(int)(gamma + 1) : (int)(gamma + 1);
            const float4 box = lds[3 * slot + (part < 3 ? part : 0)];
            dst[q] = part < 3 ? box : make_float4(__int_as_float(left), __int_as_float(right), 0.0f, 0.0f);
        }
    }
}

// Levels above a unit (a unit = the box of 2^base consecutive keys: a 1024-key chunk, or a
// group of units of the previous launch): a workgroup reduces up to 11 levels over 2048
// consecutive units in LDS and leaves the box of its whole range in `out`.
constexpr int PT_REFIT_GROUP_LOG2 = 11;
constexpr int PT_REFIT_GROUP = 1 << PT_REFIT_GROUP_LOG2;

__global__ __launch_bounds__(1024) void k_refit_upper(int base, int top, int n, const ChunkBox *__restrict__ in,
                                                      long long n_units, WideNode *__restrict__ wide,
                                                      ChunkBox *__restrict__ out)
{
    __shared__ float sb[6][PT_REFIT_GROUP];          // 48 KB
    const int t = threadIdx.x;
    const long long unit0 = (long long)blockIdx.x * PT_REFIT_GROUP;
    for (int j = t; j < PT_REFIT_GROUP; j += 1024) {
        const long long u = unit0 + j;
        float b[6] = {0, 0, 0, 0, 0, 0};
        if (u < n_units) {
#pragma unroll
            for (int c = 0; c < 3; c++) { b[c] = in[u].mn[c]; b[3 + c] = in[u].mx[c]; }
        }
#pragma unroll
        for (int c = 0; c < 6; c++) sb[c][j] = b[c];
    }
    __syncthreads();
    const int last = min(base + PT_REFIT_GROUP_LOG2, top);
    for (int k = base + 1; k <= last; k++) {
        const int sh = k - base;                         // units per block of level k: 2^sh
        const int nb = PT_REFIT_GROUP >> sh;
        for (int bl = t; bl < nb; bl += 1024) {
            const int ls = bl << sh;
            const long long s = (unit0 + ls) << base, half = 1ll << (k - 1);
            if (s + half <= n) {
                const int rs = ls + (1 << (sh - 1));
                float L[6], R[6], U[6];
#pragma unroll
                for (int c = 0; c < 6; c++) { L[c] = sb[c][ls]; R[c] = sb[c][rs]; }
#pragma unroll
                for (int c = 0; c < 3; c++) { U[c] = fminf(L[c], R[c]); U[3 + c] = fmaxf(L[3 + c], R[3 + c]); }
                const long long gamma = s + half - 1;
                const long long e = min(s + (1ll << k) - 1, (long long)n);
                // above chunk level a single-key right range can only be key n itself
                const int right = (gamma + 1 == e) ? ~(int)(gamma + 1) : (int)(gamma + 1);
                write_wide(&wide[node_id(s >> k, k, n)], L, R, (int)gamma, right);
#pragma unroll
                for (int c = 0; c < 6; c++) sb[c][ls] = U[c];
            }
        }
        __syncthreads();
    }
    if (out && t < 6) (&out[blockIdx.x].mn[0])[t] = sb[t][0];
}

// ---------------------------------------------------------------------------
// The tree in the reference's layout (TreeNode[2n]: tree_uav of ray_tracer.cpp:263-279), made
// on request from the traversal records: leaves n..2n-1 hold the sorted primitives' boxes and
// indices, internal node boxes are the unions of their children's boxes (the same min / max
// the refit evaluates), links and parents come from the closed form.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_export_leaves(const uint32_t *__restrict__ perm, const BoxRec *__restrict__ boxes,
                                                       int n, pt_TreeNode *__restrict__ tree)
{
    const int key = blockIdx.x * 256 + threadIdx.x;
    if (key >= n) return;
    const int p = (int)perm[key];
    const float4 *src = reinterpret_cast<const float4 *>(&boxes[p]);
    const float4 lo = src[0], hi = src[1];
    const float b[6] = {lo.x, lo.y, lo.z, hi.x, hi.y, hi.z};
    write_tree_node(&tree[(long long)n + key], 0, 0, node_id(key >> 1, 1, n), p, b);
}

// blockIdx.y + 1 = level k; one thread per block b of that level
__global__ __launch_bounds__(256) void k_export_nodes(int n, int top, const WideNode *__restrict__ wide,
                                                      pt_TreeNode *__restrict__ tree)
{
    const int k = (int)blockIdx.y + 1;
    const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long s = b << k, half = 1ll << (k - 1);
    if (s + half > n) return;
    const long long gamma = s + half - 1;
    const long long e = min(s + (1ll << k) - 1, (long long)n);
    const int id = node_id(b, k, n);
    const float4 *w = reinterpret_cast<const float4 *>(&wide[id]);
    const float4 w0 = w[0], w1 = w[1], w2 = w[2];
    const float L[6] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y}, R[6] = {w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
    float U[6];
#pragma unroll
    for (int c = 0; c < 3; c++) { U[c] = fminf(L[c], R[c]); U[3 + c] = fmaxf(L[3 + c], R[3 + c]); }
    write_tree_node(&tree[id], (k == 1) ? n + (int)s : (int)gamma,
                    (gamma + 1 == e) ? n + (int)gamma + 1 : (int)(gamma + 1), parent_id(b, k, n, top), 0, U);
}

// sorted {box, index} records in the reference's ray_cs_primitives layout
__global__ __launch_bounds__(256) void k_export_sorted(const uint32_t *__restrict__ perm, const BoxRec *__restrict__ boxes,
                                                       int n, pt_PrimitiveRayCS *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 *src = reinterpret_cast<const float4 *>(&boxes[perm[i]]);
    float4 *dst = reinterpret_cast<float4 *>(&out[i]);
    dst[0] = src[0];
    dst[1] = src[1];
}

// Folds the launch's per-pixel iteration counts into the decaying cost (window of about eight
// launches), clears them, and makes the pixel's sort key for the trace kernel's work queue: most
// expensive first.
__global__ __launch_bounds__(256) void k_cost_keys(uint32_t *__restrict__ cost, uint32_t *__restrict__ pixel_work,
                                                   int npix, int shift, uint32_t *__restrict__ keys)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const uint32_t old = cost[i];
    const uint32_t now = old - (old >> 3) + pixel_work[i];
    cost[i] = now;
    pixel_work[i] = 0;
    const uint32_t c = now >> shift;
    keys[i] = 65535u - (c > 65535u ? 65535u : c);
}

// ---------------------------------------------------------------------------
// NON-PARITY option PT_BUILD_TOPOLOGY_MORTON (no counterpart in the reference, whose tree is keyed
// on the leaf index): the binary radix tree over the SORTED CODES, equal codes told apart by their
// position (Karras 2012, section 4).  n keys give n - 1 internal nodes, root 0; children are named
// by the split position as everywhere else (left = gamma, right = gamma + 1, leaves as ~key).
//
// Boxes bottom-up without any inter-workgroup hand-off.  A node's descendants have ids inside its key
// range, so every node whose range lies inside one 1024-key chunk is finished by the workgroup of that
// chunk through LDS (k_refit_chunk: rounds of "both children done -> my turn", one round per level of
// the tallest subtree of the chunk), with coalesced reads and writes.  The nodes whose range crosses a
// chunk border (about one per border and level) are listed by the search kernel with their prefix
// length; a child's common prefix is strictly longer than its parent's, so they are finished longest
// prefix first, one small launch per length.
// ---------------------------------------------------------------------------
template <typename K>
__device__ __forceinline__ int lcp_codes(const K *__restrict__ codes, K ci, int i, int j, int n)
{
    if (j < 0 || j >= n) return -1;
    const K x = ci ^ codes[j];
    if (x != 0) return sizeof(K) == 8 ? __clzll((unsigned long long)x) : __clz((unsigned)x);
    return 8 * (int)sizeof(K) + __clz(i ^ j);          // i != j here
}

constexpr int LINK_RIGHT_LEAF = 1, LINK_IN_CHUNK = 2;
// A node that crosses a border has one end of its range (its own id) in its chunk and crosses the border on the other
// side; the nodes crossing one border are nested, so a chunk holds at most 2 x height of them: 128 slots per chunk
// cover every tree the build accepts (PT_MAX_TREE_HEIGHT), and a chunk with more sets `overflow` (the build fails).
constexpr int CROSS_SLOTS = 2 * PT_MAX_TREE_HEIGHT;

// One workgroup per 1024-key chunk.  links[i] = {left child (a leaf as ~key), LINK_* bits}: the right child is the
// left one's key + 1.  cross[chunk * CROSS_SLOTS + k] = {node, prefix length} for the chunk's border-crossing nodes.
template <typename K>
__global__ __launch_bounds__(PT_REFIT_CHUNK) void k_karras_codes(const K *__restrict__ codes, int n, int2 *__restrict__ links,
                                                                 uint2 *__restrict__ cross, uint32_t *__restrict__ cross_count,
                                                                 unsigned *__restrict__ overflow)
{
    __shared__ unsigned cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const int i = blockIdx.x * PT_REFIT_CHUNK + threadIdx.x;
    if (i < n - 1) {
        const K ci = codes[i];
        const int d = (lcp_codes(codes, ci, i, i + 1, n) - lcp_codes(codes, ci, i, i - 1, n) > 0) ? 1 : -1;
        const int sig_min = lcp_codes(codes, ci, i, i - d, n);
        int lmax = 2;
        while (lcp_codes(codes, ci, i, i + lmax * d, n) > sig_min) lmax *= 2;
        int l = 0;
        for (int t = lmax / 2; t >= 1; t /= 2)
            if (lcp_codes(codes, ci, i, i + (l + t) * d, n) > sig_min) l += t;
        const int j = i + l * d;
        const int sig_node = lcp_codes(codes, ci, i, j, n);
        int s = 0, t = l;
        do {
            t = (t + 1) >> 1;
            if (lcp_codes(codes, ci, i, i + (s + t) * d, n) > sig_node) s += t;
        } while (t > 1);
        const int gamma = i + s * d + min(d, 0);
        const int lo = min(i, j), hi = max(i, j);
        const bool crossing = (lo >> PT_REFIT_CHUNK_LOG2) != (hi >> PT_REFIT_CHUNK_LOG2);
        links[i] = make_int2(lo == gamma ? ~gamma : gamma, (hi == gamma + 1 ? LINK_RIGHT_LEAF : 0) | (crossing ? 0 : LINK_IN_CHUNK));
        if (crossing) {
            const unsigned slot = atomicAdd(&cnt, 1u);
            if (slot < (unsigned)CROSS_SLOTS) cross[(size_t)blockIdx.x * CROSS_SLOTS + slot] = make_uint2((unsigned)i, (unsigned)sig_node);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        cross_count[blockIdx.x] = min(cnt, (unsigned)CROSS_SLOTS);
        if (cnt > (unsigned)CROSS_SLOTS) atomicOr(overflow, 1u);
    }
}

__device__ __forceinline__ void write_wide_h(WideNode *__restrict__ w, const float *L, const float *R, int left, int right, int height)
{
    float4 *dst = reinterpret_cast<float4 *>(w);
    dst[0] = make_float4(L[0], L[1], L[2], L[3]);
    dst[1] = make_float4(L[4], L[5], R[0], R[1]);
    dst[2] = make_float4(R[2], R[3], R[4], R[5]);
    dst[3] = make_float4(__int_as_float(left), __int_as_float(right), __int_as_float(height), 0.0f);   // height: levels below the node
}

// A node the rounds below cannot finish (a chunk subtree higher than the supported tree height) gets this height,
// which every ancestor inherits: the build then fails on the root's height.
constexpr int HEIGHT_UNFINISHED = 1 << 20;

// One workgroup per 1024 keys: the leaves' boxes (and, with PACK, the two per-leaf records of the trace
// kernels, as k_refit_low), then every node whose key range lies inside the chunk.
template <bool PACK>
__global__ __launch_bounds__(PT_REFIT_CHUNK) void k_refit_chunk(const uint32_t *__restrict__ perm, const BoxRec *__restrict__ boxes,
                                                     const uint4 *__restrict__ prims16, int n, const int2 *__restrict__ links,
                                                     WideNode *__restrict__ wide, TriIsect *__restrict__ tri, TriShade *__restrict__ shade)
{
    __shared__ float lb[6][PT_REFIT_CHUNK];          // leaf boxes
    // 48 KB used three times: staging of the per-leaf records (PACK), the node boxes and heights of the rounds, staging
    // of the node records -- so that store instructions write whole lines (see k_refit_low)
    __shared__ float4 lds[3 * PT_REFIT_CHUNK];
    float (*nb)[PT_REFIT_CHUNK] = reinterpret_cast<float (*)[PT_REFIT_CHUNK]>(lds);     // node boxes
    int *hh = reinterpret_cast<int *>(lds) + 6 * PT_REFIT_CHUNK;                         // node heights; 0 = not finished yet
    const int t = threadIdx.x;
    const int base = blockIdx.x * PT_REFIT_CHUNK;
    const int key = base + t;
    float4 r[6] = {};
    if (key < n) {
        const int p = (int)perm[key];
        float b[6];
        if (PACK) {
            const float4 *src = reinterpret_cast<const float4 *>(prims16 + (size_t)p * (PRIM_DW / 4));
            const float4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3], v4 = src[4];
            const float A[3] = {v0.x, v0.y, v0.z}, B[3] = {v1.z, v1.w, v2.x}, C[3] = {v3.x, v3.y, v3.z};
#pragma unroll
            for (int k = 0; k < 3; k++) {
                // math3d.h:13-23, as in k_bounds
                const float x = A[k], y = B[k], z = C[k];
                const float yz_max = (y > z) ? y : z, yz_min = (y < z) ? y : z;
                b[3 + k] = (x > yz_max) ? x : yz_max;
                b[k] = (x < yz_min) ? x : yz_min;
            }
            r[0] = make_float4(A[0], A[1], A[2], __int_as_float(p));
            r[1] = make_float4(B[0] - A[0], B[1] - A[1], B[2] - A[2], 0.0f);
            r[2] = make_float4(C[0] - A[0], C[1] - A[1], C[2] - A[2], 0.0f);
            r[3] = make_float4(v0.w, v1.x, v1.y, v4.z);
            r[4] = make_float4(v2.y, v2.z, v2.w, 0.0f);
            r[5] = make_float4(v3.w, v4.x, v4.y, 0.0f);
        } else {
            const float4 *src = reinterpret_cast<const float4 *>(&boxes[p]);
            const float4 lo = src[0], hi = src[1];
            b[0] = lo.x; b[1] = lo.y; b[2] = lo.z; b[3] = hi.x; b[4] = hi.y; b[5] = hi.z;
        }
#pragma unroll
        for (int c = 0; c < 6; c++) lb[c][t] = b[c];
    }
    if (PACK) {
        const int left_keys = n - base;
        const int quads = 3 * (left_keys < PT_REFIT_CHUNK ? left_keys : PT_REFIT_CHUNK);
#pragma unroll
        for (int a = 0; a < 2; a++) {
            lds[3 * t] = r[3 * a]; lds[3 * t + 1] = r[3 * a + 1]; lds[3 * t + 2] = r[3 * a + 2];
            __syncthreads();
            float4 *dst = a == 0 ? reinterpret_cast<float4 *>(tri + base) : reinterpret_cast<float4 *>(shade + base);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const int q = t + i * PT_REFIT_CHUNK;
                if (q < quads) dst[q] = lds[q];
            }
            __syncthreads();
        }
    }
    int left = 0, right = 0;
    bool todo = false;
    if (key < n - 1) {
        const int2 ln = links[key];
        left = ln.x;
        const int gamma = left < 0 ? ~left : left;
        right = (ln.y & LINK_RIGHT_LEAF) ? ~(gamma + 1) : gamma + 1;
        todo = (ln.y & LINK_IN_CHUNK) != 0;
    }
    const bool mine = todo;                          // this workgroup writes the node's record
    // A node the rounds cannot finish (a chunk subtree higher than the supported tree height) keeps HEIGHT_UNFINISHED
    float KL[6] = {0, 0, 0, 0, 0, 0}, KR[6] = {0, 0, 0, 0, 0, 0};
    int kh = HEIGHT_UNFINISHED;
    hh[t] = 0;
    __syncthreads();
    // every descendant of a node of this chunk is a leaf or a node of this chunk: slots left - base, ~left - base
    for (int round = 0; round <= PT_MAX_TREE_HEIGHT; round++) {
        bool ready = false;
        float L[6], R[6];
        int h = 0;
        if (todo) {
            const int hl = left < 0 ? 0 : hh[left - base], hr = right < 0 ? 0 : hh[right - base];
            ready = (left < 0 || hl > 0) && (right < 0 || hr > 0);
            if (ready) {
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    L[c] = left < 0 ? lb[c][~left - base] : nb[c][left - base];
                    R[c] = right < 0 ? lb[c][~right - base] : nb[c][right - base];
                }
                h = 1 + max(hl, hr);
            }
        }
        if (!__syncthreads_or(todo)) break;          // also: every read of this round precedes its writes
        if (ready) {
#pragma unroll
            for (int c = 0; c < 3; c++) { nb[c][t] = fminf(L[c], R[c]); nb[3 + c][t] = fmaxf(L[3 + c], R[3 + c]); }
            hh[t] = h;
#pragma unroll
            for (int c = 0; c < 6; c++) { KL[c] = L[c]; KR[c] = R[c]; }
            kh = h;
            todo = false;
        }
        __syncthreads();
    }
    // The records of the chunk's nodes, 64 bytes each at consecutive ids: 512 at a time through LDS (80 bytes apart there:
    // no bank conflicts), skipping the ids of nodes that cross a chunk border (k_refit_cross writes those)
    float4 *dst = reinterpret_cast<float4 *>(wide + base);
    for (int half = 0; half < 2; half++) {
        if ((t >> 9) == half) {
            const int s = 5 * (t & 511);
            lds[s] = make_float4(KL[0], KL[1], KL[2], KL[3]);
            lds[s + 1] = make_float4(KL[4], KL[5], KR[0], KR[1]);
            lds[s + 2] = make_float4(KR[2], KR[3], KR[4], KR[5]);
            lds[s + 3] = make_float4(__int_as_float(left), __int_as_float(right), __int_as_float(kh), mine ? 1.0f : 0.0f);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int q = t + i * PT_REFIT_CHUNK, rec = q >> 2, part = q & 3;
            if (lds[5 * rec + 3].w != 0.0f) {
                float4 v = lds[5 * rec + part];
                if (part == 3) v.w = 0.0f;
                dst[half * 2 * PT_REFIT_CHUNK + q] = v;
            }
        }
        __syncthreads();
    }
}

// box and height of child `c` of a node whose range crosses a chunk border
__device__ __forceinline__ void cross_child_box(int c, const WideNode *__restrict__ wide, const uint32_t *__restrict__ perm,
                                                const BoxRec *__restrict__ boxes, float *B, int &height)
{
    if (c >= 0) {
        const float4 *w = reinterpret_cast<const float4 *>(&wide[c]);
        const float4 w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
        B[0] = fminf(w0.x, w1.z); B[1] = fminf(w0.y, w1.w); B[2] = fminf(w0.z, w2.x);
        B[3] = fmaxf(w0.w, w2.y); B[4] = fmaxf(w1.x, w2.z); B[5] = fmaxf(w1.y, w2.w);
        height = __float_as_int(w3.z);
    } else {
        const float4 *src = reinterpret_cast<const float4 *>(&boxes[perm[~c]]);
        const float4 lo = src[0], hi = src[1];
        B[0] = lo.x; B[1] = lo.y; B[2] = lo.z; B[3] = hi.x; B[4] = hi.y; B[5] = hi.z;
        height = 0;
    }
}

// the listed nodes whose prefix length is `level`: one lane per chunk walks the chunk's few entries
__global__ __launch_bounds__(256) void k_refit_cross(int level, int nchunks, const uint2 *__restrict__ cross,
                                                     const uint32_t *__restrict__ cross_count,
                                                     const int2 *__restrict__ links, WideNode *__restrict__ wide,
                                                     const uint32_t *__restrict__ perm, const BoxRec *__restrict__ boxes)
{
    for (int chunk = blockIdx.x * 256 + threadIdx.x; chunk < nchunks; chunk += gridDim.x * 256) {
        const unsigned count = cross_count[chunk];
        for (unsigned k = 0; k < count; k++) {
            const uint2 e = cross[(size_t)chunk * CROSS_SLOTS + k];
            if ((int)e.y != level) continue;
            const int node = (int)e.x;
            const int2 ln = links[node];
            const int left = ln.x, gamma = left < 0 ? ~left : left;
            const int right = (ln.y & LINK_RIGHT_LEAF) ? ~(gamma + 1) : gamma + 1;
            float L[6], R[6];
            int hl, hr;
            cross_child_box(left, wide, perm, boxes, L, hl);
            cross_child_box(right, wide, perm, boxes, R, hr);
            const int h = 1 + max(hl, hr);
            write_wide_h(&wide[node], L, R, left, right, h > HEIGHT_UNFINISHED ? HEIGHT_UNFINISHED : h);
        }
    }
}

// reference-layout export of the Morton-keyed tree: records first (parent 0), then the parents
__global__ __launch_bounds__(256) void k_export_leaves_codes(const uint32_t *__restrict__ perm, const BoxRec *__restrict__ boxes,
                                                             int n, pt_TreeNode *__restrict__ tree)
{
    const int key = blockIdx.x * 256 + threadIdx.x;
    if (key >= n) return;
    const int p = (int)perm[key];
    const float4 *src = reinterpret_cast<const float4 *>(&boxes[p]);
    const float4 lo = src[0], hi = src[1];
    const float b[6] = {lo.x, lo.y, lo.z, hi.x, hi.y, hi.z};
    write_tree_node(&tree[(long long)n + key], 0, 0, 0, p, b);
}
__global__ __launch_bounds__(256) void k_export_nodes_codes(int n, const WideNode *__restrict__ wide, pt_TreeNode *__restrict__ tree)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n - 1) return;
    const float4 *w = reinterpret_cast<const float4 *>(&wide[i]);
    const float4 w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
    const float L[6] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y}, R[6] = {w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
    float U[6];
#pragma unroll
    for (int c = 0; c < 3; c++) { U[c] = fminf(L[c], R[c]); U[3 + c] = fmaxf(L[3 + c], R[3 + c]); }
    const int left = __float_as_int(w3.x), right = __float_as_int(w3.y);
    write_tree_node(&tree[i], left < 0 ? n + ~left : left, right < 0 ? n + ~right : right, 0, 0, U);
}
__global__ __launch_bounds__(256) void k_export_parents_codes(int n, const WideNode *__restrict__ wide, pt_TreeNode *__restrict__ tree)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n - 1) return;
    const float4 w3 = reinterpret_cast<const float4 *>(&wide[i])[3];
    const int left = __float_as_int(w3.x), right = __float_as_int(w3.y);
    tree[left < 0 ? n + ~left : left].parent = i;
    tree[right < 0 ? n + ~right : right].parent = i;
}

inline int bit_length(unsigned v) { return v ? 32 - __builtin_clz(v) : 0; }

}  // namespace

// `passes` stable 8-bit passes over `keys` (n keys; values start as 0..n-1).  The passes
// ping-pong between kbuf/vbuf [0] and [1]; returns the index of the vbuf that holds the
// permutation.  The sorted keys themselves are produced by the last pass only with `keys_last`
// (into the kbuf of the same index).
template <typename K>
static int radix_sort_passes(hipStream_t st, const K *keys, int n, int passes, K *const kbuf[2], uint32_t *const vbuf[2],
                             uint32_t *block_hist, uint32_t *digit_total, bool keys_last = false, bool first_counted = false)
{
    const int nblocks = (n + PT_SORT_TILE - 1) / PT_SORT_TILE;
    const K *kin = keys;
    const uint32_t *vin = nullptr;
    int out = 0;
    for (int p = 0; p < passes; p++) {
        const int shift = 8 * p;
        uint32_t *dtot = digit_total + 256 * p;
        const bool first = (p == 0), last = (p == passes - 1) && !keys_last;
        if (!(first && first_counted))      // (k_morton_hist has counted the first pass's digits already)
            hipLaunchKernelGGL(k_sort_hist<K>, dim3(nblocks), dim3(PT_SORT_THREADS), 0, st, kin, n, shift, block_hist, nblocks);
        hipLaunchKernelGGL(k_sort_scan, dim3(256), dim3(256), 0, st, block_hist, dtot, nblocks);
#define PT_SCATTER(I, KO) hipLaunchKernelGGL((k_sort_scatter<K, I, KO, false>), dim3(nblocks), dim3(PT_SORT_THREADS), 0, st, kin, vin, \
                                             kbuf[out], vbuf[out], n, shift, block_hist, dtot, nblocks, nullptr, nullptr, nullptr)
        if (first) { if (last) PT_SCATTER(true, false); else PT_SCATTER(true, true); }
        else       { if (last) PT_SCATTER(false, false); else PT_SCATTER(false, true); }
#undef PT_SCATTER
        kin = kbuf[out];
        vin = vbuf[out];
        out ^= 1;
    }
    return out ^ 1;
}

// The same sort with ONE read of the keys per pass (k_sort_digit_counts, then k_sort_scatter<CHAINED>): `status` is the
// block_hist buffer (256 words per tile), digit_total the 8 * 256 + 512 words of the build (counts, tile counters, error
// word).  For n < 2^30.  Used for the eight passes over full-resolution codes (50 M keys: 3.4 ms against 4.6 ms); the two
// short passes over the reference's 15-bit codes stay with the counting pass: a pass of 50 us pays more for the start of
// the look-back chain than for reading its 4-byte keys twice (10 M keys: 0.21 ms chained against 0.15 ms).
static bool sort_chained_applies(const pt_context *ctx, int n, int passes)
{
    return ctx->knobs.sort_chained && passes > 2 && passes <= 8 && (unsigned)n < CH_AGG;
}

template <typename K, int PASSES>
static int radix_sort_chained(hipStream_t st, const K *keys, int n, K *const kbuf[2], uint32_t *const vbuf[2],
                              uint32_t *status, uint32_t *digit_total, int num_cus, bool keys_last)
{
    const int nblocks = (n + PT_SORT_TILE - 1) / PT_SORT_TILE;
    if (hipMemsetAsync(digit_total, 0, (8 * 256 + 512) * sizeof(uint32_t), st) != hipSuccess) return -1;
    const int count_grid = std::min((n + 1023) / 1024, num_cus * 8);
    hipLaunchKernelGGL((k_sort_digit_counts<K, PASSES>), dim3(count_grid), dim3(256), 0, st, keys, n, digit_total);
    const K *kin = keys;
    const uint32_t *vin = nullptr;
    int out = 0;
    for (int p = 0; p < PASSES; p++) {
        const bool first = (p == 0), last = (p == PASSES - 1) && !keys_last;
        if (hipMemsetAsync(status, 0, (size_t)nblocks * 256 * sizeof(uint32_t), st) != hipSuccess) return -1;
#define PT_SCATTER(I, KO) hipLaunchKernelGGL((k_sort_scatter<K, I, KO, true>), dim3(nblocks), dim3(PT_SORT_THREADS), 0, st, kin, vin, \
                                             kbuf[out], vbuf[out], n, 8 * p, nullptr, digit_total + 256 * p, nblocks, status, \
                                             digit_total + CH_TICKET_WORD + p, digit_total + CH_ERROR_WORD)
        if (first) { if (last) PT_SCATTER(true, false); else PT_SCATTER(true, true); }
        else       { if (last) PT_SCATTER(false, false); else PT_SCATTER(false, true); }
#undef PT_SCATTER
        kin = kbuf[out];
        vin = vbuf[out];
        out ^= 1;
    }
    return out ^ 1;
}

// Orders the pixels by the work they took in earlier launches (descending, 16-bit keys, two
// stable radix passes) so that the persistent trace kernel hands out the expensive pixels
// first and its last wave finishes on cheap ones.  Any order gives the same image.
int ptk_order_pixels(pt_context *ctx, int shift)
{
    const int npix = ctx->W * ctx->H;
    hipStream_t st = ctx->stream;
    // keys live in d_okeys[0]; pass 0 writes [1], pass 1 writes [0]: the order ends in d_ovals[0]
    hipLaunchKernelGGL(k_cost_keys, dim3((npix + 255) / 256), dim3(256), 0, st, ctx->d_cost, ctx->d_pixel_work, npix, shift, ctx->d_okeys[0]);
    uint32_t *const kb[2] = {ctx->d_okeys[1], ctx->d_okeys[0]};
    uint32_t *const vb[2] = {ctx->d_ovals[1], ctx->d_ovals[0]};
    (void)radix_sort_passes<uint32_t>(st, ctx->d_okeys[0], npix, 2, kb, vb, ctx->d_ohist, ctx->d_ototal);
    PT_HIP(ctx, hipGetLastError());
    ctx->order_valid = true;
    return PT_OK;
}

// ---------------------------------------------------------------------------
// host-side sequencing
// ---------------------------------------------------------------------------
// Stages bounds .. refit over the buffers of `b`.  `ev` (PT_STAGE_COUNT + 1 events) receives the
// stage boundaries up to the end of the refit, or is null.  Returns which of b.vals holds the
// permutation in *sorted_buf.
static bool build_keeps_centres_only(unsigned flags) { return (flags & (PT_BUILD_TREE_ONLY | PT_BUILD_TOPOLOGY_MORTON)) == 0; }

static int run_build(pt_context *ctx, const BuildBufs &b, const uint4 *prims16, int n, unsigned flags,
                     hipEvent_t *ev, int *sorted_buf, bool chained = false)
{
    hipStream_t st = ctx->stream;
    const int grid_n = (n + 255) / 256;

    if (ev) PT_HIP(ctx, hipEventRecord(ev[0], st));
    // only the Morton codes read the boxes in a default build of the index-keyed tree: centres in their place
    const bool mids = build_keeps_centres_only(flags);
    if (mids) hipLaunchKernelGGL(k_bounds<true>, dim3(grid_n), dim3(256), 0, st, prims16, n, b.boxes, b.bbox_partial);
    else      hipLaunchKernelGGL(k_bounds<false>, dim3(grid_n), dim3(256), 0, st, prims16, n, b.boxes, b.bbox_partial);
    hipLaunchKernelGGL(k_bbox_reduce, dim3(1), dim3(1024), 0, st, b.bbox_partial, grid_n, b.bbox_enc);
    if (ev) PT_HIP(ctx, hipEventRecord(ev[PT_STAGE_BOUNDS + 1], st));

    const bool full = (flags & PT_BUILD_MORTON_FULL) != 0;
    const float cells = full ? 2097151.0f : (float)PT_MORTON_CODE_LENGTH_EACH;
    // the reference's codes have 15 bits: 32-bit keys halve the sort's key traffic
#define PT_MORTON(K, M) hipLaunchKernelGGL((k_morton<K, M>), dim3(grid_n), dim3(256), 0, st, b.boxes, n, b.bbox_enc, cells, static_cast<K *>(b.morton))
    // the reference's codes: the kernel also counts the lowest digit per sorting tile, so the first radix pass starts at its scan
#define PT_MORTON_HIST(M) hipLaunchKernelGGL((k_morton_hist<uint32_t, M>), dim3((n + PT_SORT_TILE - 1) / PT_SORT_TILE), dim3(PT_SORT_THREADS), 0, st, \
                                             b.boxes, n, b.bbox_enc, cells, static_cast<uint32_t *>(b.morton), b.block_hist, (n + PT_SORT_TILE - 1) / PT_SORT_TILE)
    if (full) { if (mids) PT_MORTON(uint64_t, true); else PT_MORTON(uint64_t, false); }
    else      { if (mids) PT_MORTON_HIST(true); else PT_MORTON_HIST(false); }
#undef PT_MORTON
#undef PT_MORTON_HIST
    if (ev) PT_HIP(ctx, hipEventRecord(ev[PT_STAGE_MORTON + 1], st));

    // The reference quantises each axis to 0..21 (5 bits), so its codes occupy
    // 15 bits: two 8-bit passes sort them completely.  Full-resolution codes
    // have 63 bits: eight passes.
    const int passes = full ? 8 : 2;
    const bool by_codes = (flags & PT_BUILD_TOPOLOGY_MORTON) != 0;      // the tree search reads the sorted codes
    if (full) {
        uint64_t *const kb[2] = {static_cast<uint64_t *>(b.keys[0]), static_cast<uint64_t *>(b.keys[1])};
        if (chained) {
            *sorted_buf = radix_sort_chained<uint64_t, 8>(st, static_cast<const uint64_t *>(b.morton), n, kb, b.vals,
                                                          b.block_hist, b.digit_total, ctx->num_cus, by_codes);
            if (*sorted_buf < 0) PT_FAIL(ctx, PT_ERR_HIP, "pt_build: hipMemsetAsync of the sort's status words failed");
        } else {
            *sorted_buf = radix_sort_passes<uint64_t>(st, static_cast<const uint64_t *>(b.morton), n, passes, kb, b.vals,
                                                      b.block_hist, b.digit_total, by_codes);
        }
    } else {
        uint32_t *const kb[2] = {static_cast<uint32_t *>(b.keys[0]), static_cast<uint32_t *>(b.keys[1])};
        *sorted_buf = radix_sort_passes<uint32_t>(st, static_cast<const uint32_t *>(b.morton), n, passes, kb, b.vals,
                                                  b.block_hist, b.digit_total, by_codes, true);
    }
    if (ev) PT_HIP(ctx, hipEventRecord(ev[PT_STAGE_SORT + 1], st));

    if (by_codes) {
        // NON-PARITY topology: links by search over the sorted codes, nodes bucketed by prefix length (one
        // counting pass of the sort over the n - 1 lengths), boxes bucket by bucket from the longest prefix up
        const int sb = *sorted_buf, other = sb ^ 1;
        const uint32_t *perm = b.vals[sb];
        // the halves of the sort's ping-pong buffers that it left free: n x 8 bytes and n x 4 bytes
        uint2 *cross = static_cast<uint2 *>(b.keys[other]);              // CROSS_SLOTS x 8 bytes per 1024 keys
        uint32_t *cross_count = b.vals[other];
        unsigned *overflow = b.digit_total + 8 * 256;
        const int nchunks = (n + PT_REFIT_CHUNK - 1) / PT_REFIT_CHUNK;
        PT_HIP(ctx, hipMemsetAsync(overflow, 0, sizeof(unsigned), st));
        if (full) hipLaunchKernelGGL(k_karras_codes<uint64_t>, dim3(nchunks), dim3(PT_REFIT_CHUNK), 0, st, static_cast<const uint64_t *>(b.keys[sb]), n,
                                     b.links, cross, cross_count, overflow);
        else      hipLaunchKernelGGL(k_karras_codes<uint32_t>, dim3(nchunks), dim3(PT_REFIT_CHUNK), 0, st, static_cast<const uint32_t *>(b.keys[sb]), n,
                                     b.links, cross, cross_count, overflow);
        if (ev) PT_HIP(ctx, hipEventRecord(ev[PT_STAGE_TOPOLOGY + 1], st));
        if (flags & PT_BUILD_TREE_ONLY)
            hipLaunchKernelGGL(k_refit_chunk<false>, dim3(nchunks), dim3(PT_REFIT_CHUNK), 0, st, perm, b.boxes, prims16, n, b.links, b.wide, b.tri, b.shade);
        else
            hipLaunchKernelGGL(k_refit_chunk<true>, dim3(nchunks), dim3(PT_REFIT_CHUNK), 0, st, perm, b.boxes, prims16, n, b.links, b.wide, b.tri, b.shade);
        if (nchunks > 1) {
            // prefix lengths a border-crossing node can have: the codes' leading zero bits are common to all keys, and two
            // positions on either side of a border differ in bit 10 or above (equal codes: key bits + clz(i ^ j))
            const int longest = (full ? 64 : 32) + 31 - PT_REFIT_CHUNK_LOG2;
            const int shortest = full ? 1 : 17;
            const unsigned level_grid = (unsigned)std::min((nchunks + 255) / 256, 4 * ctx->num_cus);
            for (int level = longest; level >= shortest; level--)
                hipLaunchKernelGGL(k_refit_cross, dim3(level_grid), dim3(256), 0, st, level, nchunks, cross, cross_count, b.links, b.wide, perm, b.boxes);
        }
        if (ev) PT_HIP(ctx, hipEventRecord(ev[PT_STAGE_REFIT + 1], st));
        PT_HIP(ctx, hipGetLastError());
        return PT_OK;
    }

    // The topology of the index-keyed tree is a closed form of n (see the refit kernels): the
    // refit writes child ids from it, and nothing else has to run for this stage.
    if (ev) PT_HIP(ctx, hipEventRecord(ev[PT_STAGE_TOPOLOGY + 1], st));

    const uint32_t *perm = b.vals[*sorted_buf];
    const int nchunks = (int)(((long long)n + 1 + PT_REFIT_CHUNK - 1) / PT_REFIT_CHUNK);
    const int top = bit_length((unsigned)n);
    const bool tree_only = (flags & PT_BUILD_TREE_ONLY) != 0;
    if (tree_only)
        hipLaunchKernelGGL(k_refit_low<false>, dim3(nchunks), dim3(PT_REFIT_CHUNK), 0, st, perm, b.boxes, prims16, n, b.wide,
                           b.chunk, b.tri, b.shade);
    else
        hipLaunchKernelGGL(k_refit_low<true>, dim3(nchunks), dim3(PT_REFIT_CHUNK), 0, st, perm, b.boxes, prims16, n, b.wide,
                           b.chunk, b.tri, b.shade);
    // levels above the chunks: groups of 2048 units per launch (two launches reach 2^32 keys)
    {
        const ChunkBox *in = b.chunk;
        ChunkBox *out = b.chunk + nchunks;          // second half of the buffer, then back
        long long units = nchunks;
        for (int base = PT_REFIT_CHUNK_LOG2; base < top; base += PT_REFIT_GROUP_LOG2) {
            const long long groups = (units + PT_REFIT_GROUP - 1) / PT_REFIT_GROUP;
            hipLaunchKernelGGL(k_refit_upper, dim3((unsigned)groups), dim3(1024), 0, st, base, top, n, in, units, b.wide, out);
            in = out;
            out = b.chunk;
            units = groups;
        }
    }
    if (ev) PT_HIP(ctx, hipEventRecord(ev[PT_STAGE_REFIT + 1], st));
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

int ptk_build(pt_context *ctx, unsigned flags)
{
    BuildBufs b;
    b.boxes = ctx->d_boxes; b.bbox_partial = ctx->d_bbox_partial; b.bbox_enc = ctx->d_bbox_enc;
    b.keys[0] = ctx->d_keys[0]; b.keys[1] = ctx->d_keys[1];
    b.vals[0] = ctx->d_vals[0]; b.vals[1] = ctx->d_vals[1];
    b.morton = ctx->d_morton; b.block_hist = ctx->d_block_hist; b.digit_total = ctx->d_digit_total;
    b.wide = ctx->d_wide; b.chunk = ctx->d_chunk; b.shade = ctx->d_shade; b.tri = ctx->d_tri_sorted;
    b.links = ctx->d_links;
    const bool full = (flags & PT_BUILD_MORTON_FULL) != 0;
    const bool chained = full && sort_chained_applies(ctx, ctx->n, 8);
    if (int rc = run_build(ctx, b, reinterpret_cast<const uint4 *>(ctx->d_prims), ctx->n, flags, ctx->ev, &ctx->sorted_buf, chained)) return rc;
    ctx->morton_bits = full ? 64 : 32;
    ctx->sort_passes = full ? 8 : 2;
    ctx->boxes_valid = !build_keeps_centres_only(flags);
    ctx->tree_height = 0;
    if (flags & PT_BUILD_TOPOLOGY_MORTON) {
        // the traversal stacks are sized by the height of the tree, which only the finished tree tells
        int h = 0;
        unsigned overflow = 0;
        PT_HIP(ctx, hipMemcpyAsync(&h, &ctx->d_wide[0].pad[0], sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        PT_HIP(ctx, hipMemcpyAsync(&overflow, ctx->d_digit_total + 8 * 256, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (overflow) h = PT_MAX_TREE_HEIGHT + 1;           // more border-crossing nodes in a chunk than any accepted tree has
        if (h < 1 || h > PT_MAX_TREE_HEIGHT)
            PT_FAIL(ctx, PT_ERR_UNSUPPORTED, "pt_build: PT_BUILD_TOPOLOGY_MORTON gave a tree higher than the traversal stacks are sized for (height " +
                    std::to_string(h) + ", limit " + std::to_string(PT_MAX_TREE_HEIGHT) + "); build without the flag");
        ctx->tree_height = h;
    }
    // the per-leaf records of the trace kernels were written by the refit (unless PT_BUILD_TREE_ONLY); what
    // is left is the root's own box
    if (int rc = ptk_root_box(ctx)) return rc;
    ctx->trace_records = (flags & PT_BUILD_TREE_ONLY) == 0;
    PT_HIP(ctx, hipEventRecord(ctx->ev[PT_STAGE_PACK + 1], ctx->stream));
    if (chained) {
        // a workgroup of a chained pass that gave up waiting leaves a wrong order behind: the build has failed
        // (read back after the last stage event, so that the host's round trip is in no stage time)
        unsigned gave_up = 0;
        PT_HIP(ctx, hipMemcpyAsync(&gave_up, ctx->d_digit_total + CH_ERROR_WORD, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (gave_up) PT_FAIL(ctx, PT_ERR_HIP, "pt_build: a sorting workgroup gave up waiting for the tiles before it (PTAMD_SORT_CHAINED=0 selects the two-read passes)");
    }
    return PT_OK;
}

// ---------------------------------------------------------------------------
// Build without any allocation: every intermediate buffer of the build comes out of ONE static
// device array, for callers that cannot call hipMalloc (code under HIP graph capture, tight
// serving loops that rebuild a small scene every frame).  Each GPU the library runs on holds its
// own copy of the array, at its own address, until the process ends; host code must never use the
// array's NAME as an address (it is a host-side stand-in): the address is asked for once per
// device by ptk_scratch_init, kept in the context, and handed to the kernels as an argument like
// any other buffer.  All calls on one GPU share the array: calls that overlap on one GPU race.
// ---------------------------------------------------------------------------
static __device__ __align__(256) unsigned char scratch[PT_NOALLOC_SCRATCH_BYTES];

int ptk_scratch_init(pt_context *ctx)
{
    // the device of the context is current here (pt_create has just selected it)
    void *p = nullptr;
    PT_HIP(ctx, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) PT_FAIL(ctx, PT_ERR_HIP, "hipGetSymbolAddress(scratch) returned a null address");
    ctx->d_scratch = static_cast<unsigned char *>(p);
    return PT_OK;
}

namespace {
// widens 32-bit codes into the reference's uint64 morton_code
__global__ __launch_bounds__(256) void k_widen_codes(const uint32_t *__restrict__ in, int n, uint64_t *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}
}  // namespace

extern "C" int pt_build_tree_noalloc(pt_context *ctx, const void *dev_prims, int n, unsigned flags,
                                     uint64_t *dev_morton, int32_t *dev_perm, pt_TreeNode *dev_tree)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!dev_prims || !dev_perm || !dev_tree || n < 1)
        PT_FAIL(ctx, PT_ERR_INVALID, "pt_build_tree_noalloc: need primitives, n >= 1, dev_perm and dev_tree");
    if (n > PT_NOALLOC_MAX_PRIMS)
        PT_FAIL(ctx, PT_ERR_UNSUPPORTED, "pt_build_tree_noalloc: n exceeds PT_NOALLOC_MAX_PRIMS (" + std::to_string(PT_NOALLOC_MAX_PRIMS) +
                "); use pt_build");
    if (flags & ~PT_BUILD_KARRAS_LINKS) PT_FAIL(ctx, PT_ERR_INVALID, "pt_build_tree_noalloc: only PT_BUILD_KARRAS_LINKS is accepted");
    if ((reinterpret_cast<uintptr_t>(dev_prims) & 15u) || (reinterpret_cast<uintptr_t>(dev_tree) & 15u) ||
        (reinterpret_cast<uintptr_t>(dev_perm) & 3u) || (reinterpret_cast<uintptr_t>(dev_morton) & 7u))
        PT_FAIL(ctx, PT_ERR_INVALID, "pt_build_tree_noalloc: dev_prims and dev_tree must be 16-byte aligned, dev_morton 8, dev_perm 4");
    if (!ctx->d_scratch) PT_FAIL(ctx, PT_ERR_INVALID, "pt_build_tree_noalloc: no scratch address for this device");
    PT_HIP(ctx, hipSetDevice(ctx->device));

    // carve the scratch array (256-byte steps); sizes for the largest n are checked at compile time below
    unsigned char *p = ctx->d_scratch;
    auto take = [&](size_t bytes) { unsigned char *q = p; p += (bytes + 255) & ~(size_t)255; return q; };
    const size_t nblocks = ((size_t)n + PT_SORT_TILE - 1) / PT_SORT_TILE;
    const size_t nchunks = ((size_t)n + 1 + PT_REFIT_CHUNK - 1) / PT_REFIT_CHUNK;
    BuildBufs b;
    b.boxes = reinterpret_cast<BoxRec *>(take((size_t)n * sizeof(BoxRec)));
    b.wide = reinterpret_cast<WideNode *>(take((size_t)n * sizeof(WideNode)));
    b.morton = take((size_t)n * 4);
    b.keys[0] = take((size_t)n * 4);
    b.keys[1] = nullptr;                                     // the second (= last) pass writes no keys
    b.vals[0] = reinterpret_cast<uint32_t *>(take((size_t)n * 4));
    b.vals[1] = reinterpret_cast<uint32_t *>(dev_perm);      // two passes: the permutation ends in vals[1]
    b.bbox_partial = reinterpret_cast<float *>(take(6 * (((size_t)n + 255) / 256) * sizeof(float)));
    b.bbox_enc = reinterpret_cast<unsigned *>(take(64));
    b.block_hist = reinterpret_cast<uint32_t *>(take(256 * nblocks * sizeof(uint32_t)));
    b.digit_total = reinterpret_cast<uint32_t *>(take((8 * 256 + 512) * sizeof(uint32_t)));   // + the spare words every run_build caller provides
    b.chunk = reinterpret_cast<ChunkBox *>(take((nchunks + (nchunks + 2047) / 2048 + 1) * sizeof(ChunkBox)));
    if ((size_t)(p - ctx->d_scratch) > (size_t)PT_NOALLOC_SCRATCH_BYTES)
        PT_FAIL(ctx, PT_ERR_UNSUPPORTED, "pt_build_tree_noalloc: scratch array too small for n");
    // Nothing in the scratch array has to start at zero: every word a kernel reads was written by an
    // earlier kernel of this same call (boxes, partial boxes, codes, digit counts and totals, chunk
    // boxes, node records), so the previous call's contents are never observed.
    int sorted_buf = 0;
    if (int rc = run_build(ctx, b, static_cast<const uint4 *>(dev_prims), n, PT_BUILD_TREE_ONLY, nullptr, &sorted_buf)) return rc;
    if (sorted_buf != 1) PT_FAIL(ctx, PT_ERR_INVALID, "pt_build_tree_noalloc: unexpected sort buffer parity");
    hipStream_t st = ctx->stream;
    if (dev_morton)
        hipLaunchKernelGGL(k_widen_codes, dim3((n + 255) / 256), dim3(256), 0, st, static_cast<const uint32_t *>(b.morton), n, dev_morton);
    const int top = bit_length((unsigned)n);
    hipLaunchKernelGGL(k_export_leaves, dim3((n + 255) / 256), dim3(256), 0, st, b.vals[1], b.boxes, n, dev_tree);
    const long long nb1 = ((long long)n + 2) >> 1;
    hipLaunchKernelGGL(k_export_nodes, dim3((unsigned)((nb1 + 255) / 256), (unsigned)top), dim3(256), 0, st, n, top, b.wide, dev_tree);
    if (flags & PT_BUILD_KARRAS_LINKS)
        hipLaunchKernelGGL(k_radix_tree, dim3((n + 255) / 256), dim3(256), 0, st, n, dev_tree);
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

// bytes the carve-up above takes for the largest n (each piece rounded up to 256)
namespace {
constexpr size_t r256(size_t b) { return (b + 255) & ~(size_t)255; }
constexpr size_t noalloc_need(size_t n)
{
    return r256(n * 32) + r256(n * 64) + 3 * r256(n * 4) + r256(6 * ((n + 255) / 256) * 4) + 256 +
           r256(256 * ((n + PT_SORT_TILE - 1) / PT_SORT_TILE) * 4) + r256((8 * 256 + 512) * 4) +
           r256((((n + 1 + 1023) / 1024) + ((n + 1 + 1023) / 1024 + 2047) / 2048 + 1) * sizeof(ChunkBox));
}
static_assert(noalloc_need(PT_NOALLOC_MAX_PRIMS) <= PT_NOALLOC_SCRATCH_BYTES, "scratch array too small for PT_NOALLOC_MAX_PRIMS");
}  // namespace

// Fills ctx->d_tree (2n nodes in the reference's layout).  With PT_BUILD_KARRAS_LINKS the links are
// then rewritten by the search-based kernel that follows radix_build_cs.hlsl: the same values.
// The exports read the per-triangle boxes, which a default build does not keep: made here on first use.
static int ensure_boxes(pt_context *ctx)
{
    if (ctx->boxes_valid) return PT_OK;
    hipLaunchKernelGGL(k_bounds<false>, dim3((ctx->n + 255) / 256), dim3(256), 0, ctx->stream, reinterpret_cast<const uint4 *>(ctx->d_prims), ctx->n,
                       ctx->d_boxes, ctx->d_bbox_partial);
    PT_HIP(ctx, hipGetLastError());
    ctx->boxes_valid = true;
    return PT_OK;
}

int ptk_export_tree(pt_context *ctx)
{
    const int n = ctx->n;
    hipStream_t st = ctx->stream;
    if (int rc = ensure_boxes(ctx)) return rc;
    const int top = bit_length((unsigned)n);
    if (ctx->build_flags & PT_BUILD_TOPOLOGY_MORTON) {
        // n - 1 internal nodes: slots n - 1 and 2n stay zero
        PT_HIP(ctx, hipMemsetAsync(ctx->d_tree, 0, (2 * (size_t)n + 1) * sizeof(pt_TreeNode), st));
        hipLaunchKernelGGL(k_export_leaves_codes, dim3((n + 255) / 256), dim3(256), 0, st, ctx->d_vals[ctx->sorted_buf], ctx->d_boxes, n, ctx->d_tree);
        hipLaunchKernelGGL(k_export_nodes_codes, dim3((n + 254) / 256), dim3(256), 0, st, n, ctx->d_wide, ctx->d_tree);
        hipLaunchKernelGGL(k_export_parents_codes, dim3((n + 254) / 256), dim3(256), 0, st, n, ctx->d_wide, ctx->d_tree);
        PT_HIP(ctx, hipGetLastError());
        return PT_OK;
    }
    PT_HIP(ctx, hipMemsetAsync(&ctx->d_tree[2 * (size_t)n], 0, sizeof(pt_TreeNode), st));
    hipLaunchKernelGGL(k_export_leaves, dim3((n + 255) / 256), dim3(256), 0, st, ctx->d_vals[ctx->sorted_buf], ctx->d_boxes, n, ctx->d_tree);
    const long long nb1 = ((long long)n + 2) >> 1;        // blocks of level 1, the widest level
    hipLaunchKernelGGL(k_export_nodes, dim3((unsigned)((nb1 + 255) / 256), (unsigned)top), dim3(256), 0, st, n, top, ctx->d_wide, ctx->d_tree);
    if (ctx->build_flags & PT_BUILD_KARRAS_LINKS)
        hipLaunchKernelGGL(k_radix_tree, dim3((n + 255) / 256), dim3(256), 0, st, n, ctx->d_tree);
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

int ptk_export_sorted(pt_context *ctx, pt_PrimitiveRayCS *d_out)
{
    if (int rc = ensure_boxes(ctx)) return rc;
    hipLaunchKernelGGL(k_export_sorted, dim3((ctx->n + 255) / 256), dim3(256), 0, ctx->stream,
                       ctx->d_vals[ctx->sorted_buf], ctx->d_boxes, ctx->n, d_out);
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}
