// Internal declarations shared by the HIP translation units of libptamd.
// Device-side layouts here are private to the library; the interchange layouts
// are in include/ptamd_structs.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/ptamd.h"
#include "../../include/ptamd_math.h"

// ---- traversal layout -------------------------------------------------------
// One record per INTERNAL node: the boxes of both children and the child ids.
// A child id >= 0 is an internal node; a negative id is a leaf holding
// primitive ~id.  The out-of-range leaf 2n of the reference tree (zero box,
// primitive 0) is stored as id ~0 with the all-zero box.
struct __attribute__((aligned(16))) WideNode {
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    int32_t left, right;
    int32_t pad[2];
};
static_assert(sizeof(WideNode) == 64, "WideNode is four 16-byte loads");

// Per-leaf triangle record for intersection, in leaf-key order: vertex a and the two
// edges (b-a, c-a are pure functions of the vertices, so precomputing them changes no
// result), and the primitive id.
struct __attribute__((aligned(16))) TriIsect {
    float a[3];  int32_t prim;
    float e1[3]; float pad1;
    float e2[3]; float pad2;
};
static_assert(sizeof(TriIsect) == 48, "TriIsect is three 16-byte loads");

// Per-leaf triangle record for shading, in leaf-key order: the three vertex normals and the material.
struct __attribute__((aligned(16))) TriShade {
    float n0[3]; int32_t matid;
    float n1[3]; float pad1;
    float n2[3]; float pad2;
};
static_assert(sizeof(TriShade) == 48, "TriShade is three 16-byte loads");

// Unsorted per-primitive box, same bytes as pt_PrimitiveRayCS {min, 0, max, index}
struct __attribute__((aligned(16))) BoxRec {
    float mn[3]; float pad;
    float mx[3]; int32_t index;
};
static_assert(sizeof(BoxRec) == 32, "BoxRec");

struct ChunkBox { float mn[3]; float mx[3]; };

struct DevCounters {
    unsigned long long paths, rays, node_visits, box_tests, tri_tests;
    // wave-scheduling statistics of the instrumented trace kernel (diagnostics only):
    // iterations, lanes expanding a node, triangle-test rounds, lanes testing, lanes waiting at
    // a leaf, refill rounds, lanes waiting for a refill, lanes without a pixel
    unsigned long long sched[10];     // + clock ticks of a wave inside refill rounds, inside the traversal loop
};

constexpr int PT_SORT_THREADS = 256;
constexpr int PT_SORT_ITEMS = 16;
constexpr int PT_SORT_TILE = PT_SORT_THREADS * PT_SORT_ITEMS;   // 4096 keys per workgroup
constexpr int PT_REFIT_CHUNK_LOG2 = 10;
constexpr int PT_REFIT_CHUNK = 1 << PT_REFIT_CHUNK_LOG2;        // 1024 sorted keys per workgroup

constexpr int PT_SEED_SLOTS = 4;
// most internal streams that short trace calls are spread over (see pt_context::PipeSlot)
constexpr int PT_PIPE_SLOTS_MAX = 4;
// highest Morton-keyed tree the traversal stacks are sized for (PT_BUILD_TOPOLOGY_MORTON)
constexpr int PT_MAX_TREE_HEIGHT = 64;
// size of the library's static scratch array (pt_build_tree_noalloc): what PT_NOALLOC_MAX_PRIMS needs
constexpr size_t PT_NOALLOC_SCRATCH_BYTES = (size_t)14 << 20;

// Device buffers of one build.  pt_build fills this from the context's own allocations;
// pt_build_tree_noalloc from the caller's outputs and the library's static scratch array.
struct BuildBufs {
    BoxRec *boxes = nullptr;            // n
    float *bbox_partial = nullptr;      // 6 per 256 primitives
    unsigned *bbox_enc = nullptr;       // 6 (+ padding)
    void *keys[2] = {nullptr, nullptr}; // sort ping-pong, 4 or 8 bytes per key
    uint32_t *vals[2] = {nullptr, nullptr};
    void *morton = nullptr;             // n codes in primitive order, 4 or 8 bytes each
    uint32_t *block_hist = nullptr;     // 256 per PT_SORT_TILE keys
    uint32_t *digit_total = nullptr;    // 8 * 256, followed by 512 spare words (PT_BUILD_TOPOLOGY_MORTON: a counter)
    int2 *links = nullptr;              // PT_BUILD_TOPOLOGY_MORTON only: n {left child, flags}
    WideNode *wide = nullptr;           // n
    ChunkBox *chunk = nullptr;          // per 1024 keys, plus the upper levels' groups
    TriShade *shade = nullptr;          // n + 1 (unused with PT_BUILD_TREE_ONLY)
    TriIsect *tri = nullptr;            // n + 1 (unused with PT_BUILD_TREE_ONLY)
};

// Tuning knobs for experiments, read from the environment ONCE, at pt_create (PTAMD_* variables;
// unset = the defaults the measurements in this repository were taken with).
struct Knobs {
    int refill_t = 20, leaf_t = 12;     // PTAMD_REFILL_T, PTAMD_LEAF_T
    int wf_refill_t = 16, wf_leaf_t = 12;
    int samples_gib = 0;                // PTAMD_SAMPLES_GIB: cap of the sample staging buffer
    int no_lds_tree = 0;                // PTAMD_NO_LDS_TREE: node records from global memory even when they fit
    int max_threads = 0;                // PTAMD_MAX_THREADS: fewer waves per CU
    int no_order = 0;                   // PTAMD_NO_ORDER: image order instead of most-expensive-first
    int chunk = 0;                      // PTAMD_CHUNK: frames per work item
    int print_sched = 0;                // PTAMD_PRINT_SCHED: scheduling statistics on stderr
    int dual = 1;                       // PTAMD_DUAL: two paths per lane for trees in LDS (k_trace_dual); 0: k_trace_persistent
    int dual_min_chunk = 1;             // PTAMD_DUAL_MIN_CHUNK: k_trace_dual is used when a launch cut into sixteen items per lane has items of at least this many frames (1: always)
    int seeds_ahead = 1;                // PTAMD_SEEDS_AHEAD: path seeds of long calls on trees in LDS computed on an internal stream while the previous trace kernel runs (0: in line)
    int dual_refill_t = 8, swap_t = 4, shade_t = 48, dual_leaf_t = 8;   // PTAMD_DUAL_REFILL_T, PTAMD_SWAP_T, PTAMD_SHADE_T, PTAMD_DUAL_LEAF_T: its wave-vote thresholds
    int no_top_levels = 0;              // PTAMD_NO_TOP_LEVELS: trees in global memory without their upper levels in LDS
    int gen_lds_entries = 0;            // PTAMD_GEN_LDS_ENTRIES: cap of the LDS part of the explicit traversal stack (0: what fits)
    int pair_min_mib = 24;              // PTAMD_PAIR_MIN_MIB: child lines only when the plain records take at least this much
    int pair_max_mib = 65536;           // PTAMD_PAIR_MAX_MIB: largest array of child lines (0: never use it)
    int extra_wave = 16;                // PTAMD_EXTRA_WAVES: most waves per CU that k_trace_dual runs beyond those whose stacks fit in LDS beside the tree (their stacks: global memory); 0: none
    int fine_items = 12, fine_chunk = 0; // PTAMD_FINE_ITEMS / PTAMD_FINE_CHUNK: k_trace_dual ends a launch on about fine_items short items per lane (0: none) of fine_chunk frames each (0: a sixteenth of the launch's frames, 2 .. 8)
    int coarse_items = 4;               // PTAMD_COARSE_ITEMS: ... and hands the frames before those out in about this many long items per lane
    int sort_chained = 1;               // PTAMD_SORT_CHAINED: sorts of more than two passes read their keys once per pass (0: separate counting pass per pass)
    // Overlapped short calls, measured on scene04 with one frame per call (scripts/gpu_short_calls.py): three streams (with the
    // context's own that makes the four hardware queues a process has; a fourth stream shares a queue and loses), each launch over
    // 7/16 of the CUs, so that two launches fill the chip and the third takes the CUs the first one frees: 2.48 -> 3.33 Grays/s
    int pipe_slots = 3;                 // PTAMD_PIPE_SLOTS: internal streams that overlapped short trace calls use in turn (2 .. PT_PIPE_SLOTS_MAX)
    int short_grid = 0;                 // PTAMD_SHORT_GRID: workgroups of an overlapped short call's launch (0: 7/16 of the CUs, or by the divisor)
    int short_grid_div = 0;             // PTAMD_SHORT_GRID_DIV: ... one workgroup per this many CUs (0: unset)
};

struct pt_context {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // scene
    int n = 0;
    const pt_PrimitiveRender *d_prims = nullptr;
    bool prims_owned = false;
    pt_CB_Radix cb{};
    bool have_cb = false;
    pt_CB_Radix *d_cb = nullptr;      // followed on the device by one all-zero pt_Material (ids outside 0..15)

    // build products
    bool built = false;
    unsigned build_flags = 0;
    int capacity = 0;                 // primitives the buffers below were sized for
    BoxRec *d_boxes = nullptr;        // n
    bool boxes_valid = false;         // false after a build that kept only the boxes' centres there
    unsigned *d_bbox_enc = nullptr;   // 6 order-preserving encodings + padding
    float *d_bbox_partial = nullptr;  // 6 floats per bounds workgroup
    uint64_t *d_keys[2] = {nullptr, nullptr};
    uint32_t *d_vals[2] = {nullptr, nullptr};
    uint64_t *d_morton = nullptr;     // n codes in primitive order (kept for download), 32- or 64-bit
    int morton_bits = 64;
    int sorted_buf = 0;               // which of d_keys/d_vals holds the sorted result
    uint32_t *d_block_hist = nullptr; // 256 * nblocks
    uint32_t *d_digit_total = nullptr;// 8 * 256 + 512
    int2 *d_links = nullptr;          // n, allocated by the first PT_BUILD_TOPOLOGY_MORTON build
    size_t links_cap = 0;
    int tree_height = 0;              // PT_BUILD_TOPOLOGY_MORTON: internal nodes on the longest root-to-leaf path
    uint2 *d_gstack = nullptr;        // traversal-stack entries beyond the LDS part (Morton-keyed trees deeper than that)
    size_t gstack_cap = 0;            // entries
    float *d_xstack = nullptr;        // stack entries of k_trace_dual's waves beyond the LDS stacks: 16 waves x 32 levels x 64 lanes per CU
    pt_TreeNode *d_tree = nullptr;    // 2n+1, allocated and filled by the first pt_download_tree after a build
    bool tree_exported = false;
    WideNode *d_wide = nullptr;       // n
    float4 *d_pair = nullptr;         // 8 float4 per node (the records of its two children), made by the first trace of a
                                      // tree that stays in global memory
    size_t pair_cap = 0;
    bool pair_valid = false;
    ChunkBox *d_chunk = nullptr;      // ceil((n+1)/1024)
    float *d_root = nullptr;          // the root's own box (6 floats)
    TriShade *d_shade = nullptr;      // n+1, leaf-key order
    TriIsect *d_tri_sorted = nullptr; // n+1, leaf-key order
    bool trace_records = false;       // the two above were written by the last build (not PT_BUILD_TREE_ONLY)
    unsigned *d_work = nullptr;       // pixel queue head of the persistent trace kernel
    int num_cus = 256;
    int sort_passes = 0;
    hipEvent_t ev[PT_STAGE_COUNT + 1] = {};
    bool have_events = false;

    // images
    int W = 0, H = 0;
    uint8_t *d_diffuse = nullptr;     // RGBA8
    float4 *d_normal = nullptr;       // xyz + asfloat(matid+1)
    float4 *d_world = nullptr;
    float4 *d_output = nullptr;       // gamma-space running average (`output` / `old`)
    uint8_t *d_display = nullptr;     // RGBA8 (`output_u`)
    float4 *d_linear = nullptr;       // linear sum, alpha = sample count
    bool have_gbuffer = false;
    // work-queue order of the persistent trace kernel: per-pixel cost (decaying sum of the
    // traversal iterations of earlier launches) and the pixels sorted by it
    uint32_t *d_cost = nullptr;       // W*H
    uint32_t *d_pixel_work = nullptr; // W*H, iterations of the current launch
    float4 *d_samples = nullptr;      // staging of one launch: [W*H][frames] slots of 16 bytes (a path's seed in .w, then its radiance sample)
    uint32_t samples_frames = 0;
    size_t samples_cap_bytes = 0;     // upper bound of the staging buffer: an eighth of the device's memory
    // wavefront trace: two path queues (48-byte records), hit records (16 bytes), queue lengths
    void *d_wf_queue[2] = {nullptr, nullptr};
    void *d_wf_hits = nullptr;
    size_t wf_paths = 0;              // paths the three buffers above were sized for
    unsigned *d_wf_counts = nullptr;  // 32 x 64 counters, 128 bytes apart
    uint32_t *d_okeys[2] = {nullptr, nullptr};
    uint32_t *d_ovals[2] = {nullptr, nullptr};   // [0] holds the order when order_valid
    uint32_t *d_ohist = nullptr;      // 256 * ceil(W*H / PT_SORT_TILE)
    uint32_t *d_ototal = nullptr;     // 2 * 256
    bool order_valid = false;
    uint32_t frame_count = 0;
    uint32_t *d_seeds = nullptr;
    uint32_t *h_seeds[PT_SEED_SLOTS] = {};   // pinned host staging, used in rotation
    hipEvent_t seed_ev[PT_SEED_SLOTS] = {};  // recorded after the copy out of the slot
    bool seed_ev_used[PT_SEED_SLOTS] = {};
    int seed_slot = 0;
    uint32_t seeds_cap = 0;

    // Overlap of short consecutive trace calls (one frame per call is the reference's own run loop): such a call's seeds
    // copy, seed, trace and accumulate kernels run on one of a few internal streams, used in turn, with the buffers of that
    // slot, so that the next calls' paths fill the CUs that the previous calls' last paths leave idle (a kernel on a stream
    // starts only when EVERY workgroup of the stream's previous kernel has ended, so each stream more lets one more
    // call's workgroups take the CUs as they become free).  Accumulates are
    // chained through the slots' `done` events (the image recurrence), and the context's own stream waits for every
    // slot's `done` as soon as it is recorded, so everything else stays ordered with the context's stream as before.
    struct PipeSlot {
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        bool used = false;            // `done` has been recorded at least once
        float4 *d_samples = nullptr;
        uint32_t samples_frames = 0;
        unsigned *d_work = nullptr;
        uint32_t *d_seeds = nullptr;
        uint32_t seeds_cap = 0;
    };
    PipeSlot pipe[PT_PIPE_SLOTS_MAX];  // knobs.pipe_slots of them are used
    hipEvent_t pipe_fork = nullptr;
    int pipe_next = 0;                // slot of the next overlapped call
    int pipe_active = -1;             // slot whose buffers are swapped into the context while a call is enqueued
    bool main_dirty = true;           // work has been enqueued on the context's stream since the last fork
    bool user_stream = false;         // pt_set_stream gave a stream of the caller's: no overlap (its own kernels are not seen here)
    int overlap = 1;                  // pt_set_trace_overlap
    uint32_t h_seeds_cap = 0;         // seeds the pinned staging buffers hold
    // Seeds ahead: the path seeds of a long call (16-round hash per path, no memory to speak of) are computed on a stream of
    // the library's own, into a plain array, while the PREVIOUS call's trace kernel still runs: that kernel's waves leave a
    // SIMD's vector pipe idle about a third of the time and 32 registers per lane free, which is room for one wave of the
    // seed kernel per SIMD.  Two sets of buffers, used in turn; the trace kernel reads the array instead of the slots' .w.
    struct SeedsAhead {
        hipStream_t stream = nullptr;
        uint32_t *d_frame_seeds[2] = {nullptr, nullptr}; uint32_t frame_cap[2] = {0, 0};
        uint32_t *d_path_seeds[2] = {nullptr, nullptr}; size_t path_cap[2] = {0, 0};
        hipEvent_t ready[2] = {nullptr, nullptr};        // recorded on `stream` behind the seed kernel
        hipEvent_t released[2] = {nullptr, nullptr};     // recorded on the context's stream behind the trace kernel that read the set
        bool released_used[2] = {false, false};
        int next = 0;
    } ahead;
    const uint32_t *path_seeds = nullptr;   // set while a call whose seeds were computed ahead is being enqueued

    // options / counters
    int stack_entries = 0;            // 0 = automatic
    int schedule = 0;                 // PT_SCHEDULE_*
    int itr_limit = 15;               // a path ends when itr reaches it: max_segments + 1 (reference: MAX_ITR * 3)
    bool detailed = false;
    bool time_trace = false;          // record an event pair around the trace kernel of each launch
    std::vector<std::pair<hipEvent_t, hipEvent_t>> trace_events;
    std::vector<hipEvent_t> event_pool;   // timing events waiting for reuse
    Knobs knobs;
    int lds_attr[32] = {};             // dynamic-LDS limit already set per kernel variant (bytes)
    int near_first = 0;               // PT_TRAVERSAL_*
    unsigned char *d_scratch = nullptr;   // this device's copy of the library's static scratch array
    void *rccl_comm = nullptr;        // communicator made by pt_comm_init_rank (owned)
    DevCounters *d_counters = nullptr;
};

#define PT_FAIL(ctx, code, msg) do { (ctx)->err = (msg); return (code); } while (0)
#define PT_HIP(ctx, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_); return PT_ERR_HIP; } } while (0)

// ---- launchers implemented in the kernel translation units ------------------
int ptk_build(pt_context *ctx, unsigned flags);
int ptk_scratch_init(pt_context *ctx);
int ptk_root_box(pt_context *ctx);
int ptk_order_pixels(pt_context *ctx, int shift);
int ptk_export_sorted(pt_context *ctx, pt_PrimitiveRayCS *d_out);
int ptk_export_tree(pt_context *ctx);
int ptk_primary(pt_context *ctx, const pt_camera *cam, const uint8_t *d_albedo);
int ptk_trace_frames(pt_context *ctx, uint32_t first_frame, uint32_t count, int mode, const int *region);
bool ptk_seeds_ahead_ok(const pt_context *ctx, uint32_t count);
int ptk_path_seeds_ahead(pt_context *ctx, hipStream_t stream, uint32_t count, const uint32_t *d_frame_seeds, uint32_t *d_path_seeds);
bool ptk_tree_in_lds(const pt_context *ctx);
bool ptk_short_calls_may_overlap(const pt_context *ctx);
int ptk_resolve(pt_context *ctx);
int ptk_intersect(pt_context *ctx, const float *d_rays8, int n_rays, pt_hit *d_out);
