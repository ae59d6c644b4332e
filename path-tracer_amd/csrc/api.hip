// C ABI of libptamd (include/ptamd.h): context, uploads, build, trace, downloads.
// Host-side only; the kernels live in build_kernels.hip / trace_kernels.hip.
#include "pt_internal.hpp"

#include <utility>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <dlfcn.h>

static std::string g_create_error;

// an environment knob as an integer, clamped to the range its users are written for
static int env_int(const char *name, int fallback, int lo, int hi)
{
    const char *e = getenv(name);
    if (!e || !*e) return fallback;
    const long v = strtol(e, nullptr, 10);
    return v < lo ? lo : (v > hi ? hi : (int)v);
}

// PTAMD_* experiment knobs: read here, once per context, never on the launch path
static void read_knobs(Knobs &k)
{
    // wave-vote thresholds count lanes of one wave
    k.refill_t = env_int("PTAMD_REFILL_T", k.refill_t, 1, 64);
    k.leaf_t = env_int("PTAMD_LEAF_T", k.leaf_t, 1, 64);
    k.wf_refill_t = env_int("PTAMD_WF_REFILL_T", k.wf_refill_t, 1, 64);
    k.wf_leaf_t = env_int("PTAMD_WF_LEAF_T", k.wf_leaf_t, 1, 64);
    k.samples_gib = env_int("PTAMD_SAMPLES_GIB", 0, 0, 256);
    k.no_lds_tree = env_int("PTAMD_NO_LDS_TREE", 0, 0, 1);
    k.max_threads = env_int("PTAMD_MAX_THREADS", 0, 0, 1024);
    k.no_order = env_int("PTAMD_NO_ORDER", 0, 0, 1);
    k.chunk = env_int("PTAMD_CHUNK", 0, 0, 1 << 20);
    k.print_sched = env_int("PTAMD_PRINT_SCHED", 0, 0, 1);
    k.pair_max_mib = env_int("PTAMD_PAIR_MAX_MIB", k.pair_max_mib, 0, 1 << 20);
    k.pair_min_mib = env_int("PTAMD_PAIR_MIN_MIB", k.pair_min_mib, 0, 1 << 20);
    k.gen_lds_entries = env_int("PTAMD_GEN_LDS_ENTRIES", 0, 0, PT_MAX_TREE_HEIGHT);
    k.no_top_levels = env_int("PTAMD_NO_TOP_LEVELS", 0, 0, 1);
    k.dual = env_int("PTAMD_DUAL", k.dual, 0, 1);
    k.dual_min_chunk = env_int("PTAMD_DUAL_MIN_CHUNK", k.dual_min_chunk, 1, 1 << 20);
    k.dual_refill_t = env_int("PTAMD_DUAL_REFILL_T", k.dual_refill_t, 1, 64);
    k.swap_t = env_int("PTAMD_SWAP_T", k.swap_t, 1, 64);
    k.shade_t = env_int("PTAMD_SHADE_T", k.shade_t, 1, 64);
    k.dual_leaf_t = env_int("PTAMD_DUAL_LEAF_T", k.dual_leaf_t, 1, 64);
    k.sort_chained = env_int("PTAMD_SORT_CHAINED", k.sort_chained, 0, 1);
    k.fine_items = env_int("PTAMD_FINE_ITEMS", k.fine_items, 0, 1 << 20);
    k.seeds_ahead = env_int("PTAMD_SEEDS_AHEAD", k.seeds_ahead, 0, 1);
    k.coarse_items = env_int("PTAMD_COARSE_ITEMS", k.coarse_items, 1, 1 << 20);
    k.fine_chunk = env_int("PTAMD_FINE_CHUNK", k.fine_chunk, 0, 1 << 20);
    k.extra_wave = env_int("PTAMD_EXTRA_WAVES", k.extra_wave, 0, 16);      // a workgroup has at most 16 waves
    k.pipe_slots = env_int("PTAMD_PIPE_SLOTS", k.pipe_slots, 2, PT_PIPE_SLOTS_MAX);
    k.short_grid_div = env_int("PTAMD_SHORT_GRID_DIV", k.short_grid_div, 0, 8);
    k.short_grid = env_int("PTAMD_SHORT_GRID", k.short_grid, 0, 4096);
}

static int bit_length_u(unsigned v) { return v ? 32 - __builtin_clz(v) : 0; }

template <typename T>
static int dev_alloc(pt_context *ctx, T **p, size_t count)
{
    *p = nullptr;
    PT_HIP(ctx, hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T)));
    return PT_OK;
}
template <typename T>
static void dev_free(T *&p)
{
    if (p) (void)hipFree((void *)p);
    p = nullptr;
}

static void free_build(pt_context *ctx)
{
    dev_free(ctx->d_boxes);
    dev_free(ctx->d_bbox_partial);
    dev_free(ctx->d_keys[0]); dev_free(ctx->d_keys[1]);
    dev_free(ctx->d_vals[0]); dev_free(ctx->d_vals[1]);
    dev_free(ctx->d_morton);
    dev_free(ctx->d_block_hist);
    dev_free(ctx->d_tree);
    dev_free(ctx->d_wide);
    dev_free(ctx->d_pair); ctx->pair_cap = 0; ctx->pair_valid = false;
    dev_free(ctx->d_links); ctx->links_cap = 0;
    dev_free(ctx->d_gstack); ctx->gstack_cap = 0;
    dev_free(ctx->d_xstack);
    dev_free(ctx->d_chunk);
    dev_free(ctx->d_shade);
    dev_free(ctx->d_tri_sorted);
    ctx->capacity = 0;
    ctx->built = false;
}

static void free_images(pt_context *ctx)
{
    dev_free(ctx->d_diffuse); dev_free(ctx->d_normal); dev_free(ctx->d_world);
    dev_free(ctx->d_output); dev_free(ctx->d_display); dev_free(ctx->d_linear);
    dev_free(ctx->d_samples); ctx->samples_frames = 0;
    for (auto &sl : ctx->pipe) {        // sized by the image too
        if (sl.stream) (void)hipStreamSynchronize(sl.stream);
        dev_free(sl.d_samples); sl.samples_frames = 0;
    }
    dev_free(ctx->d_wf_queue[0]); dev_free(ctx->d_wf_queue[1]); dev_free(ctx->d_wf_hits); ctx->wf_paths = 0;
    dev_free(ctx->d_cost); dev_free(ctx->d_pixel_work); dev_free(ctx->d_okeys[0]); dev_free(ctx->d_okeys[1]);
    dev_free(ctx->d_ovals[0]); dev_free(ctx->d_ovals[1]); dev_free(ctx->d_ohist); dev_free(ctx->d_ototal);
    ctx->order_valid = false;
    ctx->W = ctx->H = 0;
    ctx->have_gbuffer = false;
}

// Every entry point that enqueues on the context's stream comes through here (overlapped trace calls excepted):
// the next overlapped call then makes its internal stream wait for that work first.
static int set_device(pt_context *ctx)
{
    PT_HIP(ctx, hipSetDevice(ctx->device));
    ctx->main_dirty = true;
    return PT_OK;
}

static void free_pipe(pt_context *ctx)
{
    for (auto &sl : ctx->pipe) {
        if (sl.stream) { (void)hipStreamSynchronize(sl.stream); (void)hipStreamDestroy(sl.stream); }
        if (sl.done) (void)hipEventDestroy(sl.done);
        dev_free(sl.d_samples); dev_free(sl.d_work); dev_free(sl.d_seeds);
        sl = pt_context::PipeSlot();
    }
    if (ctx->pipe_fork) (void)hipEventDestroy(ctx->pipe_fork);
    ctx->pipe_fork = nullptr;
}

extern "C" {

const char *pt_version(void) { return "ptamd 0.1 (gfx950)"; }

int pt_create(int device_ordinal, pt_context **out)
{
    if (!out) { g_create_error = "pt_create: out is NULL"; return PT_ERR_INVALID; }
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_create_error = std::string("pt_create: no HIP device (") + hipGetErrorString(e) + ")";
        return PT_ERR_NO_DEVICE;
    }
    if (device_ordinal < 0 || device_ordinal >= count) {
        g_create_error = "pt_create: device ordinal out of range";
        return PT_ERR_INVALID;
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_ordinal);
    if (e != hipSuccess) { g_create_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e); return PT_ERR_HIP; }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("pt_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        return PT_ERR_NO_DEVICE;
    }
    pt_context *ctx = new pt_context();
    ctx->device = device_ordinal;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->samples_cap_bytes = (size_t)prop.totalGlobalMem / 8;
    read_knobs(ctx->knobs);
    int rc = set_device(ctx);
    if (!rc) rc = ptk_scratch_init(ctx);      // this device's address of the static scratch array
    if (!rc) {
        // the constants, followed by one all-zero material for ids outside 0..15
        void *p = nullptr;
        hipError_t e2 = hipMalloc(&p, sizeof(pt_CB_Radix) + sizeof(pt_Material));
        if (e2 == hipSuccess) e2 = hipMemset(p, 0, sizeof(pt_CB_Radix) + sizeof(pt_Material));
        if (e2 != hipSuccess) { ctx->err = std::string("constants: ") + hipGetErrorString(e2); rc = PT_ERR_HIP; }
        ctx->d_cb = static_cast<pt_CB_Radix *>(p);
    }
    if (!rc) rc = dev_alloc(ctx, &ctx->d_bbox_enc, 16);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_root, 8);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_digit_total, 8 * 256 + 512);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_counters, 1);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_work, 64);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_wf_counts, 32 * 64 * 32);
    if (!rc) { hipError_t e2 = hipMemset(ctx->d_counters, 0, sizeof(DevCounters)); if (e2 != hipSuccess) rc = PT_ERR_HIP; }
    for (int i = 0; !rc && i <= PT_STAGE_COUNT; i++) {
        hipError_t e2 = hipEventCreate(&ctx->ev[i]);
        if (e2 != hipSuccess) { ctx->err = std::string("hipEventCreate: ") + hipGetErrorString(e2); rc = PT_ERR_HIP; }
    }
    if (rc) { g_create_error = ctx->err; pt_destroy(ctx); return rc; }
    ctx->have_events = true;
    *out = ctx;
    return PT_OK;
}

void pt_destroy(pt_context *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &e : ctx->trace_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->rccl_comm) (void)pt_comm_destroy(ctx);
    for (int k = 0; k < PT_SEED_SLOTS; k++) {
        if (ctx->h_seeds[k]) (void)hipHostFree(ctx->h_seeds[k]);
        if (ctx->seed_ev[k]) (void)hipEventDestroy(ctx->seed_ev[k]);
    }
    free_pipe(ctx);
    if (ctx->ahead.stream) { (void)hipStreamSynchronize(ctx->ahead.stream); (void)hipStreamDestroy(ctx->ahead.stream); }
    for (int b = 0; b < 2; b++) {
        dev_free(ctx->ahead.d_frame_seeds[b]); dev_free(ctx->ahead.d_path_seeds[b]);
        if (ctx->ahead.ready[b]) (void)hipEventDestroy(ctx->ahead.ready[b]);
        if (ctx->ahead.released[b]) (void)hipEventDestroy(ctx->ahead.released[b]);
    }
    free_build(ctx);
    free_images(ctx);
    if (ctx->prims_owned) { void *p = const_cast<pt_PrimitiveRender *>(ctx->d_prims); if (p) (void)hipFree(p); }
    dev_free(ctx->d_cb); dev_free(ctx->d_bbox_enc); dev_free(ctx->d_root); dev_free(ctx->d_digit_total);
    dev_free(ctx->d_counters); dev_free(ctx->d_seeds); dev_free(ctx->d_work); dev_free(ctx->d_wf_counts);
    for (int i = 0; i <= PT_STAGE_COUNT; i++) if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    delete ctx;
}

const char *pt_last_error(const pt_context *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

static void drop_trace_events(pt_context *ctx)
{   // the events go back to the pool: they are recorded again by later launches
    for (auto &e : ctx->trace_events) { ctx->event_pool.push_back(e.first); ctx->event_pool.push_back(e.second); }
    ctx->trace_events.clear();
}

int pt_set_stream(pt_context *ctx, void *hip_stream)
{
    if (!ctx) return PT_ERR_INVALID;
    hipStream_t next = reinterpret_cast<hipStream_t>(hip_stream);
    if (next == ctx->stream) return PT_OK;
    if (int rc = set_device(ctx)) return rc;
    // Work already queued on the old stream (image clears, constant and seed copies, a build, event
    // records) is not ordered with the new stream: drain it before switching.
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < PT_SEED_SLOTS; k++) ctx->seed_ev_used[k] = false;
    drop_trace_events(ctx);          // an event pair must not span two streams
    ctx->stream = next;
    ctx->user_stream = next != nullptr;
    return PT_OK;
}

int pt_synchronize(pt_context *ctx)
{
    if (!ctx) return PT_ERR_INVALID;
    if (int rc = set_device(ctx)) return rc;
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

// ---------------------------------------------------------------------------
static int set_primitives(pt_context *ctx, const void *dev, int n, bool owned)
{
    if (ctx->prims_owned && ctx->d_prims) (void)hipFree(const_cast<pt_PrimitiveRender *>(ctx->d_prims));
    ctx->d_prims = reinterpret_cast<const pt_PrimitiveRender *>(dev);
    ctx->prims_owned = owned;
    ctx->n = n;
    ctx->built = false;
    return PT_OK;
}

int pt_upload_primitives(pt_context *ctx, const pt_PrimitiveRender *host_prims, int n)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!host_prims || n < 1) PT_FAIL(ctx, PT_ERR_INVALID, "pt_upload_primitives: need n >= 1 primitives");
    if (n >= (1 << 30)) PT_FAIL(ctx, PT_ERR_UNSUPPORTED, "pt_upload_primitives: n must be below 2^30");
    // the constant buffer holds 16 materials (structs.hlsli: `Material mats[16]`)
    for (int i = 0; i < n; i++)
        if ((unsigned)host_prims[i].matid >= (unsigned)PT_MAX_MATERIALS)
            PT_FAIL(ctx, PT_ERR_UNSUPPORTED, "pt_upload_primitives: primitive " + std::to_string(i) + " has material id " +
                    std::to_string(host_prims[i].matid) + ", outside 0.." + std::to_string(PT_MAX_MATERIALS - 1));
    if (int rc = set_device(ctx)) return rc;
    void *d = nullptr;
    PT_HIP(ctx, hipMalloc(&d, (size_t)n * sizeof(pt_PrimitiveRender)));
    hipError_t e = hipMemcpyAsync(d, host_prims, (size_t)n * sizeof(pt_PrimitiveRender), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(d); ctx->err = std::string("upload primitives: ") + hipGetErrorString(e); return PT_ERR_HIP; }
    return set_primitives(ctx, d, n, true);
}

int pt_upload_primitives_device(pt_context *ctx, const void *dev_prims, int n)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!dev_prims || n < 1) PT_FAIL(ctx, PT_ERR_INVALID, "pt_upload_primitives_device: need n >= 1 primitives");
    if (n >= (1 << 30)) PT_FAIL(ctx, PT_ERR_UNSUPPORTED, "pt_upload_primitives_device: n must be below 2^30");
    if ((reinterpret_cast<uintptr_t>(dev_prims) & 15u) != 0)
        PT_FAIL(ctx, PT_ERR_INVALID, "pt_upload_primitives_device: pointer must be 16-byte aligned");
    return set_primitives(ctx, dev_prims, n, false);
}

int pt_set_constants(pt_context *ctx, const pt_CB_Radix *cb)
{
    if (!ctx || !cb) return PT_ERR_INVALID;
    if (int rc = set_device(ctx)) return rc;
    ctx->cb = *cb;
    ctx->have_cb = true;
    PT_HIP(ctx, hipMemcpyAsync(ctx->d_cb, &ctx->cb, sizeof(pt_CB_Radix), hipMemcpyHostToDevice, ctx->stream));
    return PT_OK;
}

// ---------------------------------------------------------------------------
static int ensure_build_buffers(pt_context *ctx)
{
    const int n = ctx->n;
    if (ctx->capacity >= n) return PT_OK;
    free_build(ctx);
    const size_t nblocks = ((size_t)n + PT_SORT_TILE - 1) / PT_SORT_TILE;
    const size_t nchunks = ((size_t)n + 1 + PT_REFIT_CHUNK - 1) / PT_REFIT_CHUNK;
    int rc = dev_alloc(ctx, &ctx->d_boxes, (size_t)n);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_bbox_partial, 6 * (((size_t)n + 255) / 256));
    if (!rc) rc = dev_alloc(ctx, &ctx->d_keys[0], (size_t)n);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_keys[1], (size_t)n);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_vals[0], (size_t)n);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_vals[1], (size_t)n);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_morton, (size_t)n);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_block_hist, 256 * nblocks);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_wide, (size_t)n);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_chunk, nchunks + (nchunks + 2047) / 2048 + 1);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_shade, (size_t)n + 1);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_tri_sorted, (size_t)n + 1);
    if (rc) { free_build(ctx); return rc; }
    ctx->capacity = n;
    return PT_OK;
}

int pt_build(pt_context *ctx, unsigned flags)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->d_prims || ctx->n < 1) PT_FAIL(ctx, PT_ERR_INVALID, "pt_build: no primitives uploaded");
    if (flags & ~(PT_BUILD_MORTON_FULL | PT_BUILD_KARRAS_LINKS | PT_BUILD_TREE_ONLY | PT_BUILD_TOPOLOGY_MORTON))
        PT_FAIL(ctx, PT_ERR_INVALID, "pt_build: unknown flag");
    if ((flags & PT_BUILD_TOPOLOGY_MORTON) && (flags & PT_BUILD_KARRAS_LINKS))
        PT_FAIL(ctx, PT_ERR_INVALID, "pt_build: PT_BUILD_KARRAS_LINKS belongs to the index-keyed tree");
    if ((flags & PT_BUILD_TOPOLOGY_MORTON) && ctx->n < 2) PT_FAIL(ctx, PT_ERR_UNSUPPORTED, "pt_build: PT_BUILD_TOPOLOGY_MORTON needs n >= 2");
    if (int rc = set_device(ctx)) return rc;
    if (int rc = ensure_build_buffers(ctx)) return rc;
    if ((flags & PT_BUILD_TOPOLOGY_MORTON) && ctx->links_cap < (size_t)ctx->n) {
        PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        dev_free(ctx->d_links);
        ctx->links_cap = 0;
        if (int rc = dev_alloc(ctx, &ctx->d_links, (size_t)ctx->capacity)) return rc;
        ctx->links_cap = (size_t)ctx->capacity;
    }
    ctx->built = false;
    ctx->tree_exported = false;
    ctx->pair_valid = false;
    if (int rc = ptk_build(ctx, flags)) return rc;
    ctx->build_flags = flags;
    ctx->built = true;
    return PT_OK;
}

int pt_build_times(pt_context *ctx, float stage_ms[PT_STAGE_COUNT], float *total_ms, int *sort_passes)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->built) PT_FAIL(ctx, PT_ERR_INVALID, "pt_build_times: nothing built");
    if (int rc = set_device(ctx)) return rc;
    PT_HIP(ctx, hipEventSynchronize(ctx->ev[PT_STAGE_COUNT]));
    float tot = 0;
    for (int i = 0; i < PT_STAGE_COUNT; i++) {
        float ms = 0;
        PT_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev[i], ctx->ev[i + 1]));
        if (stage_ms) stage_ms[i] = ms;
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (sort_passes) *sort_passes = ctx->sort_passes;
    return PT_OK;
}

#define PT_NEED_BUILT(ctx, who) do { if (!(ctx)) return PT_ERR_INVALID; \
    if (!(ctx)->built) PT_FAIL(ctx, PT_ERR_INVALID, who ": call pt_build first"); \
    if (int rc_ = set_device(ctx)) return rc_; } while (0)

static int d2h(pt_context *ctx, void *dst, const void *src, size_t bytes)
{
    PT_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_download_scene_bound(pt_context *ctx, pt_Bound *bb)
{
    PT_NEED_BUILT(ctx, "pt_download_scene_bound");
    if (!bb) return PT_ERR_INVALID;
    unsigned enc[6];
    if (int rc = d2h(ctx, enc, ctx->d_bbox_enc, sizeof(enc))) return rc;
    auto dec = [](unsigned u) { unsigned v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u; float f; std::memcpy(&f, &v, 4); return f; };
    std::memset(bb, 0, sizeof(*bb));
    for (int k = 0; k < 3; k++) { bb->min.v[k] = dec(enc[k]); bb->max.v[k] = dec(enc[3 + k]); }
    return PT_OK;
}

int pt_download_morton(pt_context *ctx, uint64_t *codes)
{
    PT_NEED_BUILT(ctx, "pt_download_morton");
    if (!codes) return PT_ERR_INVALID;
    if (ctx->morton_bits == 64) return d2h(ctx, codes, ctx->d_morton, (size_t)ctx->n * sizeof(uint64_t));
    // 32-bit codes on the device: copy into the upper half of the caller's buffer, widen in place
    uint32_t *narrow = reinterpret_cast<uint32_t *>(codes) + ctx->n;
    if (int rc = d2h(ctx, narrow, ctx->d_morton, (size_t)ctx->n * sizeof(uint32_t))) return rc;
    for (int i = 0; i < ctx->n; i++) codes[i] = narrow[i];
    return PT_OK;
}

int pt_download_permutation(pt_context *ctx, int32_t *perm)
{
    PT_NEED_BUILT(ctx, "pt_download_permutation");
    if (!perm) return PT_ERR_INVALID;
    return d2h(ctx, perm, ctx->d_vals[ctx->sorted_buf], (size_t)ctx->n * sizeof(uint32_t));
}

int pt_download_sorted_primitives(pt_context *ctx, pt_PrimitiveRayCS *out)
{
    PT_NEED_BUILT(ctx, "pt_download_sorted_primitives");
    if (!out) return PT_ERR_INVALID;
    pt_PrimitiveRayCS *d = nullptr;
    if (int rc = dev_alloc(ctx, &d, (size_t)ctx->n)) return rc;
    int rc = ptk_export_sorted(ctx, d);
    if (!rc) rc = d2h(ctx, out, d, (size_t)ctx->n * sizeof(pt_PrimitiveRayCS));
    (void)hipFree(d);
    return rc;
}

int pt_download_tree(pt_context *ctx, pt_TreeNode *out)
{
    PT_NEED_BUILT(ctx, "pt_download_tree");
    if (!out) return PT_ERR_INVALID;
    if (int rc = set_device(ctx)) return rc;
    if (!ctx->tree_exported) {
        if (!ctx->d_tree) { if (int rc = dev_alloc(ctx, &ctx->d_tree, 2 * (size_t)ctx->capacity + 1)) return rc; }
        if (int rc = ptk_export_tree(ctx)) return rc;
        ctx->tree_exported = true;
    }
    return d2h(ctx, out, ctx->d_tree, 2 * (size_t)ctx->n * sizeof(pt_TreeNode));
}

// ---------------------------------------------------------------------------
int pt_resize(pt_context *ctx, int width, int height)
{
    if (!ctx) return PT_ERR_INVALID;
    if (width < 1 || height < 1 || (long long)width * height > (1ll << 28)) PT_FAIL(ctx, PT_ERR_INVALID, "pt_resize: bad size");
    if (int rc = set_device(ctx)) return rc;
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    free_images(ctx);
    const size_t npix = (size_t)width * height;
    int rc = dev_alloc(ctx, &ctx->d_diffuse, npix * 4);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_normal, npix);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_world, npix);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_output, npix);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_display, npix * 4);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_linear, npix);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_cost, npix);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_pixel_work, npix);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_okeys[0], npix);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_okeys[1], npix);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_ovals[0], npix);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_ovals[1], npix);
    if (!rc) rc = dev_alloc(ctx, &ctx->d_ohist, 256 * ((npix + PT_SORT_TILE - 1) / PT_SORT_TILE));
    if (!rc) rc = dev_alloc(ctx, &ctx->d_ototal, 512);
    if (rc) { free_images(ctx); return rc; }
    ctx->W = width;
    ctx->H = height;
    PT_HIP(ctx, hipMemsetAsync(ctx->d_diffuse, 0, npix * 4, ctx->stream));
    PT_HIP(ctx, hipMemsetAsync(ctx->d_normal, 0, npix * 16, ctx->stream));
    PT_HIP(ctx, hipMemsetAsync(ctx->d_world, 0, npix * 16, ctx->stream));
    PT_HIP(ctx, hipMemsetAsync(ctx->d_cost, 0, npix * 4, ctx->stream));
    PT_HIP(ctx, hipMemsetAsync(ctx->d_pixel_work, 0, npix * 4, ctx->stream));
    return pt_reset_accumulation(ctx);
}

int pt_reset_accumulation(pt_context *ctx)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->W) PT_FAIL(ctx, PT_ERR_INVALID, "pt_reset_accumulation: call pt_resize first");
    if (int rc = set_device(ctx)) return rc;
    const size_t npix = (size_t)ctx->W * ctx->H;
    PT_HIP(ctx, hipMemsetAsync(ctx->d_output, 0, npix * 16, ctx->stream));
    PT_HIP(ctx, hipMemsetAsync(ctx->d_display, 0, npix * 4, ctx->stream));
    PT_HIP(ctx, hipMemsetAsync(ctx->d_linear, 0, npix * 16, ctx->stream));
    ctx->frame_count = 0;
    return PT_OK;
}

int pt_upload_gbuffer(pt_context *ctx, const uint8_t *diffuse, const float *normal, const float *world)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->W) PT_FAIL(ctx, PT_ERR_INVALID, "pt_upload_gbuffer: call pt_resize first");
    if (!diffuse || !normal || !world) return PT_ERR_INVALID;
    if (int rc = set_device(ctx)) return rc;
    const size_t npix = (size_t)ctx->W * ctx->H;
    PT_HIP(ctx, hipMemcpyAsync(ctx->d_diffuse, diffuse, npix * 4, hipMemcpyHostToDevice, ctx->stream));
    PT_HIP(ctx, hipMemcpyAsync(ctx->d_normal, normal, npix * 16, hipMemcpyHostToDevice, ctx->stream));
    PT_HIP(ctx, hipMemcpyAsync(ctx->d_world, world, npix * 16, hipMemcpyHostToDevice, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_gbuffer = true;
    ctx->order_valid = false;
    PT_HIP(ctx, hipMemsetAsync(ctx->d_cost, 0, npix * 4, ctx->stream));
    return PT_OK;
}

int pt_download_gbuffer(pt_context *ctx, uint8_t *diffuse, float *normal, float *world)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->have_gbuffer) PT_FAIL(ctx, PT_ERR_INVALID, "pt_download_gbuffer: no G-buffer yet");
    if (int rc = set_device(ctx)) return rc;
    const size_t npix = (size_t)ctx->W * ctx->H;
    if (diffuse) PT_HIP(ctx, hipMemcpyAsync(diffuse, ctx->d_diffuse, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (normal) PT_HIP(ctx, hipMemcpyAsync(normal, ctx->d_normal, npix * 16, hipMemcpyDeviceToHost, ctx->stream));
    if (world) PT_HIP(ctx, hipMemcpyAsync(world, ctx->d_world, npix * 16, hipMemcpyDeviceToHost, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

// Bytes the colour render target holds for a material colour.  The reference
// renders Kd into an sRGB back buffer and copies the raw bytes into a UNORM
// texture (CS.cpp:302,357-359,373), so its shader samples the sRGB-ENCODED
// value; srgb_quirk = 0 stores plain UNORM8.
void pt_albedo_bytes(const float kd[3], int srgb_quirk, uint8_t out_rgb[3])
{
    for (int k = 0; k < 3; k++) {
        double c = kd[k];
        if (!(c == c) || c < 0.0) c = 0.0;
        if (c > 1.0) c = 1.0;
        if (srgb_quirk) c = (c <= 0.0031308) ? 12.92 * c : 1.055 * std::pow(c, 1.0 / 2.4) - 0.055;
        out_rgb[k] = (uint8_t)(int)(c * 255.0 + 0.5);
    }
}

int pt_primary_hits(pt_context *ctx, const pt_camera *cam, const uint8_t *albedo_rgba8)
{
    PT_NEED_BUILT(ctx, "pt_primary_hits");
    if (!ctx->trace_records) PT_FAIL(ctx, PT_ERR_INVALID, "pt_primary_hits: the last build was PT_BUILD_TREE_ONLY");
    if (!cam || !albedo_rgba8) return PT_ERR_INVALID;
    if (!ctx->W) PT_FAIL(ctx, PT_ERR_INVALID, "pt_primary_hits: call pt_resize first");
    uint8_t *d_alb = nullptr;
    if (int rc = dev_alloc(ctx, &d_alb, 64)) return rc;
    hipError_t e = hipMemcpyAsync(d_alb, albedo_rgba8, 64, hipMemcpyHostToDevice, ctx->stream);
    int rc = PT_OK;
    if (e != hipSuccess) { ctx->err = std::string("albedo upload: ") + hipGetErrorString(e); rc = PT_ERR_HIP; }
    if (!rc) rc = ptk_primary(ctx, cam, d_alb);
    if (!rc) { e = hipStreamSynchronize(ctx->stream); if (e != hipSuccess) { ctx->err = hipGetErrorString(e); rc = PT_ERR_HIP; } }
    (void)hipFree(d_alb);
    if (!rc) {
        ctx->have_gbuffer = true;
        ctx->order_valid = false;
        PT_HIP(ctx, hipMemsetAsync(ctx->d_cost, 0, (size_t)ctx->W * ctx->H * 4, ctx->stream));
    }
    return rc;
}

// ---------------------------------------------------------------------------
// MT19937 raw outputs (what std::mt19937 specifies)
void pt_frame_seeds(uint32_t base_seed, uint32_t first_frame, uint32_t count, uint32_t *out)
{
    uint32_t mt[624];
    mt[0] = base_seed;
    for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    int idx = 624;
    auto next = [&]() -> uint32_t {
        if (idx >= 624) {
            for (int i = 0; i < 624; i++) {
                uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
                mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    };
    for (uint64_t i = 0, skip = (uint64_t)first_frame * 16u; i < skip; i++) (void)next();
    for (uint32_t f = 0; f < count; f++) {
        out[f] = next();
        for (int k = 1; k < 16; k++) (void)next();
    }
}

static int check_trace_ready(pt_context *ctx, const char *who)
{
    if (!ctx->built) PT_FAIL(ctx, PT_ERR_INVALID, std::string(who) + ": call pt_build first");
    if (!ctx->trace_records) PT_FAIL(ctx, PT_ERR_INVALID, std::string(who) + ": the last build was PT_BUILD_TREE_ONLY");
    if (!ctx->have_cb) PT_FAIL(ctx, PT_ERR_INVALID, std::string(who) + ": call pt_set_constants first");
    if (ctx->cb.node_count != ctx->n) PT_FAIL(ctx, PT_ERR_INVALID, std::string(who) + ": CB_Radix.node_count differs from the primitive count");
    if (!ctx->W) PT_FAIL(ctx, PT_ERR_INVALID, std::string(who) + ": call pt_resize first");
    if (!ctx->have_gbuffer) PT_FAIL(ctx, PT_ERR_INVALID, std::string(who) + ": no G-buffer (pt_upload_gbuffer / pt_primary_hits)");
    // The reference's 15-entry stack silently drops subtrees of trees deeper than 13 levels; this
    // library never drops.  stack_entries 15 is the strict-parity guard (deep trees are refused); 0, the
    // default, and 32 trace any tree with the non-dropping traversal.
    if (ctx->stack_entries == 15 && (ctx->build_flags & PT_BUILD_TOPOLOGY_MORTON))
        PT_FAIL(ctx, PT_ERR_UNSUPPORTED, std::string(who) + ": the strict-parity stack (15) does not go with PT_BUILD_TOPOLOGY_MORTON");
    if (ctx->stack_entries == 15 && bit_length_u((unsigned)ctx->n) > 14)
        PT_FAIL(ctx, PT_ERR_UNSUPPORTED, std::string(who) + ": tree deeper than the 15-entry reference stack; pt_set_stack_size(ctx, 32)");
    return PT_OK;
}

static int trace_frames(pt_context *ctx, const char *who, uint32_t first_frame, uint32_t count, const uint32_t *seeds, int mode,
                        const int *region)
{
    if (int rc = check_trace_ready(ctx, who)) return rc;
    if (!seeds || count < 1) PT_FAIL(ctx, PT_ERR_INVALID, std::string(who) + ": need count >= 1 seeds");
    if (mode != PT_ACCUM_LITERAL && mode != PT_ACCUM_LINEAR) PT_FAIL(ctx, PT_ERR_INVALID, std::string(who) + ": bad mode");
    // Short whole-image calls are overlapped (see PipeSlot): at most 8 frames, persistent schedule, a tree in LDS or read as
    // plain records (no child lines or deep stack entries, which a launch may have to create)
    const bool overlapped = ctx->overlap && !ctx->user_stream && !region && count <= 8 && ctx->W * (size_t)ctx->H * count <= ((size_t)1 << 26) &&
                            ctx->schedule == PT_SCHEDULE_PERSISTENT && ptk_short_calls_may_overlap(ctx);
    if (!overlapped) { if (int rc = set_device(ctx)) return rc; }
    else PT_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->h_seeds_cap < count) {
        // (the staging buffers are shared by the context's stream and the internal ones)
        PT_HIP(ctx, hipDeviceSynchronize());
        for (int k = 0; k < PT_SEED_SLOTS; k++) { if (ctx->h_seeds[k]) (void)hipHostFree(ctx->h_seeds[k]); ctx->h_seeds[k] = nullptr; }
        ctx->h_seeds_cap = 0;
        for (int k = 0; k < PT_SEED_SLOTS; k++) {
            PT_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_seeds[k]), (size_t)count * sizeof(uint32_t), hipHostMallocDefault));
            if (!ctx->seed_ev[k]) PT_HIP(ctx, hipEventCreateWithFlags(&ctx->seed_ev[k], hipEventDisableTiming));
            ctx->seed_ev_used[k] = false;
        }
        ctx->h_seeds_cap = count;
    }
    // an overlapped call works with its slot's stream and buffers in the context's place
    struct Swap {
        pt_context *c; pt_context::PipeSlot *sl;
        void flip() {
            std::swap(c->stream, sl->stream); std::swap(c->d_samples, sl->d_samples); std::swap(c->samples_frames, sl->samples_frames);
            std::swap(c->d_work, sl->d_work); std::swap(c->d_seeds, sl->d_seeds); std::swap(c->seeds_cap, sl->seeds_cap);
        }
        Swap(pt_context *c_, pt_context::PipeSlot *s_) : c(c_), sl(s_) { if (sl) flip(); }
        ~Swap() { if (sl) { flip(); c->pipe_active = -1; } }
    };
    pt_context::PipeSlot *sl = nullptr;
    if (overlapped) {
        const int i = ctx->pipe_next;
        sl = &ctx->pipe[i];
        // (each piece on its own: a call that failed half-way leaves a slot the next call completes)
        if (!sl->stream) {
            PT_HIP(ctx, hipStreamCreateWithFlags(&sl->stream, hipStreamNonBlocking));
            ctx->main_dirty = true;           // a new stream has not waited for anything the context's stream holds
        }
        if (!sl->done) PT_HIP(ctx, hipEventCreateWithFlags(&sl->done, hipEventDisableTiming));
        if (!sl->d_work) { if (int rc = dev_alloc(ctx, &sl->d_work, 64)) return rc; }
        if (!ctx->pipe_fork) PT_HIP(ctx, hipEventCreateWithFlags(&ctx->pipe_fork, hipEventDisableTiming));
        if (ctx->main_dirty) {
            // whatever the context's stream holds (a build, constants, the G-buffer, a reset of the images) comes first,
            // for both slots
            PT_HIP(ctx, hipEventRecord(ctx->pipe_fork, ctx->stream));
            for (auto &any : ctx->pipe) if (any.stream) PT_HIP(ctx, hipStreamWaitEvent(any.stream, ctx->pipe_fork, 0));
            ctx->main_dirty = false;
        }
        ctx->pipe_next = (i + 1) % ctx->knobs.pipe_slots;
        ctx->pipe_active = i;
    }
    Swap swap(ctx, sl);
    // a long call on the context's own stream: its path seeds are computed on a stream of the library's while the previous
    // call's trace kernel runs (pt_context::SeedsAhead)
    if (!overlapped && ctx->overlap && !ctx->user_stream && !region && ptk_seeds_ahead_ok(ctx, count)) {
        pt_context::SeedsAhead &a = ctx->ahead;
        const int b = a.next;
        a.next ^= 1;
        if (!a.stream) PT_HIP(ctx, hipStreamCreateWithFlags(&a.stream, hipStreamNonBlocking));
        if (!a.ready[b]) PT_HIP(ctx, hipEventCreateWithFlags(&a.ready[b], hipEventDisableTiming));
        if (!a.released[b]) PT_HIP(ctx, hipEventCreateWithFlags(&a.released[b], hipEventDisableTiming));
        const size_t paths = (size_t)ctx->W * ctx->H * count;
        if (a.frame_cap[b] < count || a.path_cap[b] < paths) {
            // (the set's last reader, a trace kernel on the context's stream, has to be done before the buffers go)
            PT_HIP(ctx, hipStreamSynchronize(a.stream));
            PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            dev_free(a.d_frame_seeds[b]); dev_free(a.d_path_seeds[b]);
            a.frame_cap[b] = 0; a.path_cap[b] = 0; a.released_used[b] = false;
            if (int rc = dev_alloc(ctx, &a.d_frame_seeds[b], (size_t)count)) return rc;
            if (int rc = dev_alloc(ctx, &a.d_path_seeds[b], paths)) return rc;
            a.frame_cap[b] = count; a.path_cap[b] = paths;
        }
        if (a.released_used[b]) PT_HIP(ctx, hipStreamWaitEvent(a.stream, a.released[b], 0));
        const int slot = ctx->seed_slot;
        ctx->seed_slot = (slot + 1) % PT_SEED_SLOTS;
        if (ctx->seed_ev_used[slot]) PT_HIP(ctx, hipEventSynchronize(ctx->seed_ev[slot]));
        std::memcpy(ctx->h_seeds[slot], seeds, (size_t)count * sizeof(uint32_t));
        PT_HIP(ctx, hipMemcpyAsync(a.d_frame_seeds[b], ctx->h_seeds[slot], (size_t)count * sizeof(uint32_t), hipMemcpyHostToDevice, a.stream));
        PT_HIP(ctx, hipEventRecord(ctx->seed_ev[slot], a.stream));
        ctx->seed_ev_used[slot] = true;
        if (int rc = ptk_path_seeds_ahead(ctx, a.stream, count, a.d_frame_seeds[b], a.d_path_seeds[b])) return rc;
        PT_HIP(ctx, hipEventRecord(a.ready[b], a.stream));
        PT_HIP(ctx, hipStreamWaitEvent(ctx->stream, a.ready[b], 0));
        ctx->path_seeds = a.d_path_seeds[b];
        const int rc = ptk_trace_frames(ctx, first_frame, count, mode, region);
        ctx->path_seeds = nullptr;
        if (rc) return rc;
        PT_HIP(ctx, hipEventRecord(a.released[b], ctx->stream));
        a.released_used[b] = true;
        ctx->frame_count = first_frame + count;
        return PT_OK;
    }
    if (ctx->seeds_cap < count) {
        PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        dev_free(ctx->d_seeds);
        ctx->seeds_cap = 0;
        if (int rc = dev_alloc(ctx, &ctx->d_seeds, (size_t)count)) return rc;
        ctx->seeds_cap = count;
    }
    // pinned staging slots in rotation: the call does not wait for the previous launch, only for the
    // copy that last used this slot (PT_SEED_SLOTS calls ago)
    const int slot = ctx->seed_slot;
    ctx->seed_slot = (slot + 1) % PT_SEED_SLOTS;
    if (ctx->seed_ev_used[slot]) PT_HIP(ctx, hipEventSynchronize(ctx->seed_ev[slot]));
    std::memcpy(ctx->h_seeds[slot], seeds, (size_t)count * sizeof(uint32_t));
    PT_HIP(ctx, hipMemcpyAsync(ctx->d_seeds, ctx->h_seeds[slot], (size_t)count * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    PT_HIP(ctx, hipEventRecord(ctx->seed_ev[slot], ctx->stream));
    ctx->seed_ev_used[slot] = true;
    if (int rc = ptk_trace_frames(ctx, first_frame, count, mode, region)) return rc;
    if (sl) {
        // ctx->stream is the slot's stream here, sl->stream the context's own: the latter waits for this call at once, so
        // whatever is enqueued on it later (downloads, the caller's own reads of the images) sees the finished frame
        PT_HIP(ctx, hipEventRecord(sl->done, ctx->stream));
        sl->used = true;
        PT_HIP(ctx, hipStreamWaitEvent(sl->stream, sl->done, 0));
    }
    ctx->frame_count = first_frame + count;
    return PT_OK;
}

int pt_trace_frames(pt_context *ctx, uint32_t first_frame, uint32_t count, const uint32_t *seeds, int mode)
{
    if (!ctx) return PT_ERR_INVALID;
    return trace_frames(ctx, "pt_trace_frames", first_frame, count, seeds, mode, nullptr);
}

int pt_trace_region(pt_context *ctx, uint32_t framecount, uint32_t seed, int x0, int y0, int x1, int y1)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->W) PT_FAIL(ctx, PT_ERR_INVALID, "pt_trace_region: call pt_resize first");
    if (x0 < 0 || y0 < 0) PT_FAIL(ctx, PT_ERR_INVALID, "pt_trace_region: negative offset");
    // the shader's guard keeps x <= x1 and y <= y1 (ray_tracing.hlsl:405); texels outside the image do not exist
    int region[4] = {x0, y0, x1 < ctx->W - 1 ? x1 : ctx->W - 1, y1 < ctx->H - 1 ? y1 : ctx->H - 1};
    if (region[2] < region[0] || region[3] < region[1]) {      // nothing to trace: the dispatch would be empty
        ctx->frame_count = framecount + 1;
        return PT_OK;
    }
    return trace_frames(ctx, "pt_trace_region", framecount, 1, &seed, PT_ACCUM_LITERAL, region);
}

int pt_trace(pt_context *ctx, uint32_t framecount, uint32_t seed)
{
    return pt_trace_frames(ctx, framecount, 1, &seed, PT_ACCUM_LITERAL);
}

int pt_get_frame_count(pt_context *ctx, uint32_t *frames)
{
    if (!ctx || !frames) return PT_ERR_INVALID;
    *frames = ctx->frame_count;
    return PT_OK;
}

int pt_linear_sum_device_ptr(pt_context *ctx, void **dev_ptr, size_t *bytes)
{
    if (!ctx || !dev_ptr) return PT_ERR_INVALID;
    if (!ctx->W) PT_FAIL(ctx, PT_ERR_INVALID, "pt_linear_sum_device_ptr: call pt_resize first");
    *dev_ptr = ctx->d_linear;
    if (bytes) *bytes = (size_t)ctx->W * ctx->H * 16;
    return PT_OK;
}

int pt_resolve(pt_context *ctx)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->W) PT_FAIL(ctx, PT_ERR_INVALID, "pt_resolve: call pt_resize first");
    if (int rc = set_device(ctx)) return rc;
    return ptk_resolve(ctx);
}

int pt_download_output(pt_context *ctx, float *rgba32f)
{
    if (!ctx || !rgba32f) return PT_ERR_INVALID;
    if (!ctx->W) PT_FAIL(ctx, PT_ERR_INVALID, "pt_download_output: call pt_resize first");
    if (int rc = set_device(ctx)) return rc;
    return d2h(ctx, rgba32f, ctx->d_output, (size_t)ctx->W * ctx->H * 16);
}

int pt_download_display(pt_context *ctx, uint8_t *rgba8)
{
    if (!ctx || !rgba8) return PT_ERR_INVALID;
    if (!ctx->W) PT_FAIL(ctx, PT_ERR_INVALID, "pt_download_display: call pt_resize first");
    if (int rc = set_device(ctx)) return rc;
    return d2h(ctx, rgba8, ctx->d_display, (size_t)ctx->W * ctx->H * 4);
}

int pt_output_device_ptr(pt_context *ctx, void **dev_rgba32f, void **dev_rgba8)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->W) PT_FAIL(ctx, PT_ERR_INVALID, "pt_output_device_ptr: call pt_resize first");
    if (dev_rgba32f) *dev_rgba32f = ctx->d_output;
    if (dev_rgba8) *dev_rgba8 = ctx->d_display;
    return PT_OK;
}

// ---------------------------------------------------------------------------
int pt_set_stack_size(pt_context *ctx, int entries)
{
    if (!ctx) return PT_ERR_INVALID;
    if (entries != 0 && entries != 15 && entries != 32) PT_FAIL(ctx, PT_ERR_INVALID, "pt_set_stack_size: 0 (automatic), 15 or 32");
    ctx->stack_entries = entries;
    return PT_OK;
}

int pt_set_max_segments(pt_context *ctx, int segments)
{
    if (!ctx) return PT_ERR_INVALID;
    if (segments < 0 || segments > 14) PT_FAIL(ctx, PT_ERR_INVALID, "pt_set_max_segments: 0 (reference: 14) or 1..14");
    ctx->itr_limit = segments == 0 ? 15 : segments + 1;
    return PT_OK;
}

int pt_set_trace_overlap(pt_context *ctx, int enable)
{
    if (!ctx) return PT_ERR_INVALID;
    if (int rc = set_device(ctx)) return rc;
    ctx->overlap = enable ? 1 : 0;
    return PT_OK;
}

int pt_set_traversal_order(pt_context *ctx, int order)
{
    if (!ctx) return PT_ERR_INVALID;
    if (order != PT_TRAVERSAL_REFERENCE && order != PT_TRAVERSAL_NEAR_FIRST)
        PT_FAIL(ctx, PT_ERR_INVALID, "pt_set_traversal_order: PT_TRAVERSAL_REFERENCE or PT_TRAVERSAL_NEAR_FIRST");
    ctx->near_first = order;
    return PT_OK;
}

int pt_set_trace_schedule(pt_context *ctx, int schedule)
{
    if (!ctx) return PT_ERR_INVALID;
    if (schedule != PT_SCHEDULE_PERSISTENT && schedule != PT_SCHEDULE_WAVEFRONT)
        PT_FAIL(ctx, PT_ERR_INVALID, "pt_set_trace_schedule: PT_SCHEDULE_PERSISTENT or PT_SCHEDULE_WAVEFRONT");
    ctx->schedule = schedule;
    return PT_OK;
}

int pt_set_trace_timing(pt_context *ctx, int enabled)
{
    if (!ctx) return PT_ERR_INVALID;
    if (int rc = set_device(ctx)) return rc;
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drop_trace_events(ctx);
    ctx->time_trace = enabled != 0;
    return PT_OK;
}

int pt_get_trace_times(pt_context *ctx, float *ms, int capacity, int *count)
{
    if (!ctx || !count || (capacity > 0 && !ms)) return PT_ERR_INVALID;
    if (int rc = set_device(ctx)) return rc;
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *count = (int)ctx->trace_events.size();
    for (int i = 0; i < *count && i < capacity; i++)
        PT_HIP(ctx, hipEventElapsedTime(&ms[i], ctx->trace_events[i].first, ctx->trace_events[i].second));
    drop_trace_events(ctx);
    return PT_OK;
}

int pt_set_detailed_counters(pt_context *ctx, int enabled)
{
    if (!ctx) return PT_ERR_INVALID;
    ctx->detailed = enabled != 0;
    return PT_OK;
}

int pt_get_counters(pt_context *ctx, pt_counters *out)
{
    if (!ctx || !out) return PT_ERR_INVALID;
    if (int rc = set_device(ctx)) return rc;
    DevCounters h;
    if (int rc = d2h(ctx, &h, ctx->d_counters, sizeof(h))) return rc;
    out->paths = h.paths; out->rays = h.rays; out->node_visits = h.node_visits;
    out->box_tests = h.box_tests; out->tri_tests = h.tri_tests;
    if (ctx->knobs.print_sched)
        fprintf(stderr, "sched: iters %llu int_lanes %llu leaf_rounds %llu leaf_lanes %llu at_leaf_wait %llu refills %llu idle_wait %llu no_pixel %llu refill_ticks %llu trav_ticks %llu\n",
                h.sched[0], h.sched[1], h.sched[2], h.sched[3], h.sched[4], h.sched[5], h.sched[6], h.sched[7], h.sched[8], h.sched[9]);
    return PT_OK;
}

int pt_reset_counters(pt_context *ctx)
{
    if (!ctx) return PT_ERR_INVALID;
    if (int rc = set_device(ctx)) return rc;
    PT_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, sizeof(DevCounters), ctx->stream));
    return PT_OK;
}

// ---------------------------------------------------------------------------
// Multi-GPU merge of the linear sums: RCCL, bound at first use (a single-GPU process never loads it).
// Only the five entry points below are needed; their prototypes are the public ones of rccl.h.
namespace {
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(void *id) = nullptr;                                   // ncclGetUniqueId(ncclUniqueId *)
    int (*CommInitRank)(void **comm, int nranks, pt_comm_id id, int rank) = nullptr;   // ncclUniqueId is 128 bytes, by value
    int (*CommDestroy)(void *comm) = nullptr;
    int (*CommCount)(void *comm, int *count) = nullptr;
    int (*AllReduce)(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t st) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
};
RcclApi g_rccl;

bool rccl_load()
{
    if (g_rccl.AllReduce) return true;
    // the soname first: a process that already holds an RCCL (torch.distributed) gets that same copy
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *nm : names) { h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) { g_rccl.err = std::string("cannot load librccl: ") + dlerror(); return false; }
    g_rccl.lib = h;
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(dlsym(h, "ncclCommCount"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    auto ar = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(h, "ncclAllReduce"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.CommCount || !g_rccl.GetErrorString || !ar) {
        g_rccl.err = "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclCommCount / ncclAllReduce";
        return false;
    }
    g_rccl.AllReduce = ar;
    return true;
}
constexpr int kNcclFloat32 = 7, kNcclSum = 0;       // ncclFloat32, ncclSum of rccl.h
}  // namespace

int pt_comm_unique_id(pt_comm_id *out)
{
    if (!out) return PT_ERR_INVALID;
    if (!rccl_load()) { g_create_error = g_rccl.err; return PT_ERR_UNSUPPORTED; }
    const int r = g_rccl.GetUniqueId(out);
    if (r != 0) { g_create_error = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r); return PT_ERR_HIP; }
    return PT_OK;
}

int pt_comm_init_rank(pt_context *ctx, int n_ranks, int rank, const pt_comm_id *id)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks) PT_FAIL(ctx, PT_ERR_INVALID, "pt_comm_init_rank: bad rank / size / id");
    if (ctx->rccl_comm) PT_FAIL(ctx, PT_ERR_INVALID, "pt_comm_init_rank: the context already has a communicator");
    if (!rccl_load()) PT_FAIL(ctx, PT_ERR_UNSUPPORTED, g_rccl.err);
    if (int rc = set_device(ctx)) return rc;
    void *comm = nullptr;
    const int r = g_rccl.CommInitRank(&comm, n_ranks, *id, rank);
    if (r != 0) PT_FAIL(ctx, PT_ERR_HIP, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
    ctx->rccl_comm = comm;
    return PT_OK;
}

int pt_comm_destroy(pt_context *ctx)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->rccl_comm) return PT_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    const int r = g_rccl.CommDestroy(ctx->rccl_comm);
    ctx->rccl_comm = nullptr;
    if (r != 0) PT_FAIL(ctx, PT_ERR_HIP, std::string("ncclCommDestroy: ") + g_rccl.GetErrorString(r));
    return PT_OK;
}

int pt_comm_count(pt_context *ctx, int *n_ranks)
{
    if (!ctx || !n_ranks) return PT_ERR_INVALID;
    *n_ranks = 0;
    if (!ctx->rccl_comm) return PT_OK;
    const int r = g_rccl.CommCount(ctx->rccl_comm, n_ranks);
    if (r != 0) PT_FAIL(ctx, PT_ERR_HIP, std::string("ncclCommCount: ") + g_rccl.GetErrorString(r));
    return PT_OK;
}

int pt_reduce_linear_sum(pt_context *ctx, void *nccl_comm)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->W) PT_FAIL(ctx, PT_ERR_INVALID, "pt_reduce_linear_sum: call pt_resize first");
    void *comm = nccl_comm ? nccl_comm : ctx->rccl_comm;
    if (!comm) PT_FAIL(ctx, PT_ERR_INVALID, "pt_reduce_linear_sum: no communicator (pass one, or call pt_comm_init_rank)");
    if (!rccl_load()) PT_FAIL(ctx, PT_ERR_UNSUPPORTED, g_rccl.err);
    if (int rc = set_device(ctx)) return rc;
    // one in-place all-reduce of W*H float RGBA sums (alpha = sample count) on the context's stream
    const int r = g_rccl.AllReduce(ctx->d_linear, ctx->d_linear, (size_t)ctx->W * ctx->H * 4, kNcclFloat32, kNcclSum, comm, ctx->stream);
    if (r != 0) PT_FAIL(ctx, PT_ERR_HIP, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r));
    return PT_OK;
}

int pt_intersect(pt_context *ctx, const float *rays8, int n_rays, pt_hit *out)
{
    PT_NEED_BUILT(ctx, "pt_intersect");
    if (!ctx->trace_records) PT_FAIL(ctx, PT_ERR_INVALID, "pt_intersect: the last build was PT_BUILD_TREE_ONLY");
    if (!rays8 || !out || n_rays < 1) return PT_ERR_INVALID;
    float *d_rays = nullptr;
    pt_hit *d_out = nullptr;
    int rc = dev_alloc(ctx, &d_rays, (size_t)n_rays * 8);
    if (!rc) rc = dev_alloc(ctx, &d_out, (size_t)n_rays);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(d_rays, rays8, (size_t)n_rays * 32, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) { ctx->err = hipGetErrorString(e); rc = PT_ERR_HIP; }
    }
    if (!rc) rc = ptk_intersect(ctx, d_rays, n_rays, d_out);
    if (!rc) rc = d2h(ctx, out, d_out, (size_t)n_rays * sizeof(pt_hit));
    if (d_rays) (void)hipFree(d_rays);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

}  // extern "C"
