/*
 * libptamd -- C ABI of the MI355X (gfx950) hot path of the path tracer:
 * acceleration-structure build (bounds, scene box, Morton codes, radix sort,
 * binary-radix-tree topology, bottom-up refit) and the path-trace kernel
 * (BVH traversal, triangle tests, BSDF sampling, progressive accumulation).
 *
 * It replaces the D3D11 seam of the reference -- the ShaderManager /
 * BufferManager base classes of `RayTracer` (reference: ray_tracer.hpp:7-78,
 * buffer_manager.hpp:21-31, shader_manager.hpp:10-12) and the Dispatch /
 * CopyResource / Map calls in ray_tracer.cpp:263-335 -- with a headless HIP
 * layer.  Plain pointers and sizes only; nothing throws across the boundary.
 *
 * Conventions
 *  - every call returns PT_OK (0) or a negative pt_status; pt_last_error()
 *    returns the message of the most recent failure on that context;
 *  - a context belongs to one GPU and is used from one thread at a time (the
 *    reference is single-threaded on the immediate context);
 *  - work is enqueued on the context's stream (pt_set_stream) and is
 *    asynchronous unless the call downloads to host memory;
 *  - all layouts are those of include/ptamd_structs.h.
 */
#ifndef PTAMD_H
#define PTAMD_H

#include <stddef.h>
#include <stdint.h>
#include "ptamd_structs.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pt_context pt_context;

typedef enum pt_status {
    PT_OK = 0,
    PT_ERR_INVALID = -1,      /* bad argument / call order            */
    PT_ERR_HIP = -2,          /* a HIP runtime call failed            */
    PT_ERR_NO_DEVICE = -3,    /* no usable gfx950 device              */
    PT_ERR_UNSUPPORTED = -4   /* input outside what the path supports */
} pt_status;

/* ---- context ------------------------------------------------------------
 * replaces: DXUTGetD3D11Device()/immediate context acquisition in
 * buffer_manager.cpp:259 ff. and RayTracer::release (ray_tracer.cpp:351-355). */
int         pt_create(int device_ordinal, pt_context **out);
void        pt_destroy(pt_context *ctx);
const char *pt_last_error(const pt_context *ctx);     /* ctx may be NULL: last create error */
/* hipStream_t as void*; NULL = the default stream.  Work already enqueued on the previous stream is
 * drained (the call blocks for it) before the switch, so nothing of this context is ever in flight on
 * two streams; pending pt_get_trace_times entries are dropped. */
int         pt_set_stream(pt_context *ctx, void *hip_stream);
int         pt_synchronize(pt_context *ctx);
const char *pt_version(void);

/* ---- scene upload ---------------------------------------------------------
 * replaces: addSRV(sizeof(PrimitiveRender), n, renderer_primitives.data())
 * (ray_tracer.cpp:144-147).  Host or device memory, n >= 1.
 * Material ids: the constant buffer holds 16 materials (structs.hlsli).  pt_upload_primitives checks
 * every `matid` and fails with PT_ERR_UNSUPPORTED outside 0..15.  Device-side inputs cannot be checked
 * up front (pt_upload_primitives_device, the G-buffer's asint(normal.w) - 1): there an id outside
 * 0..15 reads an all-zero material, as an out-of-range constant-buffer read does in Direct3D 11, and
 * never memory beyond the table; a non-positive G-buffer material word is background.
 * pt_upload_primitives_device keeps the caller's pointer (no copy): the records must stay valid and
 * unchanged until the next upload or pt_destroy -- pt_primary_hits and the build-product downloads read
 * them again after pt_build. */
int pt_upload_primitives(pt_context *ctx, const pt_PrimitiveRender *host_prims, int n);
int pt_upload_primitives_device(pt_context *ctx, const void *dev_prims, int n);
/* replaces: the per-frame Map/write of constant_buffers_[0] (CS.cpp:405-439).
 * node_count must equal the primitive count. */
int pt_set_constants(pt_context *ctx, const pt_CB_Radix *cb);

/* ---- build ----------------------------------------------------------------
 * replaces: the CPU half of RayTracer::loadScene (AABBs, scene box, Morton
 * codes, sort, strip: ray_tracer.cpp:41-66), RayTracer::buildRadixTree
 * (ray_tracer.cpp:263-269 -> radix_build_cs.hlsl) and RayTracer::buildBVH
 * (ray_tracer.cpp:271-279 -> bvh_build_cs.hlsl).
 * Equal Morton codes keep ascending primitive index (stable sort).  An axis on which the scene box
 * is flat (0/0 in structs.hpp:92-97) quantises to cell 0, which is what the reference's x86 float ->
 * integer conversion leaves in the bits it interleaves. */
#define PT_BUILD_DEFAULT        0u
#define PT_BUILD_MORTON_FULL    1u   /* NON-PARITY: quantise to 2^21 cells per axis instead of the reference's 22.
                                        pt_build synchronises the stream with this flag (it reads back whether every
                                        workgroup of the eight-pass sort found the tiles before it; PT_ERR_HIP if not) */
#define PT_BUILD_KARRAS_LINKS   4u   /* pt_download_tree also runs the search-based link kernel (the direct
                                        counterpart of radix_build_cs.hlsl) over the links it has written from
                                        the closed form of the index-keyed tree -- same values */
#define PT_BUILD_TOPOLOGY_MORTON 16u /* NON-PARITY: the binary radix tree over the sorted Morton codes themselves (equal
                                        codes told apart by their position; Karras 2012) instead of the reference's
                                        tree over the leaf INDEX (radix_build_cs.hlsl:18-25 takes clz(i ^ j) of the
                                        indices).  Splits then follow the cells of the code grid: on triangle soups
                                        about half the node visits per ray.  n >= 2; n - 1 internal nodes, no
                                        out-of-range leaf; trees higher than 64 levels are refused
                                        (PT_ERR_UNSUPPORTED).  Different tree, same nearest hits: images differ from
                                        the reference's only where two triangles are hit at equal distance.
                                        Persistent schedule; pt_set_stack_size 0 or 32.  pt_build synchronises the
                                        stream with this flag (it reads the height of the finished tree back) */
#define PT_BUILD_TREE_ONLY      8u   /* the tree and the sorted order only: the per-leaf triangle records of the
                                        trace kernels are not written (the refit then gathers the 32-byte boxes
                                        instead of the 80-byte primitives); pt_primary_hits, pt_intersect and
                                        pt_trace* refuse until a build without this flag */
int pt_build(pt_context *ctx, unsigned flags);

typedef enum pt_build_stage {
    PT_STAGE_BOUNDS = 0,   /* per-triangle AABB + scene box   */
    PT_STAGE_MORTON,       /* Morton codes                    */
    PT_STAGE_SORT,         /* radix sort (all passes)         */
    PT_STAGE_TOPOLOGY,     /* binary radix tree links         */
    PT_STAGE_REFIT,        /* leaf init + bottom-up AABBs; unless PT_BUILD_TREE_ONLY also the per-leaf
                              triangle records of the trace kernels (same fetch of the primitive) */
    PT_STAGE_PACK,         /* the root's own box for the trace kernels */
    PT_STAGE_COUNT
} pt_build_stage;
/* device-side durations (ms, HIP events) of the last pt_build; total = sum.
 * Synchronises the stream. */
int pt_build_times(pt_context *ctx, float stage_ms[PT_STAGE_COUNT], float *total_ms, int *sort_passes);

/* downloads of build products in the reference layouts (each synchronises) */
int pt_download_scene_bound(pt_context *ctx, pt_Bound *bb);             /* RayTracer::bb_               */
int pt_download_morton(pt_context *ctx, uint64_t *codes /* n, primitive order */);
int pt_download_permutation(pt_context *ctx, int32_t *perm /* n, sorted position -> primitive */);
int pt_download_sorted_primitives(pt_context *ctx, pt_PrimitiveRayCS *out /* n: ray_cs_primitives */);
/* tree_uav in the reference's layout.  The build keeps the tree as traversal records (both child
 * boxes per internal node); the 2n TreeNode records are derived from them on the first call after
 * a build (links from the closed form, boxes = unions of the child boxes: the refit's own values). */
int pt_download_tree(pt_context *ctx, pt_TreeNode *out /* 2n: tree_uav */);

/* Build that allocates nothing.  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot
 * allocate device memory -- code under HIP graph capture, tight loops that rebuild a small dynamic
 * scene every frame, harnesses that forbid hipMalloc.  It runs the kernels of pt_build
 * (PT_BUILD_TREE_ONLY form) over caller-owned inputs and outputs, all DEVICE pointers:
 *   dev_prims  n PrimitiveRender records (16-byte aligned), read only;
 *   dev_morton n uint64 codes in primitive order, or NULL;
 *   dev_perm   n int32: sorted position -> primitive;
 *   dev_tree   2n TreeNode records in the reference's layout (16-byte aligned): tree_uav.
 * Every intermediate buffer (boxes, codes, sort ping-pong, digit counts, partial boxes, node
 * records: about 108 bytes per primitive) lives in one static device array of 14 MiB that each GPU
 * the library runs on holds until the process ends; hence the limit n <= PT_NOALLOC_MAX_PRIMS, beyond
 * which the call fails with PT_ERR_UNSUPPORTED.  flags: 0 or PT_BUILD_KARRAS_LINKS.  The context's
 * own scene, tree and images are not touched.  Work is enqueued on the context's stream; no
 * hipMalloc / hipFree / synchronisation happens inside.  All calls on one GPU share the array (also
 * from different contexts): calls that overlap in time on one GPU race and must be serialised by the
 * caller (one stream, or events). */
#define PT_NOALLOC_MAX_PRIMS 131072
int pt_build_tree_noalloc(pt_context *ctx, const void *dev_prims, int n, unsigned flags,
                          uint64_t *dev_morton, int32_t *dev_perm, pt_TreeNode *dev_tree);

/* ---- images -----------------------------------------------------------------
 * replaces: RayTracer::resize (ray_tracer.cpp:188-261): (re)creates the
 * accumulation / display / G-buffer images and resets the frame counter. */
int pt_resize(pt_context *ctx, int width, int height);
/* replaces: CopyResource(RT0/1/2 -> gbuffer/gbuffer2/gbuffer3) (CS.cpp:369-390):
 * colour RGBA8 (the bytes the shader samples as UNORM), normal.xyz + asfloat(matid+1)
 * RGBA32F, world position RGBA32F; row-major, width*height texels each. */
int pt_upload_gbuffer(pt_context *ctx, const uint8_t *diffuse_rgba8, const float *normal_matid_rgba32f,
                      const float *worldpos_rgba32f);
int pt_download_gbuffer(pt_context *ctx, uint8_t *diffuse_rgba8, float *normal_matid_rgba32f,
                        float *worldpos_rgba32f);

typedef struct pt_camera {
    float eye[3], at[3], up[3];
    float fovy, znear, zfar;
} pt_camera;
/* stands in for: the rasterised G-buffer pass (fixed_pipeline.hlsl:49-67,
 * CS.cpp:309-393): casts one ray per pixel centre.  albedo_rgba8[16][4] are the
 * bytes the colour target holds per material (pt_albedo_bytes computes them). */
int  pt_primary_hits(pt_context *ctx, const pt_camera *cam, const uint8_t *albedo_rgba8);
void pt_albedo_bytes(const float kd[3], int srgb_quirk, uint8_t out_rgb[3]);

/* ---- trace --------------------------------------------------------------------
 * replaces: RayTracer::trace (ray_tracer.cpp:281-335 -> ray_tracing.hlsl:CSMain)
 * including the CopyResource(old <- output) and frame_count += 1.
 * One call = one frame: 1 path per pixel, literal accumulate
 *   out = pow(pow(old, 2.2) * f/(f+1) + sample/(f+1), 1/2.2),  f = framecount.
 * `seed` is g_random[0] of that frame. */
int pt_trace(pt_context *ctx, uint32_t framecount, uint32_t seed);
/* replaces: ONE Dispatch of the tile loop of RayTracer::trace (ray_tracer.cpp:302-332) with
 * CB_RT.offset = {x0, y0, x1, y1}: the shader adds (x0, y0) to the thread id and returns when
 * x > x1 or y > y1 (ray_tracing.hlsl:402-406), so the rectangle x0..x1 by y0..y1, bounds INCLUSIVE
 * and clipped to the image, receives frame `framecount` (literal accumulate); every other pixel
 * keeps its value (output and old are whole-image copies of each other).  The reference draws a new
 * seed per tile; pass the same seed to every tile to get exactly the image of one pt_trace.  Sets
 * the frame counter to framecount + 1, as the last tile of a frame does.  Always the persistent
 * schedule. */
int pt_trace_region(pt_context *ctx, uint32_t framecount, uint32_t seed, int x0, int y0, int x1, int y1);
/* `count` consecutive frames first_frame .. first_frame+count-1 in ONE launch,
 * each pixel running its frames back to back; seeds[i] belongs to frame
 * first_frame+i (host pointer).  mode PT_ACCUM_LITERAL gives exactly the images
 * `count` pt_trace calls give; PT_ACCUM_LINEAR adds the linear samples into a
 * float RGBA sum (alpha counts samples) that pt_resolve turns into the images.
 * Memory: the samples of a launch are staged on the device, 16 bytes per pixel and frame, in a
 * buffer that grows to the largest launch seen and is capped at an eighth of the device's memory
 * (at least 4 GiB); a call with more frames than fit is split into several launches. */
#define PT_ACCUM_LITERAL 0
#define PT_ACCUM_LINEAR  1
int pt_trace_frames(pt_context *ctx, uint32_t first_frame, uint32_t count, const uint32_t *seeds, int mode);
/* seeds of frames first..first+count-1 under the fixed-seed convention: output
 * number 16*f of the 32-bit Mersenne twister seeded with base_seed (the reference
 * draws 16 values per frame from std::random_device and uses the first:
 * ray_tracer.cpp:293-318). */
void pt_frame_seeds(uint32_t base_seed, uint32_t first_frame, uint32_t count, uint32_t *out);
/* frame counter (RayTracer::frame_count) */
int pt_get_frame_count(pt_context *ctx, uint32_t *frames);
int pt_reset_accumulation(pt_context *ctx);            /* frame_count = 0, images cleared */

/* linear-sum path: device pointer of the float RGBA sum (width*height*16 bytes) so a
 * caller can reduce it across GPUs (RCCL) before resolving; pt_resolve writes
 * pow(sum.rgb / sum.a, 1/2.2) to the output image and the display image. */
int pt_linear_sum_device_ptr(pt_context *ctx, void **dev_ptr, size_t *bytes);
int pt_resolve(pt_context *ctx);

/* ---- multi-GPU merge ------------------------------------------------------------
 * No counterpart in the reference (one GPU, ray_tracer.cpp:281-335).  One process per GPU; every
 * rank traces its own frames in PT_ACCUM_LINEAR mode; pt_reduce_linear_sum then runs ONE in-place
 * RCCL all-reduce (sum, float32, width*height*4 elements) of the linear sums on the context's
 * stream, after which every rank holds the sums (and sample counts) of the whole job and
 * pt_resolve gives the final image.  `nccl_comm` is an ncclComm_t of the caller (same device), or
 * NULL to use the communicator made by pt_comm_init_rank.  librccl is loaded at the first of these
 * calls; a process that already holds one (torch.distributed) shares it.
 * pt_comm_unique_id: rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other ranks
 * by any means; pt_comm_init_rank (collective: every rank calls it) is ncclCommInitRank on the
 * context's device; pt_comm_destroy releases it (pt_destroy does too); pt_comm_count reports the
 * number of ranks of the context's communicator as RCCL itself states it (ncclCommCount; 0 when the
 * context has none), so a harness can check that the merge really spans the ranks it launched. */
typedef struct pt_comm_id { char bytes[128]; } pt_comm_id;
int pt_comm_unique_id(pt_comm_id *out);                /* errors: pt_last_error(NULL) */
int pt_comm_init_rank(pt_context *ctx, int n_ranks, int rank, const pt_comm_id *id);
int pt_comm_destroy(pt_context *ctx);
int pt_comm_count(pt_context *ctx, int *n_ranks);
int pt_reduce_linear_sum(pt_context *ctx, void *nccl_comm);

/* outputs: `output` RGBA32F (textures_[output_tex_index]) and `output_u` RGBA8
 * (textures_[output_unorm_tex_index], what CS.cpp:503,511 reads) */
int pt_download_output(pt_context *ctx, float *rgba32f);
int pt_download_display(pt_context *ctx, uint8_t *rgba8);
int pt_output_device_ptr(pt_context *ctx, void **dev_rgba32f, void **dev_rgba8);

/* ---- options & counters --------------------------------------------------------- */
/* traversal stack depth.  The reference keeps 15 entries per path and silently drops subtrees
 * beyond them (ray_tracing.hlsl:235-238), which happens for trees deeper than 13 levels
 * (n > 16383).  This library never drops (it keeps one pending child per tree level), so for
 * such trees its images differ from the reference's (NON-PARITY; they equal the reference
 * algorithm run with a stack that never overflows).
 *   0  (default) and 32: trace any tree with the non-dropping traversal;
 *   15: strict-parity guard -- the trace calls refuse trees deeper than the reference's stack
 *       handles, so every image a context in this mode produces is one the reference produces.
 * Allowed: 0, 15, 32. */
int pt_set_stack_size(pt_context *ctx, int entries);

/* order in which the two children of a node are visited.  PT_TRAVERSAL_REFERENCE (default): left
 * child first, always (ray_tracing.hlsl:228-233 pushes right, then left).  PT_TRAVERSAL_NEAR_FIRST:
 * when both children pass their box test, the one the ray enters first (ties: left) -- far fewer node
 * visits on large scenes, where the reference's order walks the whole ray before it meets the
 * nearest hit.  NON-PARITY: the closest hit is the same triangle except where two triangles are hit
 * at bit-equal or rounding-close distances (the reference keeps the first in ITS visit order);
 * node-visit counters differ.  Persistent schedule only. */
#define PT_TRAVERSAL_REFERENCE  0
#define PT_TRAVERSAL_NEAR_FIRST 1
int pt_set_traversal_order(pt_context *ctx, int order);

/* No counterpart in the reference.  The reference's run loop issues one Dispatch per frame; a one-frame launch ends in a
 * long tail (every lane's last path runs in a wave that empties).  By default short whole-image trace calls (at most 8
 * frames, tree in LDS, persistent schedule) are therefore OVERLAPPED: their work runs on two library-owned streams used
 * in turn, so that the next call's paths fill the compute units the previous call's tail leaves idle; the accumulates
 * keep the order of the calls, and the context's stream waits for each call as soon as it is enqueued, so downloads,
 * other entry points and the caller's own reads of the images on that stream see finished frames exactly as before.
 * Overlap is off while a stream of the caller's is set (pt_set_stream with a non-null stream: the library cannot see
 * the caller's own kernels on it) and with enable = 0 (callers that capture the context's work in a HIP graph). */
int pt_set_trace_overlap(pt_context *ctx, int enable);

/* longest path, in rays after the G-buffer hit.  The reference ends a path when its counter `itr`
 * (1 at the first bounce, +1 per hit) reaches MAX_ITR * 3 = 15, i.e. after 14 intersect() calls
 * (ray_tracing.hlsl:245-247, 372-387).  0 (default) or 14 keeps that; 1..13 end paths earlier
 * (NON-PARITY: the depth-limited configuration of large synthetic scenes).  Russian roulette still
 * starts after itr > 5. */
int pt_set_max_segments(pt_context *ctx, int segments);

/* how pt_trace / pt_trace_frames schedule the paths of a launch on scenes whose tree fits
 * in LDS (larger scenes always use one kernel with the tree in global memory).  No
 * counterpart in the reference (one Dispatch per frame, ray_tracer.cpp:281-335); the
 * images, counters and frame counter do not depend on the choice.
 *   PT_SCHEDULE_PERSISTENT (default): one persistent kernel; a lane runs whole paths.
 *   PT_SCHEDULE_WAVEFRONT: first-bounce kernel, then one intersect kernel and one
 *     shading kernel per bounce over queues of live paths (112 bytes of queue per path
 *     of a launch). */
#define PT_SCHEDULE_PERSISTENT 0
#define PT_SCHEDULE_WAVEFRONT  1
int pt_set_trace_schedule(pt_context *ctx, int schedule);

typedef struct pt_counters {
    uint64_t paths;        /* pixel-frames traced                        */
    uint64_t rays;         /* intersect() calls = BVH traversals         */
    uint64_t node_visits;  /* stack pops (only when counting is enabled) */
    uint64_t box_tests;
    uint64_t tri_tests;
} pt_counters;
/* detailed counting (node_visits/box_tests/tri_tests) uses an instrumented
 * kernel; paths and rays are always counted */
int pt_set_detailed_counters(pt_context *ctx, int enabled);
int pt_get_counters(pt_context *ctx, pt_counters *out);   /* synchronises */
int pt_reset_counters(pt_context *ctx);

/* Environment variables read ONCE, by pt_create, for experiments (unset = defaults; none changes a
 * result): PTAMD_REFILL_T / PTAMD_LEAF_T (wave-vote thresholds of the persistent kernel, 20 / 12),
 * PTAMD_WF_REFILL_T / PTAMD_WF_LEAF_T (same, wavefront schedule), PTAMD_SAMPLES_GIB (cap of the
 * sample staging buffer), PTAMD_NO_LDS_TREE=1 (node records from global memory even when they fit
 * in LDS), PTAMD_MAX_THREADS (fewer waves per CU), PTAMD_NO_ORDER=1 (image order instead of
 * most-expensive-first), PTAMD_CHUNK (frames per work item), PTAMD_PRINT_SCHED=1 (scheduling
 * statistics of the instrumented kernel on stderr at pt_get_counters), PTAMD_PAIR_MAX_MIB (trees that do not
 * fit in LDS are traversed through an array of 128-byte lines holding the records of a node's two children, made at
 * the first trace after a build, as long as that array stays below this size: 65536; 0 = always the plain records),
 * PTAMD_PAIR_MIN_MIB (... and only when the plain records take at least this much, 24: smaller trees come from the
 * caches and are read as plain records, their upper levels from LDS unless PTAMD_NO_TOP_LEVELS=1),
 * PTAMD_DUAL=0 (trees in LDS: the single-path persistent kernel instead of the one with two paths per lane, which
 * launches with at least PTAMD_DUAL_MIN_CHUNK = 4 frames per work item use),
 * PTAMD_DUAL_REFILL_T / PTAMD_SHADE_T / PTAMD_SWAP_T / PTAMD_DUAL_LEAF_T (wave-vote thresholds of the latter: 4 / 48 / 4 / 8),
 * PTAMD_GEN_LDS_ENTRIES (PT_BUILD_TOPOLOGY_MORTON trees: entries of each lane's traversal stack kept in LDS, the deeper
 * ones live in global memory; default: as many as fit, 19 at 1024 threads),
 * PTAMD_EXTRA_WAVES (trees in LDS, two-path kernel: most waves per CU that run beyond those whose traversal stacks fit in
 * LDS beside the tree, with their stack entries in global memory; 16 = fill the workgroup to 1024 threads, 0 = none),
 * PTAMD_FINE_FRAC / PTAMD_FINE_CHUNK (two-path kernel: the last 1/FRAC of a launch's frames is handed out in work items of
 * CHUNK frames instead of the launch's item length, so that the launch ends on short items: 4 / 8; FRAC 0 = off),
 * PTAMD_SORT_CHAINED=0 (PT_BUILD_MORTON_FULL: a counting pass and a scan before each of the eight radix passes instead of
 * one count of all digits up front and passes that find their offsets by looking back over the tiles before them). */

/* device-side duration of the trace kernel of every launch since timing was enabled (or since the
 * last pt_get_trace_times): HIP events on the context's stream around the persistent trace kernel
 * alone (PT_SCHEDULE_WAVEFRONT: around the first-bounce .. last shading kernel), not around the
 * accumulate or queue-order kernels.  pt_get_trace_times synchronises, writes at most `capacity`
 * durations (ms, oldest first), returns how many launches were timed in *count and forgets them;
 * the latest 4096 launches are kept when nobody reads.
 * No counterpart in the reference. */
int pt_set_trace_timing(pt_context *ctx, int enabled);
int pt_get_trace_times(pt_context *ctx, float *ms, int capacity, int *count);

/* single closest-hit queries through the traversal kernel, for tests:
 * rays = n_rays x {origin.xyz, t_min, direction.xyz, t_max}; results per ray:
 * hit (0/1), primitive, t, barycentrics (u, v, 1-u-v) */
typedef struct pt_hit {
    int32_t hit;
    int32_t primitive;
    float   t;
    float   bary[3];
} pt_hit;
int pt_intersect(pt_context *ctx, const float *rays8, int n_rays, pt_hit *out);

#ifdef __cplusplus
}
#endif
#endif /* PTAMD_H */
