// RayTracer: the reference's host-side class (reference: ray_tracer.hpp:7-78) on
// top of the headless HIP library instead of the D3D11 ShaderManager /
// BufferManager base classes.  Method names, argument meaning and the public
// data members are the reference's; members that were D3D resource handles are
// kept as plain ints so existing call sites compile, and a few headless
// additions replace what the DXUT application shell supplied (camera, material
// constants, G-buffer, image read-back).
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ptamd.h"
#include "scene.hpp"
#include "structs.hpp"

class RayTracer
{
public:
    RayTracer();
    ~RayTracer();
    int frame_count = 0;
    int primitive_count = 0;

    // resource handles of the reference; kept for source compatibility, all -1
    int radix_build_shader_index = -1, bvh_build_shader_index = -1, ray_shader_index = -1;
    int tree_uav_index = -1, radix_cb_index = -1;
    int primitive_ray_srv_index = -1, bvh_build_lock_uav_index = -1;
    int primitive_render_srv_index = -1, output_tex_uav_index = -1, output_unorm_tex_uav_index = -1, rt_cb_index = -1;
    int old_tex_index = -1, old_tex_srv_index = -1;
    int output_tex_index = -1, output_unorm_tex_index = -1;
    int gbuffer_tex_index = -1, gbuffer_srv_index = -1, gbuffer2_tex_index = -1, gbuffer2_srv_index = -1;
    int gbuffer3_tex_index = -1, gbuffer3_srv_index = -1;
    int gbuffer4_tex_index = -1, gbuffer4_srv_index = -1;
    int halton_tex_index = -1, halton_tex_srv_index = -1;
    // the shader file names of the reference (LPCWSTR there); nothing is compiled at run time here
    const wchar_t *const RADIX_SHADER_NAME = L"radix_build_cs.hlsl";
    const wchar_t *const BVH_SHADER_NAME = L"bvh_build_cs.hlsl";
    const wchar_t *const RAY_SHADER_NAME = L"ray_tracing.hlsl";

    Scene scene;
    Bound bb_;
    std::vector<PrimitiveRender> renderer_primitives;
    std::vector<PrimitiveRay> ray_primitives;          // sorted by Morton code, ties by index
    std::vector<PrimitiveRayCS> ray_cs_primitives;

    // reference: ray_tracer.cpp:18-70.  Flattens the scene on the host; boxes, scene box,
    // Morton codes and the sort run on the GPU, and the members above are filled from the
    // device results.
    void loadScene(const char *filename);
    void loadShaders();                                 // kernels are compiled ahead of time: no-op
    void createBuffers();                               // reference: ray_tracer.cpp:91-172
    void run();                                         // reference: ray_tracer.cpp:174-186
    // the reference's signature, run(ID3D11DeviceContext *): the pointer is kept as the reference keeps it and never used
    void *pd3dImmediateContext_ = nullptr;
    void run(void *pd3dImmediateContext) { pd3dImmediateContext_ = pd3dImmediateContext; run(); }
    void resize(int width, int height);                 // reference: ray_tracer.cpp:188-261
    void buildRadixTree();                              // reference: ray_tracer.cpp:263-269
    void buildBVH();                                    // reference: ray_tracer.cpp:271-279 (fused into the build)
    void trace();                                       // reference: ray_tracer.cpp:281-335
    void finish();                                      // nothing to unbind
    void release();

    int width = 640;
    int height = 480;
    std::unordered_map<std::string, std::string> shader_define_values;

    // ---- headless additions ----------------------------------------------------
    int device = 0;
    // per-frame seeds: output 16*f of mt19937(base_seed); use_random_device = true draws
    // from std::random_device as the reference does (ray_tracer.cpp:293-318)
    uint32_t base_seed = 1;
    bool use_random_device = false;
    bool srgb_albedo_quirk = true;
    // run() queues its frame; queued frames are traced by ONE launch when this many are waiting or
    // when a result is needed (any read, snapshot, resize, new constants).  The images are those of
    // one launch per frame; a launch of 32 frames takes half the time of 32 launches.  1 = launch in run().
    int frames_per_launch = 32;
    void flush();
    // flags handed to pt_build by loadScene() and buildRadixTree().  PT_BUILD_DEFAULT builds the
    // reference's tree (radix_build_cs.hlsl:30-96, bvh_build_cs.hlsl:17-55); PT_BUILD_MORTON_FULL / PT_BUILD_TOPOLOGY_MORTON
    // are the labelled non-parity options for large scenes.  Set before loadScene().
    unsigned build_flags = PT_BUILD_DEFAULT;
    // the reference's TILEX / TILEY (ray_tracer.cpp:298-299, compiled as 1): the image is traced in
    // tiles_x * tiles_y dispatches per frame, each over width / tiles_x by height / tiles_y pixels with
    // the offset and inclusive bounds of ray_tracer.cpp:319-322.  Values above 1 launch inside run()
    // (frames are not batched).
    int tiles_x = 1, tiles_y = 1;
    // The reference fills random[] inside the tile loop (ray_tracer.cpp:315-318): a new seed per DISPATCH, 16
    // draws each.  false (default): every tile of a frame uses the frame's seed, so the image does not depend
    // on the tiling.  true: the reference's literal behaviour -- dispatch number d (counted over all frames,
    // tiles of a frame in loop order) uses output 16*d of mt19937(base_seed), or 16 fresh draws of
    // random_device; with one tile per frame the two settings are the same.
    bool tile_seed_per_dispatch = false;
    float eye[3] = {0.0f, 5.0f, 23.0f};                 // CS.cpp:49
    float at[3] = {0.0f, 5.0f, 0.0f};                   // CS.cpp:50
    // material constants + eye into CB_Radix (reference: CS.cpp:405-439)
    void updateConstants();
    // G-buffer by ray casting through the camera of CS.cpp:200-201,318-329
    void renderGBuffer();
    std::vector<float> readOutput();                    // RGBA32F, width*height*4
    std::vector<uint8_t> readDisplay();                 // RGBA8
    std::vector<TreeNode> readTree();                   // 2 * primitive_count nodes
    bool saveBMP(const char *path);                     // the display image (CS.cpp:507-514)
    // the frame counts at which the reference saves "<width>X<height>frame_<count>.bmp" (CS.cpp:446-479)
    static bool snapshotDue(int frame_count);
    // saveBMP under that name in `dir` when frame_count is on the schedule; returns whether a file was written
    bool saveScheduledSnapshot(const char *dir);
    const std::string &lastError() const { return error_; }
    bool ok() const { return error_.empty(); }
    pt_context *context() { return ctx_; }

private:
    pt_context *ctx_ = nullptr;
    bool has_built_ = false;
    bool gbuffer_ready_ = false;
    int pending_first_ = 0;                 // first of the queued frames
    std::vector<uint32_t> pending_seeds_;   // their seeds; frame pending_first_ + i uses [i]
    std::string error_;
    bool check(int rc, const char *what);
    bool ensureContext();
};
