#include "ray_tracer.hpp"

#include <cstdio>
#include <cstring>
#include <random>

RayTracer::RayTracer() { std::memset(&bb_, 0, sizeof(bb_)); }

RayTracer::~RayTracer() { release(); }

bool RayTracer::check(int rc, const char *what)
{
    if (rc == PT_OK) return true;
    error_ = std::string(what) + ": " + pt_last_error(ctx_);
    return false;
}

bool RayTracer::ensureContext()
{
    if (ctx_) return true;
    if (pt_create(device, &ctx_) != PT_OK) { error_ = std::string("pt_create: ") + pt_last_error(nullptr); ctx_ = nullptr; return false; }
    return true;
}

void RayTracer::loadScene(const char *filename)
{
    flush();
    error_.clear();
    if (!scene.loadObj(filename)) { error_ = "loadObj: " + scene.error; return; }
    // the constant buffer holds 16 materials (structs.hlsli); the reference indexes past it silently
    if (scene.materials_.size() > (size_t)PT_MAX_MATERIALS) {
        error_ = "loadScene: " + std::to_string(scene.materials_.size()) + " materials (the default one included); the constant buffer holds " +
                 std::to_string(PT_MAX_MATERIALS);
        return;
    }

    // shape -> material group -> face order; global index; pad = -1
    renderer_primitives.resize((size_t)scene.triangle_count);
    primitive_count = 0;
    for (const auto &geo : scene.geos) {
        for (size_t pg = 0; pg < geo.indicess.size(); ++pg) {
            const auto &indices = geo.indicess[pg];
            for (size_t t = 0; t < indices.size() / 3; ++t) {
                PrimitiveRender &prd = renderer_primitives[(size_t)primitive_count + t];
                prd.matid = (int32_t)geo.material_ids[pg];
                prd.pad = -1;
                for (int k = 0; k < 3; k++) prd.vertices[k] = geo.vertices_xyzn[indices[t * 3 + k]];
            }
            primitive_count += (int)(indices.size() / 3);
        }
    }
    if (primitive_count < 1) { error_ = "loadScene: no triangles"; return; }

    // boxes, scene box, Morton codes, sort (and the tree) on the GPU
    if (!ensureContext()) return;
    if (!check(pt_upload_primitives(ctx_, renderer_primitives.data(), primitive_count), "pt_upload_primitives")) return;
    if (!check(pt_build(ctx_, build_flags), "pt_build")) return;
    has_built_ = true;
    const size_t n = (size_t)primitive_count;
    std::vector<uint64_t> codes(n);
    std::vector<int32_t> perm(n);
    ray_cs_primitives.resize(n);
    if (!check(pt_download_scene_bound(ctx_, &bb_), "pt_download_scene_bound")) return;
    if (!check(pt_download_morton(ctx_, codes.data()), "pt_download_morton")) return;
    if (!check(pt_download_permutation(ctx_, perm.data()), "pt_download_permutation")) return;
    if (!check(pt_download_sorted_primitives(ctx_, ray_cs_primitives.data()), "pt_download_sorted_primitives")) return;
    ray_primitives.resize(n);
    for (size_t i = 0; i < n; i++) {
        ray_primitives[i].morton_code = codes[(size_t)perm[i]];
        ray_primitives[i].bound = ray_cs_primitives[i].bound;
        ray_primitives[i].index = ray_cs_primitives[i].index;
    }
    shader_define_values["MAT_COUNT"] = std::to_string(scene.materials_.size());
    shader_define_values["MAT_ITR"] = "5";
}

void RayTracer::loadShaders() {}

void RayTracer::createBuffers()
{
    if (!ensureContext()) return;
    resize(width, height);
    updateConstants();
}

void RayTracer::updateConstants()
{
    if (!ctx_) return;
    flush();      // queued frames belong to the old constants / G-buffer
    CB_Radix cb;
    std::memset(&cb, 0, sizeof(cb));
    cb.node_count = primitive_count;
    for (int k = 0; k < 3; k++) cb.pos.v[k] = eye[k];
    size_t i = 0;
    for (const auto &m : scene.materials_) {
        if (i >= PT_MAX_MATERIALS) break;
        Material &mat = cb.mats[i++];
        mat.ke = m.emission[0] * 10;
        mat.ns = m.shininess;
        if (mat.ke <= 0) { for (int k = 0; k < 3; k++) mat.kd.v[k] = m.diffuse[k]; }
        else { mat.kd.v[0] = mat.kd.v[1] = mat.kd.v[2] = 1; }
        for (int k = 0; k < 3; k++) mat.ks.v[k] = m.specular[k];
        mat.tr = 1 - m.dissolve;
        mat.ni = m.ior;
    }
    check(pt_set_constants(ctx_, &cb), "pt_set_constants");
}

void RayTracer::renderGBuffer()
{
    if (!ctx_) return;
    flush();      // queued frames belong to the old constants / G-buffer
    uint8_t albedo[PT_MAX_MATERIALS][4];
    std::memset(albedo, 0, sizeof(albedo));
    size_t i = 0;
    for (const auto &m : scene.materials_) {
        if (i >= PT_MAX_MATERIALS) break;
        pt_albedo_bytes(m.diffuse, srgb_albedo_quirk ? 1 : 0, albedo[i]);
        albedo[i][3] = 255;
        i++;
    }
    pt_camera cam;
    for (int k = 0; k < 3; k++) { cam.eye[k] = eye[k]; cam.at[k] = at[k]; cam.up[k] = (k == 1) ? 1.0f : 0.0f; }
    cam.fovy = 3.141592654f / 6;            // D3DX_PI / 6 (CS.cpp:201)
    cam.znear = 0.005f;
    cam.zfar = 100.0f;
    if (check(pt_primary_hits(ctx_, &cam, &albedo[0][0]), "pt_primary_hits")) gbuffer_ready_ = true;
}

void RayTracer::run()
{
    if (!has_built_) {
        buildRadixTree();
        buildBVH();
        has_built_ = true;
    }
    trace();
    finish();
}

void RayTracer::resize(int w, int h)
{
    flush();
    width = w;
    height = h;
    frame_count = 0;
    gbuffer_ready_ = false;
    if (ensureContext()) check(pt_resize(ctx_, w, h), "pt_resize");
}

void RayTracer::buildRadixTree()
{
    flush();                                // queued frames belong to the tree they were queued under
    if (ctx_ && primitive_count > 0) has_built_ = check(pt_build(ctx_, build_flags), "pt_build");
    gbuffer_ready_ = false;
}

void RayTracer::buildBVH() {}

void RayTracer::trace()
{
    if (!ctx_) return;
    if (!gbuffer_ready_) renderGBuffer();
    // seed of dispatch number `dispatch`: 16 draws, the first one is used (ray_tracer.cpp:315-318)
    auto draw_seed = [this](uint32_t dispatch) {
        uint32_t s;
        if (use_random_device) {
            static std::random_device rd;
            static std::uniform_int_distribution<uint32_t> dist(0, 0xFFFFFFFF);
            s = dist(rd);
            for (int i = 1; i < 16; i++) (void)dist(rd);
        } else {
            pt_frame_seeds(base_seed, dispatch, 1, &s);
        }
        return s;
    };
    if (tiles_x > 1 || tiles_y > 1) {
        // ray_tracer.cpp:302-332: for (w = 0; w < width; w += width / TILEX) for (h = 0; h < height; h += height / TILEY),
        // offset = {w, h, w + width / TILEX - 1, h + height / TILEY - 1}.  As there, a width that TILEX does not
        // divide gets one more, narrower column of tiles (the bounds are clipped to the image by the library).
        flush();
        const int tw = width / tiles_x, th = height / tiles_y;
        if (tw < 1 || th < 1) { error_ = "trace: more tiles than pixels"; return; }
        const uint32_t per_frame = (uint32_t)((width + tw - 1) / tw) * (uint32_t)((height + th - 1) / th);
        uint32_t tile = 0;
        uint32_t seed = tile_seed_per_dispatch ? 0u : draw_seed((uint32_t)frame_count);
        for (int w = 0; w < width; w += tw)
            for (int h = 0; h < height; h += th) {
                if (tile_seed_per_dispatch) seed = draw_seed((uint32_t)frame_count * per_frame + tile);
                tile++;
                if (!check(pt_trace_region(ctx_, (uint32_t)frame_count, seed, w, h, w + tw - 1, h + th - 1), "pt_trace_region")) return;
            }
        frame_count += 1;
        return;
    }
    const uint32_t seed = draw_seed((uint32_t)frame_count);
    // a caller that moved frame_count (the reference's ACC toggle sets it to 0) starts a new batch
    if (!pending_seeds_.empty() && frame_count != pending_first_ + (int)pending_seeds_.size()) flush();
    if (pending_seeds_.empty()) pending_first_ = frame_count;
    pending_seeds_.push_back(seed);
    frame_count += 1;
    if ((int)pending_seeds_.size() >= (frames_per_launch < 1 ? 1 : frames_per_launch)) flush();
}

void RayTracer::flush()
{
    if (pending_seeds_.empty() || !ctx_) return;
    check(pt_trace_frames(ctx_, (uint32_t)pending_first_, (uint32_t)pending_seeds_.size(), pending_seeds_.data(),
                          PT_ACCUM_LITERAL), "pt_trace_frames");
    pending_seeds_.clear();
}

void RayTracer::finish() {}

void RayTracer::release()
{
    pending_seeds_.clear();
    if (ctx_) pt_destroy(ctx_);
    ctx_ = nullptr;
    has_built_ = false;
    gbuffer_ready_ = false;
    scene.release();
}

std::vector<float> RayTracer::readOutput()
{
    flush();
    std::vector<float> img((size_t)width * height * 4);
    if (ctx_) check(pt_download_output(ctx_, img.data()), "pt_download_output");
    return img;
}

std::vector<uint8_t> RayTracer::readDisplay()
{
    flush();
    std::vector<uint8_t> img((size_t)width * height * 4);
    if (ctx_) check(pt_download_display(ctx_, img.data()), "pt_download_display");
    return img;
}

std::vector<TreeNode> RayTracer::readTree()
{
    std::vector<TreeNode> t((size_t)primitive_count * 2);
    if (ctx_) check(pt_download_tree(ctx_, t.data()), "pt_download_tree");
    return t;
}

bool RayTracer::snapshotDue(int frame_count)
{
    static const int schedule[] = {50, 100, 200, 500, 1000, 1500, 2000, 3000, 5000, 8000, 10000, 12000, 16000, 20000,
                                   30000, 35000, 40000, 50000, 60000, 70000, 80000, 100000, 120000, 150000, 200000,
                                   300000, 400000};
    for (int f : schedule) if (f == frame_count) return true;
    return false;
}

bool RayTracer::saveScheduledSnapshot(const char *dir)
{
    if (!snapshotDue(frame_count)) return false;
    char name[512];
    std::snprintf(name, sizeof(name), "%s/%dX%dframe_%d.bmp", dir && *dir ? dir : ".", width, height, frame_count);
    return saveBMP(name);
}

bool RayTracer::saveBMP(const char *path)
{
    const std::vector<uint8_t> img = readDisplay();
    if (!ok()) return false;
    FILE *f = std::fopen(path, "wb");
    if (!f) { error_ = std::string("cannot write ") + path; return false; }
    const uint32_t row = ((uint32_t)width * 3 + 3) & ~3u, size = 54 + row * (uint32_t)height;
    uint8_t hdr[54] = {'B', 'M'};
    auto put32 = [&](int off, uint32_t v) { for (int k = 0; k < 4; k++) hdr[off + k] = (uint8_t)(v >> (8 * k)); };
    put32(2, size); put32(10, 54); put32(14, 40); put32(18, (uint32_t)width); put32(22, (uint32_t)height);
    hdr[26] = 1; hdr[28] = 24; put32(34, row * (uint32_t)height);
    std::fwrite(hdr, 1, 54, f);
    std::vector<uint8_t> line(row, 0);
    for (int y = height - 1; y >= 0; y--) {
        for (int x = 0; x < width; x++) {
            const uint8_t *p = &img[((size_t)y * width + x) * 4];
            line[(size_t)x * 3 + 0] = p[2]; line[(size_t)x * 3 + 1] = p[1]; line[(size_t)x * 3 + 2] = p[0];
        }
        std::fwrite(line.data(), 1, row, f);
    }
    std::fclose(f);
    return true;
}
