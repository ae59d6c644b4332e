// Headless driver: what the reference's windowed application does in trace mode
// (reference: CS.cpp:108-117 initScene, :396-514 trace loop and snapshots).
//   pt_headless scene.obj [width height frames out.bmp [snapshot_dir]]
// With snapshot_dir the run also writes the reference's scheduled snapshots
// ("<W>X<H>frame_<count>.bmp" at 50, 100, 200, 500, 1000, ... frames, CS.cpp:446-479,507-514).
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "ray_tracer.hpp"

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s scene.obj [width height frames out.bmp [snapshot_dir]]\n", argv[0]); return 2; }
    RayTracer ray_tracer;
    ray_tracer.width = argc > 2 ? std::atoi(argv[2]) : 1162;
    ray_tracer.height = argc > 3 ? std::atoi(argv[3]) : 1008;
    const int frames = argc > 4 ? std::atoi(argv[4]) : 100;
    const char *out = argc > 5 ? argv[5] : "snapshot.bmp";
    const char *snapshot_dir = argc > 6 ? argv[6] : nullptr;
    // nobody looks at the image between frames here: queue four stratification cycles per launch (the class's default of 32
    // suits an interactive caller); snapshots and the final image flush the queue
    ray_tracer.frames_per_launch = 128;
    // (PT_HEADLESS_FRAMES_PER_LAUNCH: another queue length, e.g. 32, the class's default, or 1, the reference's own loop)
    if (const char *e = std::getenv("PT_HEADLESS_FRAMES_PER_LAUNCH")) { if (std::atoi(e) >= 1) ray_tracer.frames_per_launch = std::atoi(e); }
    ray_tracer.loadScene(argv[1]);
    ray_tracer.loadShaders();
    ray_tracer.createBuffers();
    if (!ray_tracer.ok()) { std::fprintf(stderr, "%s\n", ray_tracer.lastError().c_str()); return 1; }
    const auto t0 = std::chrono::steady_clock::now();
    while (ray_tracer.frame_count < frames && ray_tracer.ok()) {
        ray_tracer.run();
        if (snapshot_dir) ray_tracer.saveScheduledSnapshot(snapshot_dir);
    }
    if (!ray_tracer.saveBMP(out)) { std::fprintf(stderr, "%s\n", ray_tracer.lastError().c_str()); return 1; }
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("%d triangles, %dx%d, %d frames in %.3f s (%.1f frames/s) -> %s\n", ray_tracer.primitive_count,
                ray_tracer.width, ray_tracer.height, ray_tracer.frame_count, s, ray_tracer.frame_count / s, out);
    ray_tracer.release();
    return 0;
}
