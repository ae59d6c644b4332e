"""Runs the other named configurations once: the four scenes at 1162x1008 and 3840x2160, and a large soup render
(global-memory tree, deep stack) -- timings only."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ptamd
from conftest import make_soup, soup_cb

def run(ctx, W, H, frames, label):
    seeds = ptamd.frame_seeds(1, 0, frames)
    ctx.reset_accumulation(); ctx.reset_counters(); ctx.synchronize()
    t = time.time(); ctx.trace_frames(0, seeds, ptamd.PT_ACCUM_LINEAR); ctx.synchronize(); dt = time.time() - t
    c = ctx.counters()
    print(f"{label}: {W}x{H} x{frames} frames {dt*1e3:.1f} ms  {c['rays']/dt/1e6:.0f} Mrays/s  {c['paths']/dt/1e6:.0f} Mpaths/s  rays/path {c['rays']/max(1,c['paths']):.2f}", flush=True)

for scene in (() if os.environ.get("SOUP_ONLY") else ("scene01", "scene02", "scene03", "scene04")):
    g = np.load(os.path.join(ROOT, "tests", "golden", scene + ".npz"))
    ctx = ptamd.Context(0)
    ctx.upload_primitives(g["prims"].view(ptamd.PRIM_RENDER)); ctx.set_constants(g["cb"].view(ptamd.CB_RADIX)); ctx.build()
    for (W, H) in ((1162, 1008), (3840, 2160)):
        ctx.resize(W, H)
        ctx.primary_hits(ptamd.default_camera(), ptamd.albedo_table(g["mat_table"][:, 0:3]))
        run(ctx, W, H, 32, scene); run(ctx, W, H, 32, scene)
    ctx.close()

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
prims = make_soup(n, seed=12345, dtype=ptamd.PRIM_RENDER)
# The triangle test rejects |det| <= 1e-4 on the unnormalised edges (ray_tracing.hlsl:172), so a
# unit-cube soup of millions of triangles is invisible to it; scale the cube to 100 units.
prims["vertices"]["pos"] *= np.float32(100.0)
prims["matid"] = (np.arange(n) % 16 == 0) * 3          # 1/16 emitters, rest diffuse
ctx = ptamd.Context(0)
ctx.upload_primitives(prims); ctx.set_constants(soup_cb(n, eye=(50.0, 50.0, 300.0))) 
for _ in range(2): ctx.build(ptamd.PT_BUILD_MORTON_FULL)
print("soup build (full codes)", ctx.build_times(), flush=True)
ctx.set_stack_size(32)
if os.environ.get("DEPTH"): ctx.set_max_segments(int(os.environ["DEPTH"]))     # e.g. 8: the depth-limited 100M configuration
SW, SH = int(os.environ.get("SOUP_W", 1920)), int(os.environ.get("SOUP_H", 1080))
ctx.resize(SW, SH)
kd = [[0.7, 0.6, 0.5]] * 16
cam = ptamd.default_camera(eye=(50.0, 50.0, 300.0), at=(50.0, 50.0, 50.0)); cam.zfar = 1000.0
ctx.primary_hits(cam, ptamd.albedo_table(kd))
run(ctx, SW, SH, 4, f"soup {n} (21-bit codes)"); run(ctx, SW, SH, 4, f"soup {n} (21-bit codes)")
for _ in range(2): ctx.build()
print("soup build (reference codes)", ctx.build_times(), flush=True)
ctx.primary_hits(cam, ptamd.albedo_table(kd))
run(ctx, SW, SH, 1, f"soup {n} (reference 22-level codes)")
ctx.close()
