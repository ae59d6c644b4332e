"""Render rate and traversal counters of a random soup (tree in global memory): gpu_soup_trace.py [n] [W H frames]."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ptamd
from conftest import make_soup, soup_cb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
W, H, frames = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080, 4)
prims = make_soup(n, seed=12345, dtype=ptamd.PRIM_RENDER)
prims["vertices"]["pos"] *= np.float32(100.0)          # see gpu_configs.py: the reference's |det| > 1e-4 test
prims["matid"] = (np.arange(n) % 16 == 0) * 3
ctx = ptamd.Context(0)
ctx.upload_primitives(prims); ctx.set_constants(soup_cb(n, eye=(50.0, 50.0, 300.0)))
ctx.build(ptamd.PT_BUILD_MORTON_FULL)
ctx.set_stack_size(32)
if os.environ.get("DEPTH"): ctx.set_max_segments(int(os.environ["DEPTH"]))
ctx.resize(W, H)
cam = ptamd.default_camera(eye=(50.0, 50.0, 300.0), at=(50.0, 50.0, 50.0)); cam.zfar = 1000.0
ctx.primary_hits(cam, ptamd.albedo_table([[0.7, 0.6, 0.5]] * 16))
for detailed in (False, True, False):
    ctx.set_detailed_counters(detailed)
    ctx.reset_accumulation(); ctx.reset_counters(); ctx.synchronize()
    t = time.time(); ctx.trace_frames(0, ptamd.frame_seeds(1, 0, frames), ptamd.PT_ACCUM_LINEAR); ctx.synchronize(); dt = time.time() - t
    c = ctx.counters()
    print(f"n={n} {W}x{H}x{frames} detailed={detailed}: {dt*1e3:.1f} ms {c['rays']/dt/1e6:.1f} Mrays/s rays/path {c['rays']/max(1,c['paths']):.2f}"
          + (f" visits/ray {c['node_visits']/c['rays']:.0f} tests/ray {c['tri_tests']/c['rays']:.1f} Gvisits/s {c['node_visits']/dt/1e9:.2f}" if detailed else ""), flush=True)
ctx.close()
