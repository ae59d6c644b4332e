"""Per-launch time of consecutive 32-frame blocks (the queue order adapts after the first)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT)
import ptamd
scene = sys.argv[1] if len(sys.argv) > 1 else "scene04"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1162, 1008)
per = int(sys.argv[4]) if len(sys.argv) > 4 else 32
g = np.load(os.path.join(ROOT, "tests", "golden", scene + ".npz"))
ctx = ptamd.Context(0)
ctx.upload_primitives(g["prims"].view(ptamd.PRIM_RENDER)); ctx.set_constants(g["cb"].view(ptamd.CB_RADIX)); ctx.build()
ctx.resize(W, H)
ctx.primary_hits(ptamd.default_camera(), ptamd.albedo_table(g["mat_table"][:, 0:3]))
ctx.reset_accumulation(); ctx.reset_counters(); ctx.synchronize()
out = []
for step in range(8):
    seeds = ptamd.frame_seeds(1, step * per, per)
    r0 = ctx.counters()["rays"]
    t = time.time(); ctx.trace_frames(step * per, seeds, ptamd.PT_ACCUM_LINEAR); ctx.synchronize(); dt = time.time() - t
    out.append(f"{dt*1e3:.2f}ms/{(ctx.counters()['rays']-r0)/dt/1e6:.0f}")
print(scene, W, H, per, "NO_ORDER" if os.environ.get("PTAMD_NO_ORDER") else "ordered", " ".join(out), flush=True)
