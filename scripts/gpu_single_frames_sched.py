import os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT)
import ptamd
g = np.load(os.path.join(ROOT, "tests", "golden", "scene04.npz"))
ctx = ptamd.Context(0)
ctx.upload_primitives(g["prims"].view(ptamd.PRIM_RENDER)); ctx.set_constants(g["cb"].view(ptamd.CB_RADIX))
ctx.build(); ctx.resize(1162, 1008)
ctx.primary_hits(ptamd.default_camera(), ptamd.albedo_table(g["mat_table"][:, 0:3], True))
seeds = ptamd.frame_seeds(1, 0, 200)
for f in range(20):
    ctx.trace(f, seeds[f])
ctx.set_detailed_counters(True); ctx.reset_counters()
ctx.set_trace_timing(True)
for f in range(20, 36):
    ctx.trace(f, seeds[f])
print(ctx.trace_times().mean(), ctx.counters())
ctx.close()
