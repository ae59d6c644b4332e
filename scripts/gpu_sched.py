import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT)
import ptamd
os.environ["PTAMD_PRINT_SCHED"] = "1"
g = np.load(os.path.join(ROOT, "tests", "golden", "scene04.npz"))
ctx = ptamd.Context(0)
ctx.upload_primitives(g["prims"].view(ptamd.PRIM_RENDER)); ctx.set_constants(g["cb"].view(ptamd.CB_RADIX))
ctx.build(); ctx.resize(1162, 1008)
ctx.primary_hits(ptamd.default_camera(), ptamd.albedo_table(g["mat_table"][:, 0:3]))
# two plain launches let the queue order settle; the third is the instrumented one
for step in range(2):
    ctx.trace_frames(step * 32, ptamd.frame_seeds(1, step * 32, 32), ptamd.PT_ACCUM_LINEAR)
ctx.reset_counters()
ctx.set_detailed_counters(True)
ctx.trace_frames(64, ptamd.frame_seeds(1, 64, 32), ptamd.PT_ACCUM_LINEAR)
print(ctx.counters())
