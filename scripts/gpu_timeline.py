"""Wave timeline of one 128-frame launch of scene04 (needs the `tl` variant build, which records wall-clock stamps per wave in
the scheduling counters): when the first and the last wave's queue ran dry and when the first, the average and the last wave ended."""
import os, sys, re, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT)
os.environ["PTAMD_PRINT_SCHED"] = "1"
import ptamd
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 128
g = np.load(os.path.join(ROOT, "tests", "golden", "scene04.npz"))
ctx = ptamd.Context(0)
ctx.upload_primitives(g["prims"].view(ptamd.PRIM_RENDER)); ctx.set_constants(g["cb"].view(ptamd.CB_RADIX))
ctx.build(); ctx.resize(1162, 1008)
ctx.primary_hits(ptamd.default_camera(), ptamd.albedo_table(g["mat_table"][:, 0:3], True))
for s in range(3):
    ctx.trace_frames(frames * s, ptamd.frame_seeds(1, frames * s, frames), ptamd.PT_ACCUM_LINEAR)
ctx.synchronize(); ctx.reset_counters()
sys.stderr.flush()
ctx.trace_frames(frames * 3, ptamd.frame_seeds(1, frames * 3, frames), ptamd.PT_ACCUM_LINEAR)
ctx.synchronize()
print("MARK", flush=True)
ctx.counters()
ctx.close()
