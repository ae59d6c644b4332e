"""Scheduling statistics and launch time of the scene04 trace kernel (1162x1008, 128-frame launches) under the
PTAMD_* knobs of the environment.  usage: PTAMD_PRINT_SCHED=1 [PTAMD_DUAL=1 ...] python scripts/gpu_sched_stats.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT)
import ptamd
g = np.load(os.path.join(ROOT, "tests", "golden", "scene04.npz"))
ctx = ptamd.Context(0)
ctx.upload_primitives(g["prims"].view(ptamd.PRIM_RENDER)); ctx.set_constants(g["cb"].view(ptamd.CB_RADIX))
ctx.build(); ctx.resize(1162, 1008)
ctx.primary_hits(ptamd.default_camera(), ptamd.albedo_table(g["mat_table"][:, 0:3], True))
ctx.set_trace_timing(True)
for s in range(4):
    ctx.trace_frames(128 * s, ptamd.frame_seeds(1, 128 * s, 128), ptamd.PT_ACCUM_LINEAR)
ms = ctx.trace_times()
c = ctx.counters()
print("knobs", {k: v for k, v in os.environ.items() if k.startswith("PTAMD_")}, "launch ms", ms[1:].mean(), "Grays/s", c["rays"] / 4 / (ms[1:].mean() * 1e-3) / 1e9, flush=True)
if os.environ.get("PTAMD_PRINT_SCHED"):
    ctx.set_detailed_counters(True); ctx.reset_counters()
    ctx.trace_frames(512, ptamd.frame_seeds(1, 512, 128), ptamd.PT_ACCUM_LINEAR)
    print(ctx.counters())
ctx.close()
