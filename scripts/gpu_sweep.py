"""Timing sweep of the trace kernel's wave-scheduling thresholds (scene04, 1162x1008)."""
import os, sys, time, itertools
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT)
import ptamd
g = np.load(os.path.join(ROOT, "tests", "golden", "scene04.npz"))
ctx = ptamd.Context(0)
ctx.upload_primitives(g["prims"].view(ptamd.PRIM_RENDER)); ctx.set_constants(g["cb"].view(ptamd.CB_RADIX))
ctx.build(); ctx.resize(1162, 1008)
ctx.primary_hits(ptamd.default_camera(), ptamd.albedo_table(g["mat_table"][:, 0:3]))
def run(nf, reps=3):
    seeds = ptamd.frame_seeds(1, 0, nf)
    best = 1e9
    for _ in range(reps):
        ctx.reset_accumulation(); ctx.reset_counters(); ctx.synchronize()
        t = time.time(); ctx.trace_frames(0, seeds, ptamd.PT_ACCUM_LINEAR); ctx.synchronize(); best = min(best, time.time() - t)
    return ctx.counters()["rays"] / best / 1e6
frames = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [32]
rts = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [16]
lts = [int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else [16]
for nf in frames:
    for r, l in itertools.product(rts, lts):
        os.environ["PTAMD_REFILL_T"] = str(r); os.environ["PTAMD_LEAF_T"] = str(l)
        print(f"frames={nf} refill_t={r} leaf_t={l}: {run(nf):.0f} Mrays/s", flush=True)
