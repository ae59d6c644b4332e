"""scene04 padded with k tiny triangles (same view, larger tree): launch time of 128 frames at 1162x1008 against the number of
waves whose stacks fit in LDS beside the tree.  Run under PTAMD_EXTRA_WAVES=0 / 16 and PTAMD_NO_LDS_TREE=1 to compare.
usage: python scripts/gpu_extra_waves.py [k ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT)
import ptamd
g = np.load(os.path.join(ROOT, "tests", "golden", "scene04.npz"))
base = g["prims"].view(ptamd.PRIM_RENDER)
pos = base["vertices"]["pos"].reshape(-1, 3)
lo, hi = pos.min(axis=0), pos.max(axis=0)
for k in [int(a) for a in sys.argv[1:]] or [0, 84, 184, 384, 584]:
    rng = np.random.default_rng(k)
    pad = np.zeros(k, dtype=ptamd.PRIM_RENDER)
    c = (lo + (hi - lo) * rng.random((k, 3))).astype(np.float32)
    for v in range(3):
        pad["vertices"]["pos"][:, v] = c + (rng.random((k, 3), dtype=np.float32) - 0.5) * np.float32(2e-3)
    pad["vertices"]["normal"] = 1.0
    pad["matid"] = 0
    pad["pad"] = -1
    prims = np.concatenate([base, pad])
    cb = g["cb"].view(ptamd.CB_RADIX).copy()
    cb["node_count"] = len(prims)
    ctx = ptamd.Context(0)
    ctx.upload_primitives(prims); ctx.set_constants(cb)
    ctx.build(); ctx.resize(1162, 1008)
    ctx.primary_hits(ptamd.default_camera(), ptamd.albedo_table(g["mat_table"][:, 0:3], True))
    ctx.set_trace_timing(True)
    for s in range(3):
        ctx.trace_frames(128 * s, ptamd.frame_seeds(1, 128 * s, 128), ptamd.PT_ACCUM_LINEAR)
    ms = ctx.trace_times()
    c = ctx.counters()
    lds_waves = min(16, int((160 * 1024 - 48 * len(prims) - 17 * 64) // (12 * 4 * 64)))
    print(f"n={len(prims)} (LDS stacks for {lds_waves} waves) {({k2: v for k2, v in os.environ.items() if k2.startswith('PTAMD_')})}: "
          f"launch ms {ms[1:].mean():.2f}  Grays/s {c['rays'] / 3 / (ms[1:].mean() * 1e-3) / 1e9:.3f}", flush=True)
    ctx.close()
