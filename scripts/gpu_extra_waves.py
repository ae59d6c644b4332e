"""scene04 padded with k tiny triangles (same view, larger tree): time per launch (wall clock over 256 frames) at 1162x1008 against the number of
waves whose stacks fit in LDS beside the tree.  Run under PTAMD_EXTRA_WAVES=0 / 16 and PTAMD_NO_LDS_TREE=1 to compare.
FRAMES=f sets the frames per launch (default 128; 1-3: the single-path kernel).  usage: python scripts/gpu_extra_waves.py [k ...]"""
import os, sys, time
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
    F = int(os.environ.get("FRAMES", "128"))
    launches = max(3, 256 // F)
    seeds = [ptamd.frame_seeds(1, F * s, F) for s in range(launches)]
    ctx.trace_frames(0, seeds[0], ptamd.PT_ACCUM_LINEAR); ctx.synchronize(); ctx.reset_counters()
    t0 = time.perf_counter()                  # wall clock: short calls overlap on two streams, so per-kernel event times would mislead
    for s in range(launches):
        ctx.trace_frames(F * s, seeds[s], ptamd.PT_ACCUM_LINEAR)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    c = ctx.counters()
    lds_waves = min(16, int((160 * 1024 - 48 * len(prims) - 17 * 64) // (12 * 4 * 64)))
    print(f"n={len(prims)} frames/launch {F} (LDS stacks for {lds_waves} waves) {({k2: v for k2, v in os.environ.items() if k2.startswith('PTAMD_')})}: "
          f"ms per launch {dt / launches * 1e3:.2f}  Grays/s {c['rays'] / dt / 1e9:.3f}", flush=True)
    ctx.close()
