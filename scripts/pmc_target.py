"""Small torch-free target for rocprofv3 counter passes: `trace` (scene04 1162x1008, 3 launches of
128 frames, the benchmark's step) or `build` (10M-triangle soup, 3 tree-only builds and 3 full builds)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT)
import ptamd
mode = sys.argv[1]
ctx = ptamd.Context(0)
if mode == "trace":
    g = np.load(os.path.join(ROOT, "tests", "golden", "scene04.npz"))
    ctx.upload_primitives(g["prims"].view(ptamd.PRIM_RENDER)); ctx.set_constants(g["cb"].view(ptamd.CB_RADIX))
    ctx.build(); ctx.resize(1162, 1008)
    ctx.primary_hits(ptamd.default_camera(), ptamd.albedo_table(g["mat_table"][:, 0:3], True))
    for s in range(3):
        ctx.trace_frames(128 * s, ptamd.frame_seeds(1, 128 * s, 128), ptamd.PT_ACCUM_LINEAR)
    ctx.synchronize()
    print(ctx.counters())
else:
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    rng = np.random.default_rng(12345)
    c = rng.random((n, 3), dtype=np.float32); e = np.float32(0.5 * n ** (-1.0 / 3.0))
    prims = np.zeros(n, dtype=ptamd.PRIM_RENDER)
    prims["vertices"]["pos"][:, 0] = c
    prims["vertices"]["pos"][:, 1] = c + (rng.random((n, 3), dtype=np.float32) * 2 - 1) * e
    prims["vertices"]["pos"][:, 2] = c + (rng.random((n, 3), dtype=np.float32) * 2 - 1) * e
    prims["pad"] = -1
    ctx.upload_primitives(prims)
    for flags in (ptamd.PT_BUILD_TREE_ONLY, ptamd.PT_BUILD_DEFAULT):
        for _ in range(3):
            ctx.build(flags)
        print(flags, ctx.build_times())
ctx.close()
