"""Stage times of the 10M-triangle soup build (best of five), and the time of the on-demand tree export."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ptamd
from conftest import make_soup
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
prims = make_soup(n, seed=12345, dtype=ptamd.PRIM_RENDER)
ctx = ptamd.Context(0)
ctx.upload_primitives(prims)
for flags in (0, ptamd.PT_BUILD_TREE_ONLY, 0, ptamd.PT_BUILD_TREE_ONLY):
    best = None
    for _ in range(5):
        ctx.build(flags)
        t = ctx.build_times()
        if best is None or t["total"] < best["total"]: best = t
    print(flags, {k: round(v, 4) if isinstance(v, float) else v for k, v in best.items()}, flush=True)
import time
ctx.synchronize(); t = time.time(); tree = ctx.tree(); print("pt_download_tree (export + 2n x 48 B copy to host)", round((time.time() - t) * 1e3, 1), "ms")
ctx.close()
