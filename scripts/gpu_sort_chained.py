"""Eight-pass sort of full-resolution codes (PT_BUILD_MORTON_FULL): single-read passes (chained scan, the default) against
the two-read passes (PTAMD_SORT_CHAINED=0): permutations must be identical (up to 20 M keys also to numpy's stable argsort of the codes); stage times of both (best of five).
usage: python scripts/gpu_sort_chained.py [n ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ptamd
from conftest import make_soup

sizes = [int(a) for a in sys.argv[1:]] or [1, 2, 4095, 4096, 4097, 100_003, 1_000_000, 10_000_000]
flags = ptamd.PT_BUILD_MORTON_FULL | ptamd.PT_BUILD_TREE_ONLY
for n in sizes:
    prims = make_soup(n, seed=777 + n, dtype=ptamd.PRIM_RENDER)
    res = {}
    for chained in (1, 0):
        os.environ["PTAMD_SORT_CHAINED"] = str(chained)          # read once, by pt_create
        ctx = ptamd.Context(0)
        ctx.upload_primitives(prims)
        best = None
        for _ in range(5):
            ctx.build(flags)
            t = ctx.build_times()
            if best is None or t["sort"] < best["sort"]: best = t
        res[chained] = (ctx.permutation().copy(), best)
        codes = ctx.morton()
        ctx.close()
    same = np.array_equal(res[1][0], res[0][0])
    if n <= 20_000_000:          # and both are numpy's stable order of the codes
        same = same and np.array_equal(res[1][0], np.argsort(codes, kind="stable").astype(np.int32))
    print(f"n={n}: permutations identical: {same}; sort ms chained {res[1][1]['sort']:.4f} / two-read {res[0][1]['sort']:.4f}; "
          f"build ms {res[1][1]['total']:.4f} / {res[0][1]['total']:.4f}", flush=True)
    if not same:
        sys.exit(1)
