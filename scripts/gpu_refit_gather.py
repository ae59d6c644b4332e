"""How much of the refit is its random gather of the 80-byte primitives: the 10M-triangle soup built as generated
(sorted order = a random permutation of the input) and built again after the input has been put into sorted order
(then the gather reads the primitives front to back).  The second figure is the floor of ANY scheme that only
reorders or re-aligns the gather; the bytes written are the same in both."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ptamd
from conftest import make_soup
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
prims = make_soup(n, seed=12345, dtype=ptamd.PRIM_RENDER)
ctx = ptamd.Context(0)


def best_of(flags, reps=7):
    best = None
    for _ in range(reps):
        ctx.build(flags)
        t = ctx.build_times()
        if best is None or t["total"] < best["total"]:
            best = t
    return {k: round(v, 4) for k, v in best.items() if isinstance(v, float)}


ctx.upload_primitives(prims)
print("as generated      default  ", best_of(0), flush=True)
print("as generated      tree only", best_of(ptamd.PT_BUILD_TREE_ONLY), flush=True)
perm = ctx.permutation().copy()
ctx.upload_primitives(np.ascontiguousarray(prims[perm]))
print("input pre-sorted  default  ", best_of(0), flush=True)
print("input pre-sorted  tree only", best_of(ptamd.PT_BUILD_TREE_ONLY), flush=True)
assert np.array_equal(ctx.permutation(), np.arange(n)), "a sorted input must sort to the identity"
ctx.close()
