"""Full comparison of the build with the oracle at the benchmark's size: Morton codes, sorted order and the whole
reference-layout tree (2n nodes) of an n-triangle soup, compared as raw bytes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ptamd
from oracle import orc
from conftest import make_soup
for n in [int(a) for a in sys.argv[1:]] or [10_000_000]:
    prims = make_soup(n, seed=12345)
    ctx = ptamd.Context(0)
    ctx.upload_primitives(prims.view(ptamd.PRIM_RENDER)); ctx.build()
    t = time.time(); ob = orc.build(prims); to = time.time() - t
    ok_codes = np.array_equal(ctx.morton(), ob["rays"]["morton_code"])              # primitive order
    ok_perm = np.array_equal(ctx.permutation(), ob["sorted_rays"]["index"])          # stable sorted order
    tree = ctx.tree()
    ok_tree = tree.tobytes() == ob["tree"][:-1].tobytes()
    print(f"n={n}: codes {'same' if ok_codes else 'DIFFERENT'}, order {'same' if ok_perm else 'DIFFERENT'}, tree (2n x 48 B) "
          f"{'same' if ok_tree else 'DIFFERENT'}; GPU build {ctx.build_times()['total']:.2f} ms, oracle {to:.1f} s", flush=True)
    ctx.close()
    if not (ok_codes and ok_perm and ok_tree): sys.exit(1)
