"""Exploratory GPU run: timings of trace on scene04 and of the build on a random soup."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT)
import ptamd

def soup(n, seed=12345):
    rng = np.random.default_rng(seed)
    c = rng.random((n, 3), dtype=np.float32)
    e = np.float32(0.5 * n ** (-1.0 / 3.0))
    prims = np.zeros(n, dtype=ptamd.PRIM_RENDER)
    prims["vertices"]["pos"][:, 0] = c
    prims["vertices"]["pos"][:, 1] = c + (rng.random((n, 3), dtype=np.float32) * 2 - 1) * e
    prims["vertices"]["pos"][:, 2] = c + (rng.random((n, 3), dtype=np.float32) * 2 - 1) * e
    prims["vertices"]["normal"][:, :, 1] = 1.0
    prims["pad"] = -1
    return prims

scene = sys.argv[1] if len(sys.argv) > 1 else "scene04"
g = np.load(os.path.join(ROOT, "tests", "golden", scene + ".npz"))
prims = g["prims"].view(ptamd.PRIM_RENDER); cb = g["cb"].view(ptamd.CB_RADIX)
ctx = ptamd.Context(0)
ctx.upload_primitives(prims); ctx.set_constants(cb)
for _ in range(3):
    ctx.build()
print(scene, "build", ctx.build_times(), flush=True)
W, H = 1162, 1008
ctx.resize(W, H)
ctx.primary_hits(ptamd.default_camera(), ptamd.albedo_table(g["mat_table"][:, 0:3]))
for mode, name in ((ptamd.PT_ACCUM_LINEAR, "linear"), (ptamd.PT_ACCUM_LITERAL, "literal")):
    for nf in (1, 8, 32):
        seeds = ptamd.frame_seeds(1, 0, nf)
        ctx.reset_accumulation(); ctx.reset_counters(); ctx.synchronize()
        t = time.time(); ctx.trace_frames(0, seeds, mode); ctx.synchronize(); dt = time.time() - t
        c = ctx.counters()
        print(f"{name} frames={nf} {dt*1e3:.2f} ms  {c['rays']/dt/1e6:.1f} Mrays/s  {c['paths']/dt/1e6:.1f} Mpaths/s rays/path={c['rays']/c['paths']:.2f}", flush=True)
ctx.set_detailed_counters(True)
ctx.reset_accumulation(); ctx.reset_counters()
ctx.trace_frames(0, ptamd.frame_seeds(1, 0, 4), ptamd.PT_ACCUM_LINEAR); print("detailed", ctx.counters(), flush=True)
ctx.close()

for n in (1_000_000, 10_000_000):
    p = soup(n)
    ctx = ptamd.Context(0)
    ctx.upload_primitives(p)
    for _ in range(3):
        ctx.build()
    bt = ctx.build_times()
    print(f"soup n={n} build {bt}  {n/bt['total']/1e3:.1f} Mtris/s", flush=True)
    m = ctx.morton(); perm = ctx.permutation()
    ks = m[perm]
    print("  sorted:", bool(np.all(ks[:-1] <= ks[1:])), "perm ok:", bool(np.array_equal(np.sort(perm), np.arange(n))),
          "stable:", bool(np.all((ks[:-1] < ks[1:]) | (perm[:-1] < perm[1:]))), flush=True)
    ctx.close()
