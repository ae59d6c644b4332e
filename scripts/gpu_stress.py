"""Longer bit-exactness run than the test-suite affords: every scene at the benchmark size, FRAMES frames
from a late frame index in one launch (literal accumulation) against the oracle frame by frame; then random
soups with every material kind.  Prints one line per case; exits non-zero on the first difference."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ptamd
from oracle import orc
from conftest import make_soup, soup_cb, load_golden
FRAMES = int(sys.argv[1]) if len(sys.argv) > 1 else 24
W, H = 1162, 1008
bad = 0

def check(label, prims, cb, tree, alb, cam_g, cam_o, W, H, first, frames, seed_base):
    global bad
    ctx = ptamd.Context(0)
    ctx.upload_primitives(prims); ctx.set_constants(cb); ctx.build(); ctx.resize(W, H)
    ctx.primary_hits(cam_g, alb)
    sc = orc.OracleScene(prims.view(orc.PRIM_RENDER), tree if tree is not None else orc.build(prims.view(orc.PRIM_RENDER))["tree"], cb.view(orc.CB_RADIX))
    gb = orc.primary(sc, cam_o, W, H, alb)
    same_gb = all(np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8)) for a, b in zip(ctx.download_gbuffer(), gb))
    seeds = ptamd.frame_seeds(seed_base, first, frames)
    ctx.set_detailed_counters(True); ctx.reset_counters()
    t = time.time()
    ctx.trace_frames(first, seeds, ptamd.PT_ACCUM_LITERAL)
    out = ctx.output(); tg = time.time() - t
    # the recurrence starts from an empty image at frame `first`: old * first/(first+1) with old = 0
    old = np.zeros((H, W, 4), np.float32)
    tot = dict(paths=0, rays=0, node_visits=0, box_tests=0, tri_tests=0)
    t = time.time()
    for k in range(frames):
        old, _, _, c = orc.trace_frame(sc, gb, first + k, seeds[k], old)
        tot = {a: tot[a] + c[a] for a in tot}
    to = time.time() - t
    same = np.array_equal(out.view(np.uint32), old.view(np.uint32)) and ctx.counters() == tot
    print(f"{label}: {W}x{H} frames {first}..{first + frames - 1}: gbuffer {'same' if same_gb else 'DIFFERENT'}, image+counters "
          f"{'same' if same else 'DIFFERENT'} ({tot['rays']} rays; gpu {tg:.2f} s, oracle {to:.1f} s)", flush=True)
    ctx.close()
    if not (same and same_gb): bad += 1

for scene in ("scene01", "scene02", "scene03", "scene04"):
    g = load_golden(scene)
    check(scene, g["prims"].view(ptamd.PRIM_RENDER), g["cb"].view(ptamd.CB_RADIX), g["tree"], ptamd.albedo_table(g["mat_table"][:, 0:3], True),
          ptamd.default_camera(), orc.default_camera(), W, H, 1000, FRAMES, 7)
rng = np.random.default_rng(99)
for i in range(12):
    n = int(rng.integers(1, 6000))
    prims = make_soup(n, seed=500 + i, dtype=ptamd.PRIM_RENDER)
    prims["matid"] = rng.integers(0, 4, n)
    cb = soup_cb(n)
    kd = [[0.7, 0.6, 0.5], [0.1, 0.1, 0.1], [1, 1, 1], [1, 1, 1]] + [[0, 0, 0]] * 12
    eye = (0.5, 0.5, 3.0)
    check(f"soup n={n}", prims, cb, None, ptamd.albedo_table(kd, True), ptamd.default_camera(eye=eye, at=(0.5, 0.5, 0.5)),
          orc.default_camera(eye=eye, at=(0.5, 0.5, 0.5)), 320, 240, int(rng.integers(0, 5000)), 6, int(rng.integers(1, 1 << 30)))
print("FAILED" if bad else "all cases bit-identical")
sys.exit(1 if bad else 0)
