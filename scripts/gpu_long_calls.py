"""Back-to-back long trace calls, as bench.py issues them: wall-clock rays per second of a scene over CALLS (environment, default
12) calls of a number of frames each, seed kernel, accumulation and cost sort included; image hash printed.
usage: [PTAMD_LIB=...] python scripts/gpu_long_calls.py [scene W H frames_per_call ...]"""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT)
import ptamd
scene = sys.argv[1] if len(sys.argv) > 1 else "scene04"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1162, 1008)
CALLS = int(os.environ.get("CALLS", "12"))
g = np.load(os.path.join(ROOT, "tests", "golden", scene + ".npz"))
ctx = ptamd.Context(0)
ctx.upload_primitives(g["prims"].view(ptamd.PRIM_RENDER)); ctx.set_constants(g["cb"].view(ptamd.CB_RADIX))
ctx.build(); ctx.resize(W, H)
ctx.primary_hits(ptamd.default_camera(), ptamd.albedo_table(g["mat_table"][:, 0:3], True))
for per in [int(a) for a in sys.argv[4:]] or [128]:
    seeds = ptamd.frame_seeds(1, 0, per * CALLS)
    for rep in range(2):              # the first repetition warms up (buffers, cost order)
        ctx.reset_accumulation(); ctx.reset_counters(); ctx.synchronize()
        t0 = time.perf_counter()
        for c in range(CALLS):
            ctx.trace_frames(c * per, seeds[c * per:(c + 1) * per], ptamd.PT_ACCUM_LINEAR)
        ctx.synchronize()
        dt = time.perf_counter() - t0
    rays = ctx.counters()["rays"]
    h = hashlib.sha256(ctx.output().tobytes()).hexdigest()[:12]
    print(f"{scene} {W}x{H}, {CALLS} calls of {per} frames: {dt / CALLS * 1e3:.2f} ms per call, {rays / dt / 1e9:.3f} Grays/s  image {h}", flush=True)
ctx.close()
