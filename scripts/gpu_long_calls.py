"""Back-to-back long trace calls, as bench.py issues them: wall-clock rays per second of scene04 at 1162x1008 over 12 calls of
128 frames, with the calls serialized on the context's stream and overlapped on the library's internal streams
(PTAMD_OVERLAP_MAX_FRAMES).  Every configuration must give the same image (hash printed).
usage: [PTAMD_LIB=...] python scripts/gpu_long_calls.py [scene W H frames_per_call]"""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT)
import ptamd
scene = sys.argv[1] if len(sys.argv) > 1 else "scene04"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1162, 1008)
PER = int(sys.argv[4]) if len(sys.argv) > 4 else 128
CALLS = int(os.environ.get("CALLS", "12"))
g = np.load(os.path.join(ROOT, "tests", "golden", scene + ".npz"))
seeds = ptamd.frame_seeds(1, 0, PER * CALLS)


def run(max_frames, slots):
    os.environ["PTAMD_OVERLAP_MAX_FRAMES"] = str(max_frames); os.environ["PTAMD_PIPE_SLOTS"] = str(slots)
    ctx = ptamd.Context(0)
    ctx.upload_primitives(g["prims"].view(ptamd.PRIM_RENDER)); ctx.set_constants(g["cb"].view(ptamd.CB_RADIX))
    ctx.build(); ctx.resize(W, H)
    ctx.primary_hits(ptamd.default_camera(), ptamd.albedo_table(g["mat_table"][:, 0:3], True))
    for rep in range(2):
        ctx.reset_accumulation(); ctx.reset_counters(); ctx.synchronize()
        t0 = time.perf_counter()
        for c in range(CALLS):
            ctx.trace_frames(c * PER, seeds[c * PER:(c + 1) * PER], ptamd.PT_ACCUM_LINEAR)
        ctx.synchronize()
        dt = time.perf_counter() - t0
    rays = ctx.counters()["rays"]
    h = hashlib.sha256(ctx.output().tobytes()).hexdigest()[:12]
    ctx.close()
    print(f"overlap up to {max_frames} frames, {slots} streams: {dt / CALLS * 1e3:.2f} ms per call, {rays / dt / 1e9:.3f} Grays/s  image {h}", flush=True)


run(8, 3)
run(4096, 2)
run(4096, 3)
run(8, 3)
