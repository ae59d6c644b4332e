"""Short trace calls (the reference's run loop issues one Dispatch per frame): rays per second of scene04 at 1162x1008 for 1, 2,
4 and 8 frames per call, over the number of internal streams the calls are spread over (PTAMD_PIPE_SLOTS) and the share
of the CUs one call's launch takes (PTAMD_SHORT_GRID_DIV).  Every configuration must give the same image (hash printed)."""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT)
import ptamd
scene = sys.argv[1] if len(sys.argv) > 1 else "scene04"
g = np.load(os.path.join(ROOT, "tests", "golden", scene + ".npz"))
FRAMES = 256
seeds = ptamd.frame_seeds(1, 0, FRAMES)


def run(slots, div, overlap=True, grid=0):
    os.environ["PTAMD_PIPE_SLOTS"] = str(slots); os.environ["PTAMD_SHORT_GRID_DIV"] = str(div); os.environ["PTAMD_SHORT_GRID"] = str(grid)
    ctx = ptamd.Context(0)            # knobs are read here
    ctx.set_trace_overlap(overlap)
    ctx.upload_primitives(g["prims"].view(ptamd.PRIM_RENDER)); ctx.set_constants(g["cb"].view(ptamd.CB_RADIX))
    ctx.build(); ctx.resize(1162, 1008)
    ctx.primary_hits(ptamd.default_camera(), ptamd.albedo_table(g["mat_table"][:, 0:3], True))
    out = []
    for per_call in (1, 2, 4, 8):
        for rep in range(2):          # the first repetition warms up (buffers of the slots, code objects)
            ctx.reset_accumulation(); ctx.reset_counters(); ctx.synchronize()
            t0 = time.perf_counter()
            for f in range(0, FRAMES, per_call):
                ctx.trace_frames(f, seeds[f:f + per_call], ptamd.PT_ACCUM_LITERAL)
            ctx.synchronize()
            dt = time.perf_counter() - t0
        rays = ctx.counters()["rays"]
        out.append(f"{per_call}: {rays / dt / 1e9:.2f} Grays/s ({dt / FRAMES * 1e3:.3f} ms/frame)")
    h = hashlib.sha256(ctx.output().tobytes()).hexdigest()[:12]
    ctx.close()
    print(f"overlap {int(overlap)} slots {slots} grid {'1/' + str(div) if not grid else grid}  " + "  ".join(out) + f"  image {h}", flush=True)


if len(sys.argv) > 2 and sys.argv[2] == "grids":
    for grid in (96, 112, 128, 144, 160, 176, 192):
        run(3, 1, grid=grid)
    sys.exit(0)
run(2, 1, overlap=False)
for slots in (2, 3, 4):
    for div in (1, 2, 4):
        run(slots, div)
