"""scene04 1162x1008: rays per second against frames per launch (1, 2, 4, 8, 32, 128), for the kernel the PTAMD_*
knobs of the environment select (WAVEFRONT=1: the wavefront schedule; TOPO=1: PT_BUILD_TOPOLOGY_MORTON; NEAR=1: nearest child first).  usage: [PTAMD_DUAL=0] python scripts/gpu_launch_lengths.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT)
import ptamd
g = np.load(os.path.join(ROOT, "tests", "golden", "scene04.npz"))
ctx = ptamd.Context(0)
ctx.upload_primitives(g["prims"].view(ptamd.PRIM_RENDER)); ctx.set_constants(g["cb"].view(ptamd.CB_RADIX))
ctx.build(ptamd.PT_BUILD_TOPOLOGY_MORTON if os.environ.get('TOPO') else 0); ctx.resize(1162, 1008)
ctx.primary_hits(ptamd.default_camera(), ptamd.albedo_table(g["mat_table"][:, 0:3], True))
if os.environ.get("NEAR"):
    ctx.set_traversal_order(ptamd.PT_TRAVERSAL_NEAR_FIRST)
if os.environ.get("WAVEFRONT"):
    ctx.set_trace_schedule(ptamd.PT_SCHEDULE_WAVEFRONT)
for frames in [int(a) for a in sys.argv[1:]] or (1, 2, 4, 8, 32, 128):
    launches = max(3, 256 // frames)
    seeds = [ptamd.frame_seeds(1, k * frames, frames) for k in range(launches)]
    ctx.trace_frames(0, seeds[0], ptamd.PT_ACCUM_LINEAR); ctx.synchronize(); ctx.reset_counters()
    t0 = time.perf_counter()
    for k in range(launches):
        ctx.trace_frames(k * frames, seeds[k], ptamd.PT_ACCUM_LINEAR)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(f"topo={os.environ.get('TOPO', '0')} near={os.environ.get('NEAR', '0')} wavefront={os.environ.get('WAVEFRONT', '0')} dual={os.environ.get('PTAMD_DUAL', '1')} frames per launch {frames}: {ctx.counters()['rays'] / dt / 1e9:.2f} Grays/s ({dt / launches * 1e3:.2f} ms per launch)", flush=True)
ctx.close()
