"""Small torch-free target for rocprofv3 counter passes:
  trace [scene W H]  scene04 1162x1008 (or the named scene and size), 3 launches of 128 frames (the benchmark's step)
  soup100m [index|morton] [near]   the benchmark's largest configuration (bench.py --config soup100m: 100M triangles generated
                   on the device, 3840x2160, 3 launches of 2 frames), set up by bench.py's own function
  build [n]        n-triangle soup (default 10M), 3 tree-only builds and 3 default builds
  soup [n] [near] [morton]  n-triangle soup (default 10M) rendered with the options of the largest configuration
                   (full-resolution codes, deep stack, at most 8 rays per path, node records in global memory; near: nearest
                   child first; morton: PT_BUILD_TOPOLOGY_MORTON), 1920x1080, 2 launches of 2 frames"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT)
mode = sys.argv[1]
if mode == "soup100m":
    # the soup is generated on the device by torch: torch first, so that the library binds to the HIP runtime torch brings
    # (as in bench.py); the other modes stay torch-free
    import torch  # noqa: F401
import ptamd
from bench import soup
ctx = ptamd.Context(0)
if mode == "trace":
    scene = sys.argv[2] if len(sys.argv) > 2 else "scene04"
    W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1162, 1008)
    g = np.load(os.path.join(ROOT, "tests", "golden", scene + ".npz"))
    ctx.upload_primitives(g["prims"].view(ptamd.PRIM_RENDER)); ctx.set_constants(g["cb"].view(ptamd.CB_RADIX))
    ctx.build(); ctx.resize(W, H)
    ctx.primary_hits(ptamd.default_camera(), ptamd.albedo_table(g["mat_table"][:, 0:3], True))
    for s in range(3):
        ctx.trace_frames(128 * s, ptamd.frame_seeds(1, 128 * s, 128), ptamd.PT_ACCUM_LINEAR)
    ctx.synchronize()
    print(ctx.counters())
elif mode == "soup100m":
    from bench import setup_soup_scene
    topology = "morton" if "morton" in sys.argv[2:] else "index"
    keep, cb, _ = setup_soup_scene(ctx, 100_000_000, 3840, 2160, topology, "near" in sys.argv[2:])
    ctx.set_trace_timing(True)
    for s in range(3):
        ctx.trace_frames(2 * s, ptamd.frame_seeds(1, 2 * s, 2), ptamd.PT_ACCUM_LINEAR)
    print("trace ms", ctx.trace_times(), ctx.counters())
elif mode == "soup":
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    near = "near" in sys.argv[3:]
    topo = ptamd.PT_BUILD_TOPOLOGY_MORTON if "morton" in sys.argv[3:] else 0
    prims = soup(n)
    prims["vertices"]["pos"] *= np.float32(100.0)
    prims["matid"] = np.where(np.arange(n) % 16 == 0, 4, np.arange(n) % 4)
    cb = np.zeros(1, dtype=ptamd.CB_RADIX)
    cb["node_count"] = n; cb["pos"][0] = (50.0, 50.0, 300.0)
    for k in range(4):
        cb["mats"][0][k]["ns"] = 1; cb["mats"][0][k]["ni"] = 1; cb["mats"][0][k]["kd"][:3] = (0.7, 0.6, 0.5)
    cb["mats"][0][4]["ke"] = 10; cb["mats"][0][4]["ns"] = 1; cb["mats"][0][4]["ni"] = 1; cb["mats"][0][4]["kd"][:3] = (1, 1, 1)
    ctx.upload_primitives(prims); ctx.set_constants(cb); ctx.build(ptamd.PT_BUILD_MORTON_FULL | topo)
    ctx.set_stack_size(32); ctx.set_max_segments(8)
    ctx.set_traversal_order(ptamd.PT_TRAVERSAL_NEAR_FIRST if near else ptamd.PT_TRAVERSAL_REFERENCE)
    ctx.resize(1920, 1080)
    cam = ptamd.default_camera(eye=(50.0, 50.0, 300.0), at=(50.0, 50.0, 50.0)); cam.zfar = 1000.0
    ctx.primary_hits(cam, ptamd.albedo_table([[0.7, 0.6, 0.5]] * 16))
    ctx.set_trace_timing(True)
    for s in range(2):
        ctx.trace_frames(2 * s, ptamd.frame_seeds(1, 2 * s, 2), ptamd.PT_ACCUM_LINEAR)
    print("trace ms", ctx.trace_times(), ctx.counters())
else:
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    ctx.upload_primitives(soup(n))
    for flags in (ptamd.PT_BUILD_TREE_ONLY, ptamd.PT_BUILD_DEFAULT):
        for _ in range(3):
            ctx.build(flags)
        print(flags, ctx.build_times())
ctx.close()
