"""Render rate of device-generated soups of several sizes with the options of the largest configuration, with and
without the 128-byte child lines (PTAMD_PAIR_MAX_MIB=0 disables them).
usage: gpu_soup_sizes.py n1 n2 ...   (run once per setting of PTAMD_PAIR_MAX_MIB; NEAR=1: nearest child first;
TOPO=1: PT_BUILD_TOPOLOGY_MORTON; FRAMES: frames per launch, default 2)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, ROOT)
import ptamd
from bench import device_soup
import torch
for n in [int(a) for a in sys.argv[1:]]:
    d = device_soup(n, seed=2024, scale=100.0, emitter_every=16)
    ctx = ptamd.Context(0)
    ctx.upload_primitives_device(d.data_ptr(), n)
    cb = np.zeros(1, dtype=ptamd.CB_RADIX)
    cb["node_count"] = n; cb["pos"][0] = (50.0, 50.0, 300.0)
    for k in range(4):
        cb["mats"][0][k]["ns"] = 1; cb["mats"][0][k]["ni"] = 1; cb["mats"][0][k]["kd"][:3] = (0.7, 0.6, 0.5)
    cb["mats"][0][4]["ke"] = 10; cb["mats"][0][4]["ns"] = 1; cb["mats"][0][4]["ni"] = 1; cb["mats"][0][4]["kd"][:3] = (1, 1, 1)
    topo = int(os.environ.get("TOPO", "0"))
    ctx.set_constants(cb); ctx.build(ptamd.PT_BUILD_MORTON_FULL | (ptamd.PT_BUILD_TOPOLOGY_MORTON if topo else 0))
    bt = ctx.build_times()
    ctx.set_stack_size(32); ctx.set_max_segments(8)
    ctx.set_traversal_order(int(os.environ.get("NEAR", "0")))
    ctx.resize(1920, 1080)
    cam = ptamd.default_camera(eye=(50.0, 50.0, 300.0), at=(50.0, 50.0, 50.0)); cam.zfar = 1000.0
    ctx.primary_hits(cam, ptamd.albedo_table([[0.7, 0.6, 0.5]] * 16))
    ctx.set_trace_timing(True)
    ctx.set_detailed_counters(bool(int(os.environ.get("DETAIL", "0"))))
    fr = int(os.environ.get("FRAMES", "2"))           # frames per launch
    for s in range(3):
        ctx.trace_frames(fr * s, ptamd.frame_seeds(1, fr * s, fr), ptamd.PT_ACCUM_LINEAR)
    ms = ctx.trace_times(); c = ctx.counters()
    print(f"n {n} pair_max_mib {os.environ.get('PTAMD_PAIR_MAX_MIB', 'default')} near {os.environ.get('NEAR', '0')} topo {topo}: build {bt['total']:.2f} ms (topology {bt['topology']:.2f}, refit {bt['refit']:.2f}), "
          f"{ms[1:].mean():.1f} ms per launch, {c['rays'] / 3 / (ms[1:].mean() * 1e-3) / 1e6:.1f} Mrays/s, visits/ray {c['node_visits'] / max(c['rays'], 1):.0f}", flush=True)
    ctx.close(); del d
