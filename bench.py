#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--config scene04|soup100m]

--config scene04 (default; the configuration the metric is quoted on): scene04 at 1162x1008, progressive
accumulation with the fixed-seed convention.  One STEP = one block of 128 consecutive frames (four cycles of
the shader's frame-index stratification) = 128 paths per pixel, traced by ONE launch of the path-trace kernel
per GPU.  The default K = 8 steps is the 1024 spp of that configuration.  With N GPUs every rank holds the
scene and its tree, traces its own blocks (weak scaling), and the float accumulation sums are merged by one
RCCL all-reduce (pt_reduce_linear_sum, through the C ABI) + resolve at the end, inside the timed region.
Also measured, outside the timed region: the LBVH build on 10M- and 50M-triangle random soups (per GPU,
replicas), and a bounded CPU baseline (the oracle, rank 0, N = 1 only).

--config scene02_4k / scene03_4k: scene02 / scene03 at 3840x2160; the named configuration is 4096 spp split over 8 GPUs =
512 spp = 4 steps per GPU (--steps 4).  Same kernels and roofline as scene04; no build or CPU legs are skipped.

--config soup100m: the largest named configuration -- a 100M-triangle random soup (generated on the device) at
3840x2160, at most 8 rays per path, full-resolution codes, tree over the Morton codes (--topology index: the
reference's index-keyed tree), nearest child first (--near-first 0: the reference's order), deep stack
(NON-PARITY options, see include/ptamd.h).
One STEP = one launch of 2 frames.  Node records are read from global memory: this one is bound by the rate at which the
memory system serves dependent random requests.

`python bench.py --gpus N` with N > 1 and no launcher (WORLD_SIZE unset) starts its own N ranks through
torch.distributed.run before anything touches a GPU; under a launcher whose WORLD_SIZE differs from --gpus it refuses.

value = BVH traversals (`intersect()` calls, counted by the kernel) of all ranks / wall time.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, data sheet
SIMDS = 256 * 4                # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4                # maximum shader clock
# a SIMD issues one wave64 vector instruction per 2 clocks: chip-wide peak in wave-instructions per second
VALU_PEAK_GINST = SIMDS * CLOCK_GHZ / 2.0
LDS_PEAK_GBS = 256 * 256 * CLOCK_GHZ          # 256 B per clock per CU
# algorithmic bytes of the traversal model: per node visit 24 B box + 8 B children + 4 B index,
# per triangle test three positions = 36 B
BYTES_PER_VISIT = 36
BYTES_PER_TRI_TEST = 36
# build, algorithmic bytes per triangle with 15-bit codes (two 8-bit sort passes):
# boxes+codes 72, scene box 24, sort 8 + 2*24, gather 64, links 16, refit 144 = 376 for the tree;
# the default build also leaves the two 48-byte per-leaf records of the trace kernels: + 96
BUILD_BYTES_TREE = 376
BUILD_BYTES_DEFAULT = 376 + 96
# bytes each stage's kernels request per triangle (what this implementation reads + writes, by design)
STAGE_BYTES = {
    # (sort: the first pass's digits are counted by the Morton kernel: scatter 4 r + 8 w; second pass count 4 r, scatter 8 r + 4 w)
    "tree_only": {"bounds": 80 + 32, "morton": 32 + 4, "sort": (4 + 8) + (4 + 8 + 4), "refit": 4 + 32 + 64},
    "default": {"bounds": 80 + 12, "morton": 12 + 4, "sort": (4 + 8) + (4 + 8 + 4), "refit": 4 + 80 + 64 + 96},
}
PROFILE_BUILD = os.path.join(ROOT, "profiles", "r03_build_pmc.json")
# counter profiles of the trace kernel, one per workload the benchmark can run (scripts/pmc_target.py takes them on
# the same scene, image size and launch length): key = (scene, width, height, frames per launch)
PROFILE_TRACE = {
    ("scene04", 1162, 1008, 128): os.path.join(ROOT, "profiles", "r03_trace_pmc.json"),
    ("scene02", 3840, 2160, 128): os.path.join(ROOT, "profiles", "r03_trace_scene02_4k_pmc.json"),
    ("scene03", 3840, 2160, 128): os.path.join(ROOT, "profiles", "r03_trace_scene03_4k_pmc.json"),
}
# ... and of the 100M-triangle soup at 3840x2160, 2-frame launches: key = (topology, nearest child first)
PROFILE_SOUP = {
    ("index", 0): os.path.join(ROOT, "profiles", "r03_soup100m_pmc.json"),
    ("index", 1): os.path.join(ROOT, "profiles", "r03_soup100m_near_first_pmc.json"),
    ("morton", 1): os.path.join(ROOT, "profiles", "r03_soup100m_morton_pmc.json"),
}
# Dependent random record reads per second that the memory system serves (scripts/micro/random_records.hip,
# profiles/r03_random_records.txt): the roof of traversals whose node array is far larger than the caches.
# (array MiB, G records/s); between the points the rate is interpolated in log(size)
RANDOM_RECORD_RATE = ((160, 54.0), (1280, 54.7), (6400, 23.7), (12800, 21.5))


def random_record_peak(array_bytes):
    import math
    mib = array_bytes / 2.0 ** 20
    pts = RANDOM_RECORD_RATE
    if mib <= pts[0][0]:
        return pts[0][1]
    for (a, ra), (b, rb) in zip(pts, pts[1:]):
        if mib <= b:
            t = (math.log(mib) - math.log(a)) / (math.log(b) - math.log(a))
            return ra + t * (rb - ra)
    return pts[-1][1]
KERNEL_SOURCES = ("path-tracer_amd/csrc/trace_kernels.hip", "path-tracer_amd/csrc/build_kernels.hip",
                  "path-tracer_amd/csrc/api.hip", "path-tracer_amd/csrc/pt_internal.hpp", "include/ptamd_math.h")


def sources_hash():
    """Identifies the kernel sources a counter profile was taken with (the GPU box has no git history)."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def load_profile(path):
    """Counter summary written by scripts/pmc_summarize.py, or None; `current` says whether it was taken with
    the kernel sources of this tree."""
    try:
        p = json.load(open(path))
    except Exception:
        return None
    p["_current"] = p.get("_meta", {}).get("sources_sha256_16") == sources_hash()
    return p


def soup(n, seed=12345):
    """Random triangle soup in the reference's PrimitiveRender layout."""
    import ptamd
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


def device_soup(n, seed=12345, scale=1.0, emitter_every=0):
    """The same kind of soup generated on the GPU (no host copy), (n, 20) float32 = PrimitiveRender records;
    materials cycle 0..3, every `emitter_every`-th triangle gets material 4."""
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    e = 0.5 * n ** (-1.0 / 3.0)
    out = torch.zeros((n, 20), dtype=torch.float32, device="cuda")
    c = torch.rand((n, 3), generator=g, device="cuda")
    out[:, 0:3] = c * scale
    out[:, 6:9] = (c + (torch.rand((n, 3), generator=g, device="cuda") * 2 - 1) * e) * scale
    out[:, 12:15] = (c + (torch.rand((n, 3), generator=g, device="cuda") * 2 - 1) * e) * scale
    del c
    for k in (3, 9, 15):
        out[:, k:k + 3] = torch.rand((n, 3), generator=g, device="cuda") * 2 - 1
    ids = out.view(torch.int32)
    i = torch.arange(n, device="cuda", dtype=torch.int32)
    ids[:, 18] = torch.where(i % emitter_every == 0, torch.full_like(i, 4), i % 4) if emitter_every else i % 4
    ids[:, 19] = -1
    torch.cuda.synchronize()
    return out


def setup_soup_scene(ctx, n, W, H, topology, near_first):
    """The largest named configuration on `ctx`: an n-triangle soup generated on the device (every rank: the same
    scene), 4 diffuse materials + 1 emitter, full-resolution codes, deep stack, at most 8 rays per path; built twice
    (the second build's times are returned), image resized, G-buffer cast.  Shared with scripts/pmc_target.py so
    that counter profiles are taken on exactly the benchmark's workload."""
    import ptamd
    keep = device_soup(n, seed=2024, scale=100.0, emitter_every=16)
    ctx.upload_primitives_device(keep.data_ptr(), n)
    cb = np.zeros(1, dtype=ptamd.CB_RADIX)
    cb["node_count"] = n
    cb["pos"][0] = (50.0, 50.0, 300.0)
    for k, kd in enumerate(((0.7, 0.6, 0.5), (0.3, 0.6, 0.7), (0.6, 0.7, 0.3), (0.75, 0.75, 0.75))):
        cb["mats"][0][k]["ns"] = 1; cb["mats"][0][k]["ni"] = 1; cb["mats"][0][k]["kd"][:3] = kd
    cb["mats"][0][4]["ke"] = 10; cb["mats"][0][4]["ns"] = 1; cb["mats"][0][4]["ni"] = 1; cb["mats"][0][4]["kd"][:3] = (1, 1, 1)
    ctx.set_constants(cb)
    soup_flags = ptamd.PT_BUILD_MORTON_FULL | (ptamd.PT_BUILD_TOPOLOGY_MORTON if topology == "morton" else 0)
    ctx.build(soup_flags)
    ctx.build(soup_flags)
    scene_build = ctx.build_times()
    ctx.resize(W, H)
    ctx.set_stack_size(32)
    ctx.set_max_segments(8)
    ctx.set_traversal_order(ptamd.PT_TRAVERSAL_NEAR_FIRST if near_first else ptamd.PT_TRAVERSAL_REFERENCE)
    cam = ptamd.default_camera(eye=(50.0, 50.0, 300.0), at=(50.0, 50.0, 50.0))
    cam.zfar = 1000.0
    ctx.primary_hits(cam, ptamd.albedo_table([list(cb["mats"][0][m]["kd"][:3]) for m in range(16)]))
    return keep, cb, scene_build


class _DevArray:
    """Zero-copy view of library-owned device memory for torch (`__cuda_array_interface__`)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (ptr, False), "version": 2}


def hbm_copy_gbs(nbytes=1 << 30, reps=10):
    """Device-to-device copy rate (read + write bytes per second): what a plain streaming kernel reaches
    on this GPU, beside the 8 TB/s of the data sheet."""
    import torch
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    b = torch.empty_like(a)
    b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    del a, b
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def bench_build(ctx_factory, n, reps, on_device, copy_gbs):
    """LBVH build of an n-triangle soup, HIP events inside the library, median of `reps`.  Headline = the
    default build (the one pt_trace* can use); the tree-only build (PT_BUILD_TREE_ONLY) is the secondary figure."""
    import ptamd
    ctx = ctx_factory()
    if on_device:
        d = device_soup(n)
        ctx.upload_primitives_device(d.data_ptr(), n)
    else:
        ctx.upload_primitives(soup(n))

    def median_times(flags):
        times = []
        for _ in range(reps + 1):
            ctx.build(flags)
            times.append(ctx.build_times())
        times = times[1:]
        return {k: sorted(t[k] for t in times)[len(times) // 2] for k in list(ptamd.STAGES) + ["total"]}

    tree = median_times(ptamd.PT_BUILD_TREE_ONLY)
    full = median_times(ptamd.PT_BUILD_DEFAULT)
    ctx.close()
    tot_tree, tot = tree.pop("total"), full.pop("total")
    gbs = n * BUILD_BYTES_DEFAULT / (tot * 1e-3) / 1e9
    gbs_tree = n * BUILD_BYTES_TREE / (tot_tree * 1e-3) / 1e9

    def stage_gbs(times, model):
        return {k: n * b / (times[k] * 1e-3) / 1e9 for k, b in model.items() if times.get(k, 0) > 0}

    prof = load_profile(PROFILE_BUILD) if n == 10_000_000 else None
    traffic = None
    if prof and prof["_current"]:
        per_build = {"k_bounds": 1, "k_bbox_reduce": 1, "k_morton_hist": 1, "k_sort_hist": 1, "k_sort_scan": 2,
                     "k_sort_scatter": 2, "k_refit_low<pack>": 1, "k_refit_upper": 2}
        try:
            traffic = sum(prof[k]["hbm_bytes_per_launch"] * c for k, c in per_build.items())
        except Exception:
            traffic = None
    return {
        "workload": f"{n} random triangles (soup{', generated on the device' if on_device else ''}), 15-bit codes, 2 radix passes; "
                    f"default build = tree + the trace kernels' per-leaf records",
        "mtris_per_s": n / (tot * 1e-3) / 1e6,
        "ms": tot,
        "stage_ms": full,
        "stage_GBps_requested": stage_gbs(full, STAGE_BYTES["default"]),
        "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                     "bytes_per_tri": BUILD_BYTES_DEFAULT, "traffic": traffic,
                     "traffic_note": None if traffic else "no counter profile taken with these kernel sources",
                     "measured_copy_GBps": copy_gbs, "frac_of_measured_copy": gbs / copy_gbs},
        "tree_only": {"mtris_per_s": n / (tot_tree * 1e-3) / 1e6, "ms": tot_tree, "stage_ms": tree,
                      "stage_GBps_requested": stage_gbs(tree, STAGE_BYTES["tree_only"]),
                      "roofline": {"bound": "hbm", "achieved": gbs_tree, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": gbs_tree / HBM_PEAK_GBS, "bytes_per_tri": BUILD_BYTES_TREE,
                                   "frac_of_measured_copy": gbs_tree / copy_gbs},
                      "note": "PT_BUILD_TREE_ONLY: tree and sorted order; pt_trace* refuse until a default build"},
    }


def cpu_model():
    """Model name of the host CPU as /proc/cpuinfo gives it, with family / model / stepping: the GPU boxes are virtual
    machines whose model name carries no model number ("AMD EPYC"), the numbers still tell the generation."""
    info = {}
    try:
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k = k.strip()
            if k in ("model name", "cpu family", "model", "stepping") and k not in info:
                info[k] = v.strip()
            if len(info) == 4:
                break
    except Exception:
        pass
    if "model name" not in info:
        return "unknown"
    ids = ", ".join(f"{k} {info[k]}" for k in ("cpu family", "model", "stepping") if k in info)
    return info["model name"] + (f" ({ids})" if ids else "")


def cpu_baseline(g, W, H, budget_s=10.0, max_frames=32):
    """The oracle (CPU restatement) on this host's cores, on whole frames of the same workload: all threads
    within a time budget, then one frame on a single thread."""
    from oracle import orc
    threads = orc.n_threads() if "ORC_THREADS" in os.environ else max(1, min(16, os.cpu_count() or 1))
    prims = g["prims"].view(orc.PRIM_RENDER)
    cb = g["cb"].view(orc.CB_RADIX)
    t0 = time.time()
    ob = orc.build(prims)
    build_s = time.time() - t0
    sc = orc.OracleScene(prims, ob["tree"], cb)
    kd = [list(m[0:3]) for m in g["mat_table"]] + [[0, 0, 0]] * (16 - len(g["mat_table"]))
    gb = orc.primary(sc, orc.default_camera(), W, H, orc.albedo_table(cb, True, kd_override=kd), threads=threads)
    acc = np.zeros((H, W, 4), np.float32)
    seeds = orc.frame_seeds(1, 0, max_frames)
    rays, frames = 0, 0
    t0 = time.time()
    while frames < max_frames and (time.time() - t0) < budget_s:
        rays += orc.trace_frame_linear(sc, gb, frames, seeds[frames], acc, threads=threads)["rays"]
        frames += 1
    dt = time.time() - t0
    # one thread: the top quarter of frame 0 (rows 0 .. H/4), about 1 M rays
    sub = tuple(a[:H // 4] for a in gb)
    acc1 = np.zeros((H // 4, W, 4), np.float32)
    t1 = time.time()
    rays1 = orc.trace_frame_linear(sc, sub, 0, seeds[0], acc1, threads=1)["rays"]
    dt1 = time.time() - t1
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "cpu": cpu_model(), "host_cpus": os.cpu_count(),
            "single_thread_mrays_per_s": rays1 / dt1 / 1e6,
            "sample": f"scene04 {W}x{H}, frames 0..{frames - 1} ({rays} rays in {dt:.1f} s on {threads} threads); one thread: "
                      f"rows 0..{H // 4 - 1} of frame 0 ({rays1} rays in {dt1:.1f} s); "
                      f"CPU LBVH build of its {len(prims)} triangles {build_s * 1e3:.1f} ms single-threaded"}


def trace_roofline_scene(kern_avg_ms, rays_per_launch, bytes_per_ray, launches_per_step, profile_path):
    """Roofline of the trace kernel with the tree in LDS.  The kernel is bound by vector-instruction issue:
    `achieved` = vector wave-instructions per launch (hardware counter SQ_INSTS_VALU of the committed profile,
    taken on this same launch shape) over the launch duration measured HERE with HIP events; `peak` = one wave64
    instruction per 2 clocks per SIMD.  The algorithmic traversal bytes are served by LDS and L1/L2 and reported
    as such, not as an HBM fraction."""
    prof = load_profile(profile_path) if profile_path else None
    # the kernel the library runs for trees in LDS: two paths per lane (k_trace_dual) unless PTAMD_DUAL=0
    kernel = "k_trace_persistent" if os.environ.get("PTAMD_DUAL") == "0" else "k_trace_dual"
    k = (prof or {}).get(kernel)
    on_chip = rays_per_launch * bytes_per_ray / (kern_avg_ms * 1e-3) / 1e9
    out = {"bound": "valu", "unit": "G wave-instructions/s", "peak": VALU_PEAK_GINST, "kernel": kernel,
           "launch_ms_avg": kern_avg_ms, "launches_per_step": launches_per_step,
           "algorithmic_bytes_served_on_chip_GBps": on_chip,
           "peak_note": f"{SIMDS} SIMDs x {CLOCK_GHZ} GHz / 2 clocks per wave64 instruction"}
    if not k or "SQ_INSTS_VALU" not in k:
        out.update({"achieved": None, "frac": None, "traffic": None,
                    "note": "no counter profile of THIS workload (scene, image size, frames per launch) under profiles/; "
                            "same kernel, see the default configuration's line for its roofline"})
        return out
    valu = k["SQ_INSTS_VALU"]
    achieved = valu / (kern_avg_ms * 1e-3) / 1e9
    lanes = k["SQ_THREAD_CYCLES_VALU"] / valu if "SQ_THREAD_CYCLES_VALU" in k else None
    out.update({
        "achieved": achieved, "frac": achieved / VALU_PEAK_GINST,
        "valu_instructions_per_launch": valu,
        "active_lanes_per_instruction": lanes,
        "useful_lane_frac": (achieved / VALU_PEAK_GINST) * lanes / 64.0 if lanes else None,
        "traffic": k.get("hbm_bytes_per_launch") if prof["_current"] else None,
        "hbm": {"bytes_per_launch": k.get("hbm_bytes_per_launch"),
                "GBps": (k.get("hbm_bytes_per_launch", 0) / (kern_avg_ms * 1e-3) / 1e9) if k.get("hbm_bytes_per_launch") else None,
                "peak": HBM_PEAK_GBS},
        "profile": {"file": os.path.relpath(profile_path, ROOT), "taken_with_these_kernel_sources": prof["_current"],
                    "meta": prof.get("_meta")},
    })
    if "SQ_INSTS_LDS" in k:
        # every node visit is three 16-byte LDS reads per lane; conflict cycles are extra LDS-array cycles
        lds_cycles = 4.0 * k["SQ_INSTS_LDS"] + k.get("SQ_LDS_BANK_CONFLICT", 0.0)
        clocks = k.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        out["lds"] = {"instructions_per_launch": k["SQ_INSTS_LDS"], "bank_conflict_cycles": k.get("SQ_LDS_BANK_CONFLICT"),
                      "array_busy_frac": (lds_cycles / 256.0 / clocks) if clocks else None,
                      "note": "4 LDS-array cycles per ds_read_b128 wave-instruction + conflict cycles, per CU, over the launch's clocks"}
    if not prof["_current"]:
        out["note"] = "counter profile was taken with other kernel sources: instruction counts may be stale, traffic withheld"
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` started as ONE process: start the N ranks (one per GPU) as a child
    `torch.distributed.run` job with this same command line, pass rank 0's JSON line through, and return the
    job's exit code.  The parent makes no GPU call (torch and the library are not even imported yet)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PTAMD_BENCH_LAUNCHED_BY=str(os.getpid()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench.py: --gpus {n} without WORLD_SIZE: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    lines = 0
    for line in child.stdout:
        if line.startswith("{"):
            lines += 1
        sys.stdout.write(line)
        sys.stdout.flush()
    rc = child.wait()
    if rc == 0 and lines != 1:
        print(f"bench.py: the {n}-rank job printed {lines} JSON lines, expected exactly one", file=sys.stderr)
        return 3
    return rc


def trace_roofline_soup(kern_avg_ms, rays_per_launch, visits_per_ray, bytes_per_ray, launches_per_step, n, profile_path):
    """Roofline of the trace kernel on a tree far larger than the caches.  Every node visit below the cached top of the tree
    is a dependent read of one record at a random address, and the memory system serves a bounded number of such
    requests per second whatever their size up to a 128-byte line (micro-benchmark: RANDOM_RECORD_RATE).  `achieved` =
    memory-side read requests per launch (hardware counter TCC_EA0_RDREQ of the committed profile of this workload)
    over the launch duration measured HERE with HIP events; `peak` = the highest rate the micro-benchmark reaches (arrays
    up to 1.3 GiB: 54.7 G requests/s, which in 128-byte lines is 7.0 TB/s, i.e. the memory bandwidth that is achievable at
    all).  Uniformly random reads over an array of THIS tree's size are slower than that (address translation; reported as
    `uniform_random_rate_at_tree_size`); a traversal beats that figure because consecutive visits stay within few pages.
    `traffic` = HBM bytes per launch from the same profile (128-byte lines).  The algorithmic figure (36 B per node visit +
    36 B per triangle test) is kept, labelled as such."""
    prof = load_profile(profile_path) if profile_path else None
    k = (prof or {}).get("k_trace_persistent")
    array_bytes = n * 128
    peak = max(r for _, r in RANDOM_RECORD_RATE)
    algorithmic = rays_per_launch * bytes_per_ray / (kern_avg_ms * 1e-3) / 1e9
    out = {"bound": "memory requests", "unit": "G requests/s", "peak": peak, "kernel": "k_trace_persistent (node records in global memory)",
           "launch_ms_avg": kern_avg_ms, "launches_per_step": launches_per_step,
           "peak_note": "dependent random record reads per second at their best (arrays of 0.16-1.3 GiB; = 7.0 TB/s of 128-byte lines), "
                        "scripts/micro/random_records.hip; table RANDOM_RECORD_RATE in bench.py, profiles/r03_random_records.txt",
           "uniform_random_rate_at_tree_size": random_record_peak(array_bytes),
           "tree_child_lines_GiB": array_bytes / 2.0 ** 30,
           "node_visits_per_s_G": rays_per_launch * visits_per_ray / (kern_avg_ms * 1e-3) / 1e9,
           "algorithmic_GBps": algorithmic, "algorithmic_frac_of_hbm_peak": algorithmic / HBM_PEAK_GBS,
           "algorithmic_note": "36 B per node visit + 36 B per triangle test over the launch time: NOT memory traffic (the top of "
                               "the tree is served by L1/L2, and a visit below it moves a whole line)"}
    if not k or "TCC_EA0_RDREQ_sum" not in k:
        out.update({"achieved": None, "frac": None, "traffic": None,
                    "note": "no counter profile of THIS workload (triangles, image size, frames per launch, options) under profiles/"})
        return out
    req = k["TCC_EA0_RDREQ_sum"]
    achieved = req / (kern_avg_ms * 1e-3) / 1e9
    out.update({"achieved": achieved, "frac": achieved / peak, "memory_read_requests_per_launch": req,
                "requests_per_ray": req / rays_per_launch,
                "traffic": k.get("hbm_bytes_per_launch") if prof["_current"] else None,
                "hbm_GBps": (k["hbm_bytes_per_launch"] / (kern_avg_ms * 1e-3) / 1e9) if k.get("hbm_bytes_per_launch") else None,
                "hbm_frac_of_peak": (k["hbm_bytes_per_launch"] / (kern_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if k.get("hbm_bytes_per_launch") else None,
                "l2_hit_rate": (k["TCC_HIT_sum"] / k["TCC_REQ_sum"]) if k.get("TCC_REQ_sum") else None,
                "profile": {"file": os.path.relpath(profile_path, ROOT), "taken_with_these_kernel_sources": prof["_current"],
                            "meta": prof.get("_meta")}})
    if not prof["_current"]:
        out["note"] = "counter profile was taken with other kernel sources: request counts may be stale, traffic withheld"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="scene04", choices=["scene04", "scene02_4k", "scene03_4k", "soup100m"],
                    help="scene04: the configuration the metric is quoted on; scene02_4k / scene03_4k: 3840x2160, 4096 spp over 8 GPUs "
                         "= 4 steps of 128 frames per GPU (pass --steps 4); soup100m: 100M triangles at 3840x2160, depth 8")
    ap.add_argument("--scene", default="scene04")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--build-tris", type=int, default=10_000_000)
    ap.add_argument("--build-tris-large", type=int, default=50_000_000)
    ap.add_argument("--soup-tris", type=int, default=100_000_000)
    ap.add_argument("--soup-frames", type=int, default=2)
    ap.add_argument("--near-first", type=int, default=1, help="soup100m: nearest-child-first traversal order (a labelled non-parity option)")
    ap.add_argument("--topology", default="morton", choices=["morton", "index"],
                    help="soup100m: tree over the sorted Morton codes (PT_BUILD_TOPOLOGY_MORTON, a labelled non-parity option) "
                         "or the reference's tree over the leaf index")
    ap.add_argument("--no-build", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        # never measure another number of GPUs than the one asked for
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}: refusing to run", file=sys.stderr)
        sys.exit(2)

    # a hang (a rank that never arrives at a collective) ends with every thread's stack on stderr instead of
    # holding the GPU until somebody else's time limit
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("PTAMD_BENCH_WATCHDOG_S", "900")), exit=True)

    import torch
    import torch.distributed as dist
    import ptamd
    import ptdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a box with fewer GPUs than ranks: PTAMD_BENCH_DEVICE pins every rank to one device and
    # PTAMD_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU); never used by the driver
    if os.environ.get("PTAMD_BENCH_DEVICE"):
        local_rank = int(os.environ["PTAMD_BENCH_DEVICE"])
    backend = os.environ.get("PTAMD_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    stream = torch.cuda.current_stream()
    soup_mode = args.config == "soup100m"
    if args.config in ("scene02_4k", "scene03_4k"):
        args.scene = args.config[:7]
        args.width, args.height = args.width or 3840, args.height or 2160

    def make_ctx():
        return ptamd.Context(local_rank, stream=stream.cuda_stream)

    ctx = make_ctx()
    g = None
    keep = None
    if soup_mode:
        W, H = args.width or 3840, args.height or 2160
        n = args.soup_tris
        keep, cb, scene_build = setup_soup_scene(ctx, n, W, H, args.topology, args.near_first)
        fpb = args.soup_frames
    else:
        W, H = args.width or 1162, args.height or 1008
        g = np.load(os.path.join(ROOT, "tests", "golden", args.scene + ".npz"))
        ctx.upload_primitives(g["prims"].view(ptamd.PRIM_RENDER))
        ctx.set_constants(g["cb"].view(ptamd.CB_RADIX))
        ctx.build()
        ctx.build()         # (the first build of a process also loads the kernels' code objects: 3.6 ms in its last stage)
        scene_build = ctx.build_times()
        ctx.resize(W, H)
        ctx.primary_hits(ptamd.default_camera(), ptamd.albedo_table(g["mat_table"][:, 0:3], True))
        fpb = ptdist.FRAMES_PER_BLOCK
    ptr, nbytes = ctx.linear_sum_ptr()
    lin = torch.as_tensor(_DevArray(ptr, (H, W, 4)), device=torch.device("cuda", local_rank))
    assert lin.data_ptr() == ptr and lin.numel() * 4 == nbytes

    # the merge of the accumulation sums: RCCL through the C ABI (pt_reduce_linear_sum).  The communicator's id
    # travels over torch.distributed; if any rank cannot create its communicator all ranks agree to fall back
    # to torch.distributed.all_reduce (the same RCCL), and the JSON line says which one ran.
    reduce_via = "none (one GPU)"
    rccl_ranks = 0          # ranks of the communicator pt_reduce_linear_sum uses, as ncclCommCount states them
    if world > 1:
        reduce_via = "torch.distributed.all_reduce"
        if backend == "nccl":
            ok = 0
            cid = [None]
            if rank == 0:
                try:
                    cid[0] = ptamd.unique_comm_id()
                except Exception as e:   # noqa: BLE001 -- the broadcast below still happens: nobody waits for a rank that gave up
                    print(f"bench.py rank 0: pt_comm_unique_id failed: {e}", file=sys.stderr)
            dist.broadcast_object_list(cid, src=0)
            if cid[0] is not None:
                try:
                    ctx.comm_init_rank(world, rank, cid[0])
                    ok = 1
                except Exception as e:   # noqa: BLE001 -- reported below, every rank takes the same branch
                    print(f"bench.py rank {rank}: pt_comm_init_rank failed: {e}", file=sys.stderr)
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                reduce_via = "pt_reduce_linear_sum (ncclAllReduce through the C ABI)"
                rccl_ranks = ctx.comm_count()
                if rccl_ranks != world:
                    print(f"bench.py rank {rank}: the RCCL communicator spans {rccl_ranks} ranks, {world} were launched", file=sys.stderr)
                    sys.exit(4)
            elif rank == 0:
                print("bench.py: WARNING: no RCCL communicator through the C ABI on at least one rank (see above); the sums are "
                      "merged by torch.distributed.all_reduce instead -- `reduce_fallback` is true in the JSON line", file=sys.stderr, flush=True)
        elif rank == 0:
            print(f"bench.py: WARNING: backend {backend} (rehearsal): the sums are merged by torch.distributed.all_reduce, not by "
                  "pt_reduce_linear_sum", file=sys.stderr, flush=True)

    def reduce():
        if world == 1:
            return
        if reduce_via.startswith("pt_reduce"):
            ctx.reduce_linear_sum()
        else:
            if backend != "nccl":
                # the gloo rehearsal stages the tensor through host memory; with trace kernels still in flight on
                # the stream its staging copy was seen never to complete (two processes on one GPU), so drain first
                torch.cuda.synchronize()
            ptdist.reduce_linear_sum(lin, dist)

    def note(msg):
        if os.environ.get("PTAMD_BENCH_VERBOSE"):
            print(f"[bench rank {rank} {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    def step(s):
        first, cnt = ptdist.block_frames(s, rank, world, fpb)
        ctx.trace_frames(first, ptamd.frame_seeds(1, first, cnt), ptamd.PT_ACCUM_LINEAR)

    # per-ray traversal statistics of this workload (instrumented kernel, one block): they
    # turn the ray count into the algorithmic bytes of the roofline model
    ctx.set_detailed_counters(True)
    ctx.reset_counters()
    step(0)
    det = ctx.counters()
    ctx.set_detailed_counters(False)
    visits_per_ray = det["node_visits"] / det["rays"]
    tris_per_ray = det["tri_tests"] / det["rays"]
    bytes_per_ray = visits_per_ray * BYTES_PER_VISIT + tris_per_ray * BYTES_PER_TRI_TEST

    note("statistics block traced")
    ctx.set_trace_timing(True)      # HIP events around the trace kernel of every launch, on the library's stream
    for s in range(args.warmup):
        step(s)
    # (the first launches differ from the steady state: no per-pixel cost order yet, cold code; they are reported on their
    # own so that a profiler's per-kernel average over ALL launches of the run can be checked against the line)
    warm_ms = ctx.trace_times()
    if world > 1:
        torch.cuda.synchronize()
        note("warm-up traced")
        dist.barrier()
        note("barrier passed")
        reduce()                    # communicator warm-up outside the timed region
        torch.cuda.synchronize()
        note("warm-up reduce done")
    ctx.reset_accumulation()
    ctx.reset_counters()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(s)
    reduce()
    ctx.resolve()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0

    cnt = ctx.counters()
    tot = torch.tensor([dt, float(cnt["rays"]), float(cnt["paths"])], dtype=torch.float64, device="cuda")
    if world > 1:
        tmax = tot[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt = float(tmax.item())
    rays, paths = float(tot[1].item()), float(tot[2].item())
    kern_ms = ctx.trace_times()
    # one trace kernel launch per step, or several when a step's samples exceed the staging buffer (4K images)
    assert len(kern_ms) >= args.steps and len(kern_ms) % args.steps == 0, "whole launches per step expected"
    kern_avg_ms = float(kern_ms.mean())
    rays_per_launch = cnt["rays"] / len(kern_ms)

    # sanity of the result image: every pixel saw every frame of the job
    total_frames = ptdist.all_frames(args.steps, world, fpb)
    alpha = lin[..., 3]
    assert float(alpha.min().item()) == float(total_frames) == float(alpha.max().item()), "accumulation count mismatch"

    if soup_mode:
        roofline = trace_roofline_soup(kern_avg_ms, rays_per_launch, visits_per_ray, bytes_per_ray, len(kern_ms) // args.steps,
                                       int(ctx.n), PROFILE_SOUP.get((args.topology, int(bool(args.near_first))))
                                       if (W, H, fpb, int(ctx.n)) == (3840, 2160, 2, 100_000_000) else None)
    else:
        roofline = trace_roofline_scene(kern_avg_ms, rays_per_launch, bytes_per_ray, len(kern_ms) // args.steps,
                                        PROFILE_TRACE.get((args.scene, W, H, fpb)) if len(kern_ms) == args.steps else None)

    roofline["warmup_launch_ms"] = [round(float(x), 3) for x in warm_ms]

    builds = {}
    if not args.no_build and not soup_mode:
        copy_gbs = hbm_copy_gbs()
        for key, n, on_device in (("build", args.build_tris, False), ("build_large", args.build_tris_large, True)):
            if n <= 0:
                continue
            b = bench_build(make_ctx, n, 5, on_device, copy_gbs)
            if world > 1:
                t = torch.tensor([b["mtris_per_s"]], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                b["mtris_per_s_all_gpus"] = float(t.item())
                b["note"] = "replicas only: every GPU builds the whole soup"
            else:
                b["mtris_per_s_all_gpus"] = b["mtris_per_s"]
            builds[key] = b

    if rank == 0:
        if soup_mode:
            workload = (f"synthetic {ctx.n}-triangle soup {W}x{H}, at most 8 rays per path, {total_frames} spp "
                        f"({args.steps} steps x {fpb} frames per GPU), full-resolution codes, deep stack, "
                        f"{'tree over the Morton codes' if args.topology == 'morton' else 'index-keyed tree'}, "
                        f"{'nearest-child-first' if args.near_first else 'reference'} traversal order (NON-PARITY options)")
            data = "soup generated on the device (torch generator seed 2024), fixed frame seeds mt19937(1), G-buffer by ray cast"
        else:
            workload = (f"{args.scene}.obj {W}x{H}, {total_frames} spp progressive ({args.steps} steps x {fpb} frames per GPU, "
                        f"linear-sum accumulation, RCCL all-reduce + resolve at the end)")
            data = f"{args.scene} fixture (reference asset), fixed seeds mt19937(1), G-buffer by ray cast"
        out = {
            "metric": "Mrays/s (whole node, scene04 1162x1008) + LBVH build Mtris/s @1/2/4/8 GPU",
            "value": rays / dt / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "rccl_ranks": rccl_ranks,
            "reduce_fallback": world > 1 and not reduce_via.startswith("pt_reduce"),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": data,
            "config": {"workload": workload, "triangles": int(ctx.n), "parallelism": f"frames{world}", "reduce": reduce_via},
            "mpaths_per_s": paths / dt / 1e6,
            # the reference's only published figure for this path (report.md:96): another GPU, unnamed scene,
            # a different unit than the metric -- listed for orientation, not as vs_baseline
            "reference_published": {"mpaths_per_s": 14.0, "hardware": "NVIDIA GTX 1060", "image": "1k x 1k", "source": "report.md:96"},
            "rays_per_path": rays / paths,
            "traversal": {"node_visits_per_ray": visits_per_ray, "tri_tests_per_ray": tris_per_ray,
                          "bytes_per_ray": bytes_per_ray},
            "roofline": roofline,
            "scene_build_ms": scene_build["total"],
            "scene_build_stage_ms": {k: scene_build[k] for k in ptamd.STAGES},
            "kernel_sources_sha256_16": sources_hash(),
        }
        out.update(builds)
        if world == 1 and not args.no_cpu_baseline and not soup_mode:
            out["cpu_baseline"] = cpu_baseline(g, W, H)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
