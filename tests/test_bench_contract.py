"""CPU: the benchmark's output contract, checked on the committed record of the last GPU run
(profiles/r03_bench_default.json) and on bench.py's command line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_every_contract_field():
    line = open(os.path.join(ROOT, "profiles", "r03_bench_default.json")).read().strip().splitlines()[-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Mrays/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None                       # the reference publishes no number for this metric
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    # the trace kernel with its tree in LDS is bound by vector-instruction issue: the fraction is achieved / peak of
    # THAT unit and can be recomputed from the committed counter summary and the launch time of the line
    assert r["bound"] in ("hbm", "mfma", "valu") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0 < r["frac"] <= 1
    pmc = json.load(open(os.path.join(ROOT, r["profile"]["file"])))
    valu = pmc[r["kernel"]]["SQ_INSTS_VALU"]
    assert abs(r["achieved"] - valu / (r["launch_ms_avg"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["profile"]["taken_with_these_kernel_sources"] is True
    assert r["profile"]["meta"]["sources_sha256_16"] == d["kernel_sources_sha256_16"]
    assert abs(r["traffic"] - pmc[r["kernel"]]["hbm_bytes_per_launch"]) < 1
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port")
    # value = rays of all steps / wall time: consistent with ms_per_step and the ray statistics
    rays = d["value"] * 1e6 * d["ms_per_step"] * 1e-3 * d["steps"]
    paths = d["mpaths_per_s"] * 1e6 * d["ms_per_step"] * 1e-3 * d["steps"]
    assert abs(rays / paths - d["rays_per_path"]) < 1e-6 * d["rays_per_path"]
    for key, n in (("build", 10_000_000), ("build_large", 50_000_000)):
        b = d[key]
        assert b["roofline"]["bound"] == "hbm" and 0 < b["roofline"]["frac"] <= 1
        assert abs(b["roofline"]["frac"] - b["roofline"]["achieved"] / b["roofline"]["peak"]) < 1e-9
        assert abs(b["mtris_per_s"] - n / (b["ms"] * 1e-3) / 1e6) < 1e-6 * b["mtris_per_s"]
        # the headline build is the default one (tree + the trace kernels' records); tree-only is the faster, secondary figure
        assert abs(b["roofline"]["achieved"] - n * 472 / (b["ms"] * 1e-3) / 1e9) < 1e-6 * b["roofline"]["achieved"]
        assert b["tree_only"]["ms"] < b["ms"]
    for k in ("cpu", "single_thread_mrays_per_s"):
        assert k in c, k


def test_kernel_stats_agree_with_the_bench_line():
    """The profiler's average duration of the trace kernel and the HIP-event time inside bench.py
    come from the same run and must agree."""
    import csv
    d = json.loads(open(os.path.join(ROOT, "profiles", "r03_bench_default.json")).read().strip().splitlines()[-1])
    avg, total, calls = None, 0.0, 0
    for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r03_bench_default_kernel_stats.csv"))):
        # the variant the benchmark's steps run: the uninstrumented one (first template argument false) of the kernel the
        # line names; of several matching rows (other variants ran once for the statistics block) the one that took the most time
        if (d["roofline"]["kernel"] + "<false") in r["Name"] and float(r["TotalDurationNs"]) > total:
            total, avg, calls = float(r["TotalDurationNs"]), float(r["AverageNs"]) * 1e-6, int(r["Calls"])
    assert avg is not None
    # the profiler's average is over every launch of the run, warm-up included; the line's is over the timed steps and
    # names the warm-up launches (no cost order yet, cold code) separately
    r = d["roofline"]
    timed = d["steps"] * r["launches_per_step"]
    whole_run = (sum(r["warmup_launch_ms"]) + timed * r["launch_ms_avg"]) / (len(r["warmup_launch_ms"]) + timed)
    assert len(r["warmup_launch_ms"]) + timed == calls
    assert abs(avg - whole_run) < 0.01 * avg


def test_every_committed_bench_line_has_a_counter_backed_roofline():
    """The lines of the other named configurations (3840x2160 scenes, the 100M-triangle soup): no null in the roofline,
    and each figure recomputes from the counter summary the line names and the launch time in the line."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r03_bench_*.json")))
    assert len(files) >= 6
    for f in files:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert r.get(k) is not None, (f, k)
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] <= 1
        assert r["profile"]["taken_with_these_kernel_sources"] is True
        assert r["profile"]["meta"]["sources_sha256_16"] == d["kernel_sources_sha256_16"]
        pmc = json.load(open(os.path.join(ROOT, r["profile"]["file"])))
        if r["bound"] == "valu":
            k = pmc[r["kernel"]]
            assert abs(r["achieved"] - k["SQ_INSTS_VALU"] / (r["launch_ms_avg"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
        else:
            # trees far larger than the caches: memory-side read requests per second against the best rate of dependent
            # random record reads; the algorithmic 36 B per visit stays in the line as a labelled figure
            assert r["bound"] == "memory requests" and "soup100m" in f
            k = pmc["k_trace_persistent"]
            assert abs(r["achieved"] - k["TCC_EA0_RDREQ_sum"] / (r["launch_ms_avg"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
            assert r["algorithmic_GBps"] > 0 and r["uniform_random_rate_at_tree_size"] < r["peak"]
        assert abs(r["traffic"] - k["hbm_bytes_per_launch"]) < 1


def test_bench_command_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


def test_bench_refuses_a_rank_count_other_than_the_one_asked_for():
    """--gpus N under a launcher that made another number of ranks must fail, not measure that other number."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 2 and "refusing" in out.stderr and out.stdout.strip() == ""


def test_bench_without_a_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts a 2-rank torch.distributed.run job (which cannot
    succeed without GPUs: the point here is that it is started, and that its failure is the parent's exit code)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["PTAMD_BENCH_WATCHDOG_S"] = "60"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-build"],
                         env=env, capture_output=True, text=True, timeout=240)
    assert "starting 2 ranks" in out.stderr and "--nproc-per-node 2" in out.stderr
    import torch
    if not torch.cuda.is_available():
        assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_committed_records_belong_to_the_kernel_sources_in_the_tree():
    """Every r03 counter summary and bench line under profiles/ was taken with the kernel sources of this tree (the hash
    bench.py prints): after a kernel edit the records have to be collected again (scripts/gpu_collect_profiles.sh)."""
    import glob
    sys.path.insert(0, ROOT)
    import bench
    h = bench.sources_hash()
    summaries = sorted(glob.glob(os.path.join(ROOT, "profiles", "r03_*_pmc.json")))
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r03_bench_*.json")))
    assert len(summaries) >= 9 and len(lines) >= 6
    for f in summaries:
        assert json.load(open(f))["_meta"]["sources_sha256_16"] == h, f
    for f in lines:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        assert d["kernel_sources_sha256_16"] == h, f
        assert d["roofline"]["profile"]["taken_with_these_kernel_sources"] is True, f
