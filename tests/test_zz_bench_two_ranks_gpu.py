"""GPU: bench.py's multi-rank path rehearsed on ONE device -- two ranks launched by torch.distributed.run,
both pinned to device 0, gloo in place of RCCL (which refuses two ranks on one GPU).  Exercises the frame
partition, the all-reduce of the accumulation sums, the max-over-ranks timing and the JSON line; bench.py
itself asserts that every pixel received every frame of the job."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_ranks_on_one_device():
    env = dict(os.environ, PTAMD_BENCH_DEVICE="0", PTAMD_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-build",
           "--width", "581", "--height", "504"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                 # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "frames2"
    assert "512 spp" in d["config"]["workload"]                # 2 ranks x 2 steps x 128 frames
    assert d["value"] > 0 and "cpu_baseline" not in d          # the CPU baseline runs at N = 1 only
    assert d["config"]["reduce"] == "torch.distributed.all_reduce"     # gloo rehearsal; RCCL goes through pt_reduce_linear_sum


def test_bench_starts_its_own_ranks():
    """Plain `python bench.py --gpus 2` (no launcher, WORLD_SIZE unset): the parent starts the two ranks itself,
    passes rank 0's line through and reports two GPUs -- never a silent one-GPU run."""
    env = dict(os.environ, PTAMD_BENCH_DEVICE="0", PTAMD_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-build",
           "--width", "581", "--height", "504"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "frames2"
    # the rehearsal merges over gloo: the line must say that the C-ABI RCCL reduce did not run
    assert d["reduce_fallback"] is True and d["rccl_ranks"] == 0
    assert "WARNING" in out.stderr
