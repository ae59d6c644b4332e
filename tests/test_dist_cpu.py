"""CPU, two processes over gloo: the multi-GPU partition (frame blocks dealt round-robin to ranks,
one all-reduce of the linear sums, resolve) reproduces the single-process image.  The per-rank
renderer here is the oracle, standing in for the GPU kernel it is bit-identical to."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ptdist
from oracle import orc
from conftest import load_golden

W, H, STEPS, FPB = 40, 32, 2, 4


def _render_blocks(rank, world, steps):
    g = load_golden("scene01")
    sc = orc.OracleScene(g["prims"], g["tree"], g["cb"])
    kd = [list(m[0:3]) for m in g["mat_table"]] + [[0, 0, 0]] * (16 - len(g["mat_table"]))
    gb = orc.primary(sc, orc.default_camera(), W, H, orc.albedo_table(g["cb"], True, kd_override=kd), threads=1)
    acc = np.zeros((H, W, 4), np.float32)
    for s in range(steps):
        first, cnt = ptdist.block_frames(s, rank, world, FPB)
        seeds = orc.frame_seeds(1, first, cnt)
        for i in range(cnt):
            orc.trace_frame_linear(sc, gb, first + i, seeds[i], acc, threads=1)
    return acc


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    acc = torch.from_numpy(_render_blocks(rank, world, STEPS))
    ptdist.reduce_linear_sum(acc, dist)
    if rank == 0:
        np.save(os.path.join(out_dir, "sum.npy"), acc.numpy())
    dist.destroy_process_group()


def test_block_assignment_covers_every_frame_once():
    for world in (1, 2, 4, 8):
        seen = []
        for s in range(3):
            for r in range(world):
                first, cnt = ptdist.block_frames(s, r, world)
                assert cnt == ptdist.FRAMES_PER_BLOCK and first % cnt == 0
                seen.extend(range(first, first + cnt))
        assert sorted(seen) == list(range(ptdist.all_frames(3, world)))


def test_two_rank_sum_equals_single_process(tmp_path):
    world = 2
    port = 29500 + os.getpid() % 2000
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    total = np.load(tmp_path / "sum.npy")
    # the same frames traced by one process, block after block
    parts = [_render_blocks(r, world, STEPS) for r in range(world)]
    assert np.array_equal(total, parts[0] + parts[1])            # the reduction is exactly the sum of the rank images
    assert np.all(total[..., 3] == ptdist.all_frames(STEPS, world, FPB))
    # against strictly sequential accumulation of the same frames: float re-association only
    g = load_golden("scene01")
    sc = orc.OracleScene(g["prims"], g["tree"], g["cb"])
    kd = [list(m[0:3]) for m in g["mat_table"]] + [[0, 0, 0]] * (16 - len(g["mat_table"]))
    gb = orc.primary(sc, orc.default_camera(), W, H, orc.albedo_table(g["cb"], True, kd_override=kd), threads=1)
    seq = np.zeros((H, W, 4), np.float32)
    nf = ptdist.all_frames(STEPS, world, FPB)
    seeds = orc.frame_seeds(1, 0, nf)
    for f in range(nf):
        orc.trace_frame_linear(sc, gb, f, seeds[f], seq, threads=1)
    np.testing.assert_allclose(total, seq, rtol=1e-5, atol=1e-6)
    a, _ = orc.resolve(total, nf)
    b, _ = orc.resolve(seq, nf)
    assert np.abs(a - b).max() < 1e-5
