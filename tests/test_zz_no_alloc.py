"""GPU: pt_build_tree_noalloc (the build that allocates nothing: all intermediate buffers in the library's
static device array) against pt_build on the same primitives.  Bit-exact: codes, permutation, all 2n
tree nodes.  Two calls in a row on one context, the second with a smaller input, so that whatever the
first call left in the shared array would show; then the reference scenes against the committed fixtures."""
import numpy as np
import pytest
import torch

import ptamd
from conftest import SCENES, make_soup

pytestmark = pytest.mark.gpu


def _expected(ctx, prims, flags=0):
    ctx.upload_primitives(prims)
    ctx.build(flags)
    return ctx.morton().copy(), ctx.permutation().copy(), ctx.tree().copy()


def _noalloc(ctx, prims, flags=0, want_codes=True):
    n = len(prims)
    dev = torch.device("cuda", 0)
    d_prims = torch.from_numpy(np.ascontiguousarray(prims).view(np.uint8).reshape(-1)).to(dev)
    # outputs pre-filled with a pattern: every byte the call owes must be overwritten
    d_codes = torch.full((n,), -1, dtype=torch.int64, device=dev)
    d_perm = torch.full((n,), -7, dtype=torch.int32, device=dev)
    d_tree = torch.full((2 * n * 12,), 0x5A5A5A5A, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.build_tree_noalloc(d_prims.data_ptr(), n, d_codes.data_ptr() if want_codes else 0, d_perm.data_ptr(), d_tree.data_ptr(), flags)
    ctx.synchronize()
    codes = d_codes.cpu().numpy().view(np.uint64)
    perm = d_perm.cpu().numpy()
    tree = d_tree.cpu().numpy().view(ptamd.TREENODE)
    return codes, perm, tree


def _same_tree(a, b, bits=True):
    """bits: boxes compared as raw bits (library against library); otherwise with == on floats, since
    min/max of +0 and -0 may keep either zero on the CPU (the x flip of the loader makes -0.0)."""
    for f in ("left", "right", "parent", "index", "padding"):
        assert np.array_equal(a[f], b[f]), f
    for f in ("min", "pad", "max"):
        if bits:
            assert np.array_equal(a["bound"][f].view(np.uint32), b["bound"][f].view(np.uint32)), f
        else:
            assert np.array_equal(a["bound"][f], b["bound"][f]), f


def test_noalloc_build_equals_pt_build_twice_in_a_row(gpu_ctx_factory):
    ref = gpu_ctx_factory()
    ctx = gpu_ctx_factory()
    for n, seed in ((100_003, 5), (4_099, 6)):            # the second call is smaller than the first
        prims = make_soup(n, seed=seed, dtype=ptamd.PRIM_RENDER)
        e_codes, e_perm, e_tree = _expected(ref, prims)
        codes, perm, tree = _noalloc(ctx, prims)
        assert np.array_equal(codes, e_codes)
        assert np.array_equal(perm, e_perm)
        _same_tree(tree, e_tree)


@pytest.mark.parametrize("n", [1, 2, 3, 1023, 1024, 1025, 4096, 4097, ptamd.PT_NOALLOC_MAX_PRIMS])
def test_noalloc_build_edge_sizes(gpu_ctx_factory, n):
    ref = gpu_ctx_factory()
    ctx = gpu_ctx_factory()
    prims = make_soup(n, seed=900 + n % 97, dtype=ptamd.PRIM_RENDER)
    e_codes, e_perm, e_tree = _expected(ref, prims, ptamd.PT_BUILD_KARRAS_LINKS)
    codes, perm, tree = _noalloc(ctx, prims, ptamd.PT_BUILD_KARRAS_LINKS, want_codes=(n % 2 == 1))
    if n % 2 == 1:
        assert np.array_equal(codes, e_codes)
    assert np.array_equal(perm, e_perm)
    _same_tree(tree, e_tree)


def test_noalloc_build_leaves_the_free_device_memory_unchanged(gpu_ctx_factory):
    """hipMemGetInfo before and after: calls of the entry point take no device memory (nothing allocated, nothing
    freed), at two sizes and with the codes on and off.  The first call of a process is made before the measurement: the
    HIP runtime loads a library's code objects into device memory when its first kernel is launched."""
    ctx = gpu_ctx_factory()
    dev = torch.device("cuda", 0)
    sizes = (50_000, 3_000)
    bufs = []
    for n in sizes:
        prims = make_soup(n, seed=n, dtype=ptamd.PRIM_RENDER)
        bufs.append((n, torch.from_numpy(np.ascontiguousarray(prims).view(np.uint8).reshape(-1)).to(dev),
                     torch.empty((n,), dtype=torch.int64, device=dev), torch.empty((n,), dtype=torch.int32, device=dev),
                     torch.empty((2 * n * 12,), dtype=torch.int32, device=dev)))
    n, p, c, m, t = bufs[0]
    ctx.build_tree_noalloc(p.data_ptr(), n, c.data_ptr(), m.data_ptr(), t.data_ptr(), ptamd.PT_BUILD_KARRAS_LINKS)
    ctx.synchronize()
    torch.cuda.synchronize()
    free_before, total = torch.cuda.mem_get_info(dev)
    for rep in range(3):
        for n, p, c, m, t in bufs:
            ctx.build_tree_noalloc(p.data_ptr(), n, c.data_ptr() if rep % 2 == 0 else 0, m.data_ptr(), t.data_ptr(),
                                   ptamd.PT_BUILD_KARRAS_LINKS if rep == 1 else 0)
            # (no synchronisation inside the loop either: a caller under graph capture could not make one)
    free_enqueued, _ = torch.cuda.mem_get_info(dev)
    ctx.synchronize()
    free_after, _ = torch.cuda.mem_get_info(dev)
    assert free_before == free_enqueued == free_after, (free_before, free_enqueued, free_after)
    perm = bufs[1][3].cpu().numpy()
    assert np.array_equal(np.sort(perm), np.arange(sizes[1]))       # and the calls did their work


@pytest.mark.parametrize("scene", SCENES)
def test_noalloc_build_reference_scenes(gpu_ctx_factory, golden, scene):
    g = golden[scene]
    ctx = gpu_ctx_factory()
    codes, perm, tree = _noalloc(ctx, g["prims"].view(ptamd.PRIM_RENDER))
    assert np.array_equal(codes, g["morton"])
    assert np.array_equal(perm, g["perm"])
    _same_tree(tree, g["tree"][:-1].view(ptamd.TREENODE), bits=False)


def test_noalloc_build_rejects_what_it_cannot_hold(gpu_ctx_factory):
    ctx = gpu_ctx_factory()
    buf = torch.zeros(1024, dtype=torch.int32, device="cuda")
    with pytest.raises(ptamd.PtError, match="PT_NOALLOC_MAX_PRIMS"):
        ctx.build_tree_noalloc(buf.data_ptr(), ptamd.PT_NOALLOC_MAX_PRIMS + 1, 0, buf.data_ptr(), buf.data_ptr())
    with pytest.raises(ptamd.PtError):
        ctx.build_tree_noalloc(buf.data_ptr(), 0, 0, buf.data_ptr(), buf.data_ptr())
    with pytest.raises(ptamd.PtError, match="PT_BUILD_KARRAS_LINKS"):
        ctx.build_tree_noalloc(buf.data_ptr(), 8, 0, buf.data_ptr(), buf.data_ptr(), ptamd.PT_BUILD_MORTON_FULL)
    with pytest.raises(ptamd.PtError, match="aligned"):
        ctx.build_tree_noalloc(buf.data_ptr() + 4, 8, 0, buf.data_ptr(), buf.data_ptr())
