"""GPU: acceleration-structure build through the C ABI against the oracle and the golden fixtures.
Bit-exact: Morton codes, stable permutation, tree links, leaf payloads; boxes compared with == on floats."""
import numpy as np
import pytest

import ptamd
from oracle import orc
from conftest import SCENES, make_soup

pytestmark = pytest.mark.gpu


def _assert_build_equals(ctx, ob):
    n = ctx.n
    bb = ctx.scene_bound()
    assert np.array_equal(bb["min"], ob["bb"]["min"]) and np.array_equal(bb["max"], ob["bb"]["max"])
    assert np.array_equal(ctx.morton(), ob["rays"]["morton_code"])
    assert np.array_equal(ctx.permutation(), ob["sorted_rays"]["index"])
    cs = ctx.sorted_primitives()
    assert np.array_equal(cs["index"], ob["sorted_cs"]["index"])
    assert np.array_equal(cs["bound"]["min"], ob["sorted_cs"]["bound"]["min"])
    assert np.array_equal(cs["bound"]["max"], ob["sorted_cs"]["bound"]["max"])
    t, o = ctx.tree(), ob["tree"][:2 * n]
    for f in ("left", "right", "parent", "index", "padding"):
        assert np.array_equal(t[f], o[f]), f
    assert np.array_equal(t["bound"]["min"], o["bound"]["min"])
    assert np.array_equal(t["bound"]["max"], o["bound"]["max"])
    assert np.all(t["bound"]["pad"] == 0)


@pytest.mark.parametrize("scene", SCENES)
def test_scene_build_bit_exact(gpu_ctx_factory, golden, scene):
    g = golden[scene]
    ctx = gpu_ctx_factory()
    ctx.upload_primitives(g["prims"])
    ctx.build()
    ob = orc.build(g["prims"])
    _assert_build_equals(ctx, ob)
    # and against the committed fixtures
    assert np.array_equal(ctx.morton(), g["morton"])
    assert np.array_equal(ctx.permutation(), g["perm"])
    bt = ctx.build_times()
    assert bt["sort_passes"] == 2 and bt["total"] > 0


@pytest.mark.parametrize("n", [1, 2, 3, 5, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 2047, 2048, 2049,
                               4095, 4096, 4097, 8191, 12289, 65537, 300001])
def test_soup_build_bit_exact(gpu_ctx_factory, n):
    prims = make_soup(n, seed=1000 + n)
    ctx = gpu_ctx_factory()
    ctx.upload_primitives(prims)
    ctx.build()
    _assert_build_equals(ctx, orc.build(prims))


@pytest.mark.parametrize("n", [1, 2, 7, 862, 1025, 2416, 65537, 1000003])
def test_closed_form_links_equal_search_kernel(gpu_ctx_factory, n):
    """Links written from the closed form of the index-keyed tree (default) and by the
    search-based kernel that follows radix_build_cs.hlsl are the same, and equal the oracle's."""
    prims = make_soup(n, seed=n)
    ctx = gpu_ctx_factory()
    ctx.upload_primitives(prims)
    ctx.build()
    a = ctx.tree()
    ctx.build(ptamd.PT_BUILD_KARRAS_LINKS)
    b = ctx.tree()
    assert a.tobytes() == b.tobytes()
    o = orc.radix_tree(n)
    for f in ("left", "right", "parent"):
        assert np.array_equal(a[f], o[f][:2 * n]), f


@pytest.mark.parametrize("n", [1, 3, 1024, 1025, 2416, 300001])
def test_tree_only_build_gives_the_same_tree_and_refuses_to_trace(gpu_ctx_factory, n):
    """PT_BUILD_TREE_ONLY gathers boxes instead of primitives in the refit and writes no trace records:
    same tree, same order; ray queries are refused until a full build."""
    prims = make_soup(n, seed=77 + n)
    ctx = gpu_ctx_factory()
    ctx.upload_primitives(prims)
    ctx.build()
    full = ctx.tree().tobytes()
    perm = ctx.permutation().copy()
    ctx.build(ptamd.PT_BUILD_TREE_ONLY)
    assert ctx.tree().tobytes() == full
    assert np.array_equal(ctx.permutation(), perm)
    rays = np.zeros((4, 8), np.float32)
    rays[:, 5] = 1.0
    rays[:, 7] = 10000.0
    with pytest.raises(ptamd.PtError, match="TREE_ONLY"):
        ctx.intersect(rays)
    ctx.build()
    ctx.intersect(rays)


def test_rebuild_and_context_reuse(gpu_ctx_factory, golden):
    ctx = gpu_ctx_factory()
    for scene in ("scene04", "scene01", "scene04"):      # shrinking and growing n on one context
        g = golden[scene]
        ctx.upload_primitives(g["prims"])
        ctx.build()
        assert np.array_equal(ctx.permutation(), g["perm"])
        t = ctx.tree()
        assert np.array_equal(t["left"], g["tree"]["left"][:-1])


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_flat_scene_box_axis_quantises_to_cell_zero(gpu_ctx_factory, axis):
    """All vertices in one plane: (mid - min) / (max - min) is 0/0 on that axis.  The reference's x86
    conversion leaves zero in the bits it interleaves; library and oracle define the same."""
    n = 3000
    prims = make_soup(n, seed=31 + axis)
    prims["vertices"]["pos"][:, :, axis] = np.float32(0.25)
    ctx = gpu_ctx_factory()
    ctx.upload_primitives(prims)
    ctx.build()
    ob = orc.build(prims)
    _assert_build_equals(ctx, ob)
    bit = 2 - axis                                      # x is the highest bit of each triple
    assert not np.any(ctx.morton() & np.uint64(sum(1 << (3 * k + bit) for k in range(21))))


def test_heavy_ties_are_stable(gpu_ctx_factory):
    """All triangles in one Morton cell but one: the order of equal codes is ascending index."""
    n = 5000
    prims = make_soup(n, seed=5)
    prims["vertices"]["pos"] *= np.float32(1e-3)          # everything in cell 0 ...
    prims["vertices"]["pos"][n // 2] += np.float32(1.0)   # ... except one far triangle
    ctx = gpu_ctx_factory()
    ctx.upload_primitives(prims)
    ctx.build()
    ob = orc.build(prims)
    assert len(np.unique(ob["rays"]["morton_code"])) <= 3
    _assert_build_equals(ctx, ob)


def _expected_full_codes(prims):
    """PT_BUILD_MORTON_FULL: same arithmetic with 2^21 - 1 cells per axis (float32, strict IEEE)."""
    f = np.float32
    pos = prims["vertices"]["pos"].astype(f)
    mn, mx = pos.min(axis=1), pos.max(axis=1)
    bmn, bmx = mn.min(axis=0), mx.max(axis=0)
    mid = ((mn + mx).astype(f) / f(2)).astype(f)
    q = ((((mid - bmn).astype(f) / (bmx - bmn).astype(f)).astype(f)) * f(2097151.0)).astype(f)
    q = q.astype(np.uint64)
    code = np.zeros(len(prims), dtype=np.uint64)
    for bit in range(21):
        for axis, sh in ((0, 2), (1, 1), (2, 0)):
            code |= ((q[:, axis] >> np.uint64(bit)) & np.uint64(1)) << np.uint64(3 * bit + sh)
    return code


def test_full_resolution_codes_eight_passes(gpu_ctx_factory):
    n = 200003
    prims = make_soup(n, seed=77)
    ctx = gpu_ctx_factory()
    ctx.upload_primitives(prims)
    ctx.build(ptamd.PT_BUILD_MORTON_FULL)
    assert ctx.build_times()["sort_passes"] == 8
    codes = ctx.morton()
    assert np.array_equal(codes, _expected_full_codes(prims))
    assert codes.max() >= 2 ** 60
    perm = ctx.permutation()
    assert np.array_equal(perm, np.argsort(codes, kind="stable").astype(np.int32))
    # topology does not depend on the keys; leaves follow the permutation
    t = ctx.tree()
    o = orc.radix_tree(n)
    assert np.array_equal(t["left"], o["left"][:-1]) and np.array_equal(t["parent"], o["parent"][:-1])
    assert np.array_equal(t["index"][n:], perm)


@pytest.mark.parametrize("n", [1, 2, 4095, 4097, 12289, 300007])
def test_eight_pass_sort_single_read_and_two_read_passes_agree(gpu_ctx_factory, monkeypatch, n):
    """The eight passes over full-resolution codes read their keys once per pass (one count of all digits up front, then
    look-back over the tiles' status words); PTAMD_SORT_CHAINED=0 (read at pt_create) selects a counting pass per pass.
    Both must give numpy's stable order, also with heavy ties (triangles snapped onto a coarse grid share their codes)
    and for a ragged last tile."""
    prims = make_soup(n, seed=500 + n)
    if n > 4097:   # half of the triangles share 64 centres: long runs of equal keys across many tiles
        snap = prims["vertices"]["pos"][: n // 2]
        snap[...] = np.floor(snap * 4) / 4
    perms = []
    for chained in ("1", "0"):
        monkeypatch.setenv("PTAMD_SORT_CHAINED", chained)
        ctx = gpu_ctx_factory()
        ctx.upload_primitives(prims)
        ctx.build(ptamd.PT_BUILD_MORTON_FULL | ptamd.PT_BUILD_TREE_ONLY)
        ctx.build(ptamd.PT_BUILD_MORTON_FULL | ptamd.PT_BUILD_TREE_ONLY)      # status words and counters are reset per build
        assert ctx.build_times()["sort_passes"] == 8
        codes = ctx.morton()
        perms.append(ctx.permutation())
        assert np.array_equal(perms[-1], np.argsort(codes, kind="stable").astype(np.int32))
    assert np.array_equal(perms[0], perms[1])


def test_tree_export_is_on_demand_and_repeatable(gpu_ctx_factory, golden):
    """The reference-layout tree is derived from the traversal records when asked for: asking
    twice, or after tracing, gives the same bytes; traversal works without ever asking."""
    g = golden["scene01"]
    ctx = gpu_ctx_factory()
    ctx.upload_primitives(g["prims"])
    ctx.build()
    hits = ctx.intersect(np.array([[0, 5, 4, 1e-4, 0, 0, -1, 1e4]], dtype=np.float32))
    assert hits["hit"][0] == 1
    a = ctx.tree()
    b = ctx.tree()
    assert a.tobytes() == b.tobytes()
    for f in ("left", "right", "parent", "index"):
        assert np.array_equal(a[f], g["tree"][f][:-1]), f
    assert np.array_equal(a["bound"]["min"], g["tree"]["bound"]["min"][:-1])
    assert np.array_equal(a["bound"]["max"], g["tree"]["bound"]["max"][:-1])


def test_large_soup_properties(gpu_ctx_factory):
    """10M triangles (the size of the build benchmark): properties that need no CPU build."""
    n = 10_000_000
    prims = make_soup(n, seed=12345)
    ctx = gpu_ctx_factory()
    ctx.upload_primitives(prims)
    ctx.build()
    codes, perm = ctx.morton(), ctx.permutation()
    ks = codes[perm]
    assert np.all((ks[:-1] < ks[1:]) | ((ks[:-1] == ks[1:]) & (perm[:-1] < perm[1:])))
    assert np.array_equal(np.sort(perm), np.arange(n, dtype=np.int32))
    # codes: spot-check 100k triangles against the oracle arithmetic
    sel = np.random.default_rng(0).choice(n, 100_000, replace=False)
    pos = prims["vertices"]["pos"]
    bb = ctx.scene_bound()
    assert np.array_equal(bb["min"][0], pos.reshape(-1, 3).min(axis=0))
    assert np.array_equal(bb["max"][0], pos.reshape(-1, 3).max(axis=0))
    sub = np.zeros(len(sel), dtype=orc.PRIM_RAY)
    sub["bound"]["min"] = pos[sel].min(axis=1)
    sub["bound"]["max"] = pos[sel].max(axis=1)
    import ctypes as C
    orc.lib().orc_morton_all(sub.ctypes.data_as(C.c_void_p), C.c_int(len(sel)), bb.ctypes.data_as(C.c_void_p))
    assert np.array_equal(sub["morton_code"], codes[sel])
    t = ctx.tree()
    # every internal box is the union of its children's boxes (vectorised over all nodes but
    # those whose right child is the out-of-range leaf)
    l, r = t["left"][:n], t["right"][:n]
    ok = r < 2 * n
    mn = np.minimum(t["bound"]["min"][l[ok]], t["bound"]["min"][r[ok]])
    mx = np.maximum(t["bound"]["max"][l[ok]], t["bound"]["max"][r[ok]])
    assert np.array_equal(t["bound"]["min"][:n][ok], mn) and np.array_equal(t["bound"]["max"][:n][ok], mx)
    assert np.array_equal(t["parent"][l], np.arange(n)) and np.array_equal(t["parent"][r[ok]], np.arange(n)[ok])
    assert np.array_equal(t["index"][n:], perm)
    assert np.array_equal(t["bound"]["min"][n:], pos[perm].min(axis=1))


def test_error_paths(gpu_ctx_factory, golden):
    ctx = gpu_ctx_factory()
    with pytest.raises(ptamd.PtError, match="no primitives"):
        ctx.build()
    with pytest.raises(ptamd.PtError):
        ctx.upload_primitives(np.zeros(0, dtype=ptamd.PRIM_RENDER))
    g = golden["scene01"]
    ctx.upload_primitives(g["prims"])
    with pytest.raises(ptamd.PtError, match="pt_build first"):
        ctx.morton()
    with pytest.raises(ptamd.PtError, match="unknown flag"):
        ctx.build(64)
    ctx.build()
    ctx.resize(32, 32)
    with pytest.raises(ptamd.PtError, match="pt_set_constants"):
        ctx.trace(0, 1)
    cb = g["cb"].copy()
    cb["node_count"] = 5
    ctx.set_constants(cb)
    with pytest.raises(ptamd.PtError, match="node_count"):
        ctx.trace(0, 1)
    ctx.set_constants(g["cb"])
    with pytest.raises(ptamd.PtError, match="G-buffer"):
        ctx.trace(0, 1)
    with pytest.raises(ptamd.PtError):
        ctx.resize(0, 10)
