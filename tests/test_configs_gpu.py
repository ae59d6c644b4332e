"""GPU: the named benchmark configurations that are larger than the reference's four scenes, each at the
bar its size allows.

  * LBVH build of a 50M-triangle soup: size-independent properties (sortedness with stable ties, permutation,
    every box the union of its children, parents, leaves), soup generated on the device.
  * The option set of the 100M-triangle render -- full-resolution codes (PT_BUILD_MORTON_FULL), deep stack,
    pt_set_max_segments(8), node records in global memory, 4 diffuse + 1 emissive material -- against the
    oracle run with exactly those options, bit for bit, at n = 200k; then at the full 100M / 3840x2160 size
    through properties (sample counts, finite sums, identical counters and image on a second run).
  * The non-parity traversal order (PT_TRAVERSAL_NEAR_FIRST) against the oracle's restatement of it, and
    against the reference order on the reference's scenes.
All of these are NON-PARITY configurations except the 50M build: the reference's 15-entry stack drops
subtrees of trees this deep and its codes have 22 cells per axis."""
import numpy as np
import pytest

import ptamd
from oracle import orc
from conftest import SCENES, device_soup, make_soup, soup_cb5

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


def _same_tree(a, b):
    """links as integers, boxes with == on floats: min / max of +0 and -0 may keep either zero on the CPU (the
    loader's x flip makes -0.0 in the reference's scenes)"""
    for f in ("left", "right", "parent", "index", "padding"):
        assert np.array_equal(a[f], b[f]), f
    for f in ("min", "pad", "max"):
        assert np.array_equal(a["bound"][f], b["bound"][f]), f


CAM = dict(eye=(50.0, 50.0, 300.0), at=(50.0, 50.0, 50.0))


def _soup_setup(ctx, prims, W, H, flags, near_first=False):
    n = len(prims)
    cb = soup_cb5(n, eye=CAM["eye"])
    ctx.upload_primitives(prims); ctx.set_constants(cb); ctx.build(flags); ctx.resize(W, H)
    ctx.set_stack_size(32)
    ctx.set_max_segments(8)
    ctx.set_traversal_order(ptamd.PT_TRAVERSAL_NEAR_FIRST if near_first else ptamd.PT_TRAVERSAL_REFERENCE)
    alb = ptamd.albedo_table([list(cb["mats"][0][m]["kd"][:3]) for m in range(16)])
    cam = ptamd.default_camera(**CAM); cam.zfar = 1000.0
    ctx.primary_hits(cam, alb)
    return cb, alb


@pytest.mark.parametrize("pair_lines", [True, False])
@pytest.mark.parametrize("near_first", [False, True])
def test_large_scene_option_set_against_the_oracle(gpu_ctx_factory, monkeypatch, near_first, pair_lines):
    """n = 200k (18 levels, node records 9.6 MB: global memory), full-resolution codes, stack 32, at most 8
    rays per path, 4 diffuse + 1 emissive material, cube scaled to 100 units (the triangle test rejects
    |det| <= 1e-4 on unnormalised edges, ray_tracing.hlsl:172).  Library == oracle with the same options, with the
    128-byte child lines the traversal reads (the records of both children of a node in one line) and with the plain
    64-byte records (PTAMD_PAIR_MAX_MIB=0, read when the context is created)."""
    n, W, H = 200_000, 200, 150
    # child lines are used from 24 MiB of plain records on: forced on (or off) for this small tree
    monkeypatch.setenv("PTAMD_PAIR_MIN_MIB", "0")
    if not pair_lines:
        monkeypatch.setenv("PTAMD_PAIR_MAX_MIB", "0")
        # plain records, upper levels in LDS: in the reference's order also with two paths per lane (a kernel that
        # launches with items of four frames or more use; forced onto this two-frame launch)
        monkeypatch.setenv("PTAMD_DUAL_MIN_CHUNK", "1")
    prims = make_soup(n, seed=4242, dtype=ptamd.PRIM_RENDER)
    prims["vertices"]["pos"] *= np.float32(100.0)
    prims["matid"] = np.where(np.arange(n) % 16 == 0, 4, np.arange(n) % 4)
    ctx = gpu_ctx_factory()
    cb, alb = _soup_setup(ctx, prims, W, H, ptamd.PT_BUILD_MORTON_FULL, near_first)
    ob = orc.build_full_codes(prims.view(orc.PRIM_RENDER))
    assert np.array_equal(ctx.morton(), ob["rays"]["morton_code"])
    assert np.array_equal(ctx.permutation(), ob["sorted_rays"]["index"])
    assert ctx.tree().tobytes() == ob["tree"][:2 * n].tobytes()
    ocam = orc.default_camera(**CAM); ocam.zfar = 1000.0
    sc = orc.OracleScene(prims.view(orc.PRIM_RENDER), ob["tree"], cb, stack_size=64, max_segments=8, near_first=near_first)
    # the G-buffer pass always visits in the reference's order
    gb = orc.primary(orc.OracleScene(prims.view(orc.PRIM_RENDER), ob["tree"], cb, stack_size=64), ocam, W, H, alb)
    d, nn, w = ctx.download_gbuffer()
    assert np.array_equal(d, gb[0]) and np.array_equal(_bits(nn), _bits(gb[1])) and np.array_equal(_bits(w), _bits(gb[2]))
    ctx.set_detailed_counters(True)
    seeds = ptamd.frame_seeds(1, 0, 2)
    ctx.trace_frames(0, seeds, ptamd.PT_ACCUM_LINEAR)
    acc = np.zeros((H, W, 4), np.float32)
    tot = dict(paths=0, rays=0, node_visits=0, box_tests=0, tri_tests=0)
    for f in range(2):
        c = orc.trace_frame_linear(sc, gb, f, seeds[f], acc)
        tot = {k: tot[k] + c[k] for k in tot}
    assert np.array_equal(_bits(ctx.linear_sum()), _bits(acc))
    got = ctx.counters()
    assert got["paths"] == tot["paths"] and got["rays"] == tot["rays"] and got["tri_tests"] == tot["tri_tests"]
    # node visits: the library counts both children at every expansion, the oracle every pop; with the
    # reference's order these are the same events, nearest-first the oracle's ordering tests are not counted
    assert got["node_visits"] == tot["node_visits"]
    assert 0.5 * tot["paths"] < tot["rays"] <= 8 * tot["paths"]
    print("near_first", near_first, "visits/ray", got["node_visits"] / got["rays"], "tests/ray", got["tri_tests"] / got["rays"])


@pytest.mark.parametrize("scene", SCENES)
def test_near_first_order_on_the_reference_scenes(gpu_ctx_factory, golden, scene):
    """PT_TRAVERSAL_NEAR_FIRST on the reference's scenes (tree in LDS): equal to the oracle's restatement of the
    same order bit for bit; against the reference's order the images may differ only where two triangles are hit
    at (nearly) the same distance -- counted and bounded here."""
    W, H = 160, 136
    g = golden[scene]
    ctx = gpu_ctx_factory()
    ctx.upload_primitives(g["prims"]); ctx.set_constants(g["cb"]); ctx.build(); ctx.resize(W, H)
    alb = ptamd.albedo_table(g["mat_table"][:, 0:3], True)
    ctx.primary_hits(ptamd.default_camera(), alb)
    gb = ctx.download_gbuffer()
    seeds = ptamd.frame_seeds(1, 0, 4)
    res = {}
    for near in (False, True):
        ctx.set_traversal_order(ptamd.PT_TRAVERSAL_NEAR_FIRST if near else ptamd.PT_TRAVERSAL_REFERENCE)
        ctx.set_detailed_counters(True)
        ctx.reset_accumulation(); ctx.reset_counters()
        ctx.trace_frames(0, seeds, ptamd.PT_ACCUM_LINEAR)
        sc = orc.OracleScene(g["prims"], g["tree"], g["cb"], near_first=near)
        acc = np.zeros((H, W, 4), np.float32)
        tot = dict(paths=0, rays=0, node_visits=0, box_tests=0, tri_tests=0)
        for f in range(4):
            c = orc.trace_frame_linear(sc, gb, f, seeds[f], acc)
            tot = {k: tot[k] + c[k] for k in tot}
        assert np.array_equal(_bits(ctx.linear_sum()), _bits(acc)), near
        assert ctx.counters() == tot
        res[near] = (acc, tot)
    differing = np.any(res[True][0] != res[False][0], axis=-1).mean()
    print(scene, "pixels that differ between the two orders:", differing,
          "visits/ray", res[False][1]["node_visits"] / res[False][1]["rays"], "->", res[True][1]["node_visits"] / res[True][1]["rays"])
    assert differing < 0.02
    # the visit counts are close on these scenes (most visited boxes lie in front of the hit either way)
    assert 0.8 < res[True][1]["node_visits"] / res[False][1]["node_visits"] < 1.05


def test_build_50m_soup_properties(gpu_ctx_factory):
    """LBVH build at 50M triangles (the larger build size of the benchmark), generated on the device.
    Checked on the device with torch over ALL elements: codes ascending with ties in ascending primitive
    order, a permutation, every internal box the union of its children's, parents consistent, leaves follow
    the permutation; codes of a 200k sample against the oracle's arithmetic."""
    import torch
    n = 50_000_000
    ctx = gpu_ctx_factory()
    d = device_soup(n, seed=777)
    ctx.upload_primitives_device(d.data_ptr(), n)
    ctx.build()
    bt = ctx.build_times()
    print("50M build ms", bt)
    codes = torch.from_numpy(ctx.morton().view(np.int64)).to("cuda")
    perm = torch.from_numpy(ctx.permutation()).to("cuda").long()
    ks = codes[perm]
    assert bool(torch.all((ks[:-1] < ks[1:]) | ((ks[:-1] == ks[1:]) & (perm[:-1] < perm[1:]))))
    seen = torch.zeros(n, dtype=torch.bool, device="cuda")
    seen[perm] = True
    assert bool(seen.all())
    del seen, ks
    # scene box and a sample of codes against the oracle's arithmetic
    pos = d.view(n, 20)[:, [0, 1, 2, 6, 7, 8, 12, 13, 14]].reshape(n, 3, 3)
    lo, hi = pos.min(dim=1).values, pos.max(dim=1).values
    bb = ctx.scene_bound()
    assert np.array_equal(bb["min"][0], lo.min(dim=0).values.cpu().numpy()) and np.array_equal(bb["max"][0], hi.max(dim=0).values.cpu().numpy())
    sel = torch.randint(0, n, (200_000,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    sub = np.zeros(len(sel), dtype=orc.PRIM_RAY)
    sub["bound"]["min"] = lo[sel].cpu().numpy()
    sub["bound"]["max"] = hi[sel].cpu().numpy()
    import ctypes as C
    orc.lib().orc_morton_all(sub.ctypes.data_as(C.c_void_p), C.c_int(len(sel)), bb.ctypes.data_as(C.c_void_p))
    assert np.array_equal(sub["morton_code"].view(np.int64), codes[sel].cpu().numpy())
    del codes
    # the tree in the reference's layout: 100M nodes of 48 bytes, checked as 12 int32 / float32 columns
    t = torch.from_numpy(ctx.tree().view(np.int32).reshape(2 * n, 12)).to("cuda")
    left, right, parent, index = t[:n, 0].long(), t[:n, 1].long(), t[:, 2].long(), t[:, 3]
    fb = t.view(torch.float32)
    bmin, bmax = fb[:, 4:7], fb[:, 8:11]
    ok = right < 2 * n                                       # the one node whose right child is the out-of-range leaf 2n
    assert int((~ok).sum()) == 1
    ar = torch.arange(n, device="cuda")
    assert bool(torch.all(parent[left] == ar)) and bool(torch.all(parent[right[ok]] == ar[ok]))
    assert bool(torch.all(bmin[:n][ok] == torch.minimum(bmin[left[ok]], bmin[right[ok]])))
    assert bool(torch.all(bmax[:n][ok] == torch.maximum(bmax[left[ok]], bmax[right[ok]])))
    assert bool(torch.all(index[n:].long() == perm))
    assert bool(torch.all(bmin[n:] == lo[perm])) and bool(torch.all(bmax[n:] == hi[perm]))
    # one leaf per key, every leaf reachable from exactly one parent
    kids = torch.cat([left, right[ok]])
    assert int(torch.unique(kids).numel()) == 2 * n - 1      # all nodes but the root
    assert bt["sort_passes"] == 2
    ctx.close()


def test_render_100m_soup_properties(gpu_ctx_factory):
    """The largest configuration at full size: 100M triangles, 3840x2160, at most 8 rays per path, full-resolution
    codes, deep stack, nearest-child-first traversal.  Properties: every pixel received every frame, sums are
    finite and non-negative, a second run gives the same image and the same counters bit for bit (nothing in the
    result depends on which lane traced which path), the emitters are seen."""
    n, W, H, frames = 100_000_000, 3840, 2160, 2
    ctx = gpu_ctx_factory()
    d = device_soup(n, seed=2024, scale=100.0, emitter_every=16)
    ctx.upload_primitives_device(d.data_ptr(), n)
    cb = soup_cb5(n, eye=CAM["eye"])
    ctx.set_constants(cb)
    ctx.build(ptamd.PT_BUILD_MORTON_FULL)
    print("100M build ms", ctx.build_times())
    ctx.resize(W, H)
    ctx.set_stack_size(32); ctx.set_max_segments(8); ctx.set_traversal_order(ptamd.PT_TRAVERSAL_NEAR_FIRST)
    cam = ptamd.default_camera(**CAM); cam.zfar = 1000.0
    ctx.primary_hits(cam, ptamd.albedo_table([list(cb["mats"][0][m]["kd"][:3]) for m in range(16)]))
    seeds = ptamd.frame_seeds(1, 0, frames)
    runs = []
    for _ in range(2):
        ctx.reset_accumulation(); ctx.reset_counters()
        ctx.trace_frames(0, seeds, ptamd.PT_ACCUM_LINEAR)
        runs.append((ctx.linear_sum(), ctx.counters()))
    a, c = runs[0]
    assert np.array_equal(_bits(a), _bits(runs[1][0])) and c == runs[1][1]
    assert np.all(a[..., 3] == frames) and np.isfinite(a).all() and np.all(a[..., :3] >= 0)
    assert c["paths"] == W * H * frames and 0.5 * c["paths"] < c["rays"] <= 8 * c["paths"]
    assert (a[..., :3].max(axis=-1) > 0).mean() > 0.05      # light reaches a part of the image
    # the same scene through the tree over the Morton codes (another non-parity option): a different tree finds the
    # same nearest hits, so the image is the same except where two triangles are hit at equal distance
    ctx.build(ptamd.PT_BUILD_MORTON_FULL | ptamd.PT_BUILD_TOPOLOGY_MORTON)
    print("100M build ms, tree over the codes", ctx.build_times())
    ctx.set_detailed_counters(True)
    ctx.reset_accumulation(); ctx.reset_counters()
    ctx.trace_frames(0, seeds, ptamd.PT_ACCUM_LINEAR)
    a2, c2 = ctx.linear_sum(), ctx.counters()
    assert np.all(a2[..., 3] == frames) and c2["paths"] == c["paths"]
    differing = np.any(a2 != a, axis=-1).mean()
    print("pixels that differ between the two trees", differing, "visits per ray", c2["node_visits"] / c2["rays"])
    assert differing < 0.01 and abs(c2["rays"] - c["rays"]) < 0.001 * c["rays"]
    ctx.close()
    del d


def test_build_20m_morton_keyed_tree_properties(gpu_ctx_factory):
    """PT_BUILD_TOPOLOGY_MORTON at 20M triangles (about 20k chunks of keys and their border-crossing nodes), checked
    on the device over ALL nodes: n - 1 internal nodes, every node but the root is the child of exactly one node,
    parents consistent, every internal box the union of its children's, leaves follow the sorted order, and the
    radix-tree property: the keys below a node share a longer code prefix with each other than with the keys next
    to the node's range."""
    import torch
    n = 20_000_000
    ctx = gpu_ctx_factory()
    d = device_soup(n, seed=99)
    ctx.upload_primitives_device(d.data_ptr(), n)
    ctx.build(ptamd.PT_BUILD_MORTON_FULL | ptamd.PT_BUILD_TOPOLOGY_MORTON)
    print("20M build ms", ctx.build_times())
    perm = torch.from_numpy(ctx.permutation()).to("cuda").long()
    codes = torch.from_numpy(ctx.morton().view(np.int64)).to("cuda")[perm]          # sorted, below 2^63
    t = torch.from_numpy(ctx.tree().view(np.int32).reshape(2 * n, 12)).to("cuda")
    left, right, parent, index = t[:n - 1, 0].long(), t[:n - 1, 1].long(), t[:, 2].long(), t[:, 3]
    fb = t.view(torch.float32)
    bmin, bmax = fb[:, 4:7], fb[:, 8:11]
    ar = torch.arange(n - 1, device="cuda")
    assert bool(torch.all(parent[left] == ar)) and bool(torch.all(parent[right] == ar))
    assert bool(torch.all(bmin[:n - 1] == torch.minimum(bmin[left], bmin[right])))
    assert bool(torch.all(bmax[:n - 1] == torch.maximum(bmax[left], bmax[right])))
    kids = torch.cat([left, right])
    assert int(torch.unique(kids).numel()) == 2 * n - 2 and int(kids.min()) == 1      # all nodes but the root, once each
    assert bool(torch.all(t[n - 1] == 0))                                            # the unused slot
    assert bool(torch.all(index[n:].long() == perm))
    pos = d.view(n, 20)[:, [0, 1, 2, 6, 7, 8, 12, 13, 14]].reshape(n, 3, 3)
    assert bool(torch.all(bmin[n:] == pos.min(dim=1).values[perm])) and bool(torch.all(bmax[n:] == pos.max(dim=1).values[perm]))
    # children are named by the split position: left = gamma (or leaf n + gamma), right = gamma + 1 (or leaf n + gamma + 1)
    gamma = torch.where(left >= n, left - n, left)
    assert bool(torch.all(torch.where(right >= n, right - n, right) == gamma + 1))
    # key range of every node from its leftmost / rightmost descendants, top-down over the (at most 64) levels
    lo = torch.full((2 * n,), -1, dtype=torch.int64, device="cuda"); hi = lo.clone()
    lo[0], hi[0] = 0, n - 1
    frontier = torch.zeros(1, dtype=torch.int64, device="cuda")
    levels = 0
    while frontier.numel():
        l, r = left[frontier], right[frontier]
        g = gamma[frontier]
        lo[l], hi[l] = lo[frontier], g
        lo[r], hi[r] = g + 1, hi[frontier]
        nxt = torch.cat([l, r])
        frontier = nxt[nxt < n - 1]
        levels += 1
    assert levels <= 64
    assert bool(torch.all(lo[n:] == torch.arange(n, device="cuda"))) and bool(torch.all(hi[n:] == lo[n:]))
    a, b = lo[:n - 1], hi[:n - 1]

    def msb(x):                  # index of the highest set bit of positive int64 values
        r = torch.zeros_like(x)
        for sh in (32, 16, 8, 4, 2, 1):
            m = (x >> sh) != 0
            r += torch.where(m, sh, 0)
            x = torch.where(m, x >> sh, x)
        return r

    def prefix(i, j):            # common prefix of the (code, position) keys i != j; codes are below 2^63
        x = codes[i] ^ codes[j]
        return torch.where(x != 0, 63 - msb(x.clamp(min=1)), 64 + 31 - msb((i ^ j).clamp(min=1)))
    inner = prefix(a, b)
    before = torch.where(a > 0, prefix(a, (a - 1).clamp(min=0)), torch.full_like(a, -1))
    after = torch.where(b < n - 1, prefix(b, (b + 1).clamp(max=n - 1)), torch.full_like(a, -1))
    assert bool(torch.all(inner > before)) and bool(torch.all(inner > after))
    # and the split is where the prefix of the range ends: the two keys around gamma differ in the first bit after it
    assert bool(torch.all(prefix(gamma, gamma + 1) == inner))
    ctx.close()


@pytest.mark.parametrize("pair_lines", [True, False])
@pytest.mark.parametrize("near_first", [False, True])
@pytest.mark.parametrize("full", [True, False])
def test_morton_keyed_topology_against_the_oracle(gpu_ctx_factory, monkeypatch, full, near_first, pair_lines):
    """PT_BUILD_TOPOLOGY_MORTON (NON-PARITY: the radix tree over the sorted codes instead of the reference's tree over
    the leaf index), n = 200k, with full-resolution codes and with the reference's 15-bit codes (thousands of equal
    codes, told apart by position).  Tree, G-buffer, linear sums and counters equal the oracle's restatement of
    the same option bit for bit; the tree is a different one than the reference's, the nearest hits are the same."""
    n, W, H = 200_000, 200, 150
    # child lines are used from 24 MiB of plain records on: forced on (or off) for this small tree
    monkeypatch.setenv("PTAMD_PAIR_MIN_MIB", "0")
    if not pair_lines:
        monkeypatch.setenv("PTAMD_PAIR_MAX_MIB", "0")
    if not full:
        # two stack entries per lane in LDS, the rest of every lane's stack in global memory: the path that trees
        # higher than the LDS part take
        monkeypatch.setenv("PTAMD_GEN_LDS_ENTRIES", "2")
    prims = make_soup(n, seed=777, dtype=ptamd.PRIM_RENDER)
    prims["vertices"]["pos"] *= np.float32(100.0)
    prims["matid"] = np.where(np.arange(n) % 16 == 0, 4, np.arange(n) % 4)
    flags = ptamd.PT_BUILD_TOPOLOGY_MORTON | (ptamd.PT_BUILD_MORTON_FULL if full else 0)
    ctx = gpu_ctx_factory()
    cb, alb = _soup_setup(ctx, prims, W, H, flags, near_first)
    ob = orc.build_codes_topology(prims.view(orc.PRIM_RENDER), full)
    assert np.array_equal(ctx.morton(), ob["rays"]["morton_code"])
    assert np.array_equal(ctx.permutation(), ob["sorted_rays"]["index"])
    got_tree = ctx.tree()
    for f in ("left", "right", "parent", "index"):
        assert np.array_equal(got_tree[f], ob["tree"][f][:2 * n]), f
    assert got_tree.tobytes() == ob["tree"][:2 * n].tobytes()
    height = orc.tree_height(ob["tree"], n)
    print("tree height", height)
    ocam = orc.default_camera(**CAM); ocam.zfar = 1000.0
    sc = orc.OracleScene(prims.view(orc.PRIM_RENDER), ob["tree"], cb, stack_size=120, max_segments=8, near_first=near_first)
    gb = orc.primary(orc.OracleScene(prims.view(orc.PRIM_RENDER), ob["tree"], cb, stack_size=120), ocam, W, H, alb)
    d, nn, w = ctx.download_gbuffer()
    assert np.array_equal(d, gb[0]) and np.array_equal(_bits(nn), _bits(gb[1])) and np.array_equal(_bits(w), _bits(gb[2]))
    ctx.set_detailed_counters(True)
    seeds = ptamd.frame_seeds(1, 0, 2)
    ctx.trace_frames(0, seeds, ptamd.PT_ACCUM_LINEAR)
    acc = np.zeros((H, W, 4), np.float32)
    tot = dict(paths=0, rays=0, node_visits=0, box_tests=0, tri_tests=0)
    for f in range(2):
        c = orc.trace_frame_linear(sc, gb, f, seeds[f], acc)
        tot = {k: tot[k] + c[k] for k in tot}
    assert np.array_equal(_bits(ctx.linear_sum()), _bits(acc))
    assert ctx.counters() == tot
    # against the reference's tree over the same order: fewer node visits
    ob_ref = orc.build_full_codes(prims.view(orc.PRIM_RENDER)) if full else orc.build(prims.view(orc.PRIM_RENDER))
    sc_ref = orc.OracleScene(prims.view(orc.PRIM_RENDER), ob_ref["tree"], cb, stack_size=120, max_segments=8, near_first=near_first)
    acc_ref = np.zeros((H, W, 4), np.float32)
    c_ref = orc.trace_frame_linear(sc_ref, gb, 0, seeds[0], acc_ref)
    acc0 = np.zeros((H, W, 4), np.float32)
    c0 = orc.trace_frame_linear(sc, gb, 0, seeds[0], acc0)
    print("full", full, "near_first", near_first, "visits/ray", c0["node_visits"] / c0["rays"], "index-keyed", c_ref["node_visits"] / c_ref["rays"])
    assert c0["node_visits"] < 0.8 * c_ref["node_visits"]
    assert np.any(acc0 != acc_ref, axis=-1).mean() < 0.01       # other tree, same nearest hits (ties aside)


@pytest.mark.parametrize("scene", SCENES)
def test_morton_keyed_topology_on_the_reference_scenes(gpu_ctx_factory, golden, scene):
    """The same option on the reference's scenes (15-bit codes, many ties; node records in global memory)."""
    W, H = 160, 136
    g = golden[scene]
    n = len(g["prims"])
    ctx = gpu_ctx_factory()
    ctx.upload_primitives(g["prims"]); ctx.set_constants(g["cb"]); ctx.build(ptamd.PT_BUILD_TOPOLOGY_MORTON); ctx.resize(W, H)
    ob = orc.build_codes_topology(g["prims"])
    _same_tree(ctx.tree(), ob["tree"][:2 * n])
    alb = ptamd.albedo_table(g["mat_table"][:, 0:3], True)
    ctx.primary_hits(ptamd.default_camera(), alb)
    gb = ctx.download_gbuffer()
    sc = orc.OracleScene(g["prims"], ob["tree"], g["cb"], stack_size=120)
    ogb = orc.primary(sc, orc.default_camera(), W, H, alb)
    assert all(np.array_equal(_bits(a), _bits(b)) for a, b in zip(gb, ogb))
    seeds = ptamd.frame_seeds(1, 0, 3)
    ctx.set_detailed_counters(True)
    ctx.trace_frames(0, seeds, ptamd.PT_ACCUM_LINEAR)
    acc = np.zeros((H, W, 4), np.float32)
    tot = dict(paths=0, rays=0, node_visits=0, box_tests=0, tri_tests=0)
    for f in range(3):
        c = orc.trace_frame_linear(sc, gb, f, seeds[f], acc)
        tot = {k: tot[k] + c[k] for k in tot}
    assert np.array_equal(_bits(ctx.linear_sum()), _bits(acc))
    assert ctx.counters() == tot
    # strict-parity stack and the option exclude each other; rebuilding without the flag gives the reference tree again
    ctx.set_stack_size(15)
    with pytest.raises(ptamd.PtError):
        ctx.trace_frames(3, ptamd.frame_seeds(1, 3, 1), ptamd.PT_ACCUM_LINEAR)
    ctx.set_stack_size(0)
    ctx.build()
    _same_tree(ctx.tree(), g["tree"][:2 * n])


def _single_bit_scene(bits, duplicates):
    """Triangles whose full-resolution codes are 0 (`duplicates` times) and single bits: their radix tree is a
    comb, one level per distinct code, plus the levels that tell the equal codes apart by position."""
    cells = 2097151.0
    cent = [(0.5, 0.5, 0.5)] * duplicates
    for b in range(bits):
        axis, j = 2 - (b % 3), b // 3             # bit b of the code: axis x is the highest of each triple
        c = [0.5, 0.5, 0.5]
        c[axis] = float(2 ** j) + 0.5
        cent.append(tuple(c))
    n = len(cent) + 1
    prims = np.zeros(n, dtype=ptamd.PRIM_RENDER)
    c = np.array(cent, dtype=np.float32)
    prims["vertices"]["pos"][:-1, 0] = c - np.float32(0.25)
    prims["vertices"]["pos"][:-1, 1] = c
    prims["vertices"]["pos"][:-1, 2] = c + np.float32(0.25)
    prims["vertices"]["pos"][-1, 1] = cells + 0.25        # spans the scene box: [-0.25, cells + 0.25] would shift the grid,
    prims["vertices"]["pos"][-1, 0] = -0.25               # so the box is made symmetric around the cell grid instead
    prims["vertices"]["pos"][-1, 2] = -0.25
    prims["vertices"]["normal"][:, :, 1] = 1.0
    prims["pad"] = -1
    return prims


def test_morton_keyed_tree_height_limit(gpu_ctx_factory):
    """Codes that differ in ever lower bits give a comb-shaped radix tree.  Up to the supported height the library
    equals the oracle; a higher tree is refused with PT_ERR_UNSUPPORTED (the traversal stacks are sized for 64
    levels) and the context stays usable."""
    flags = ptamd.PT_BUILD_MORTON_FULL | ptamd.PT_BUILD_TOPOLOGY_MORTON
    ctx = gpu_ctx_factory()
    heights = {}
    for bits, dup in ((40, 3), (61, 16)):
        prims = _single_bit_scene(bits, dup)
        n = len(prims)
        ob = orc.build_codes_topology(prims.view(orc.PRIM_RENDER), True)
        heights[bits] = orc.tree_height(ob["tree"], n)
        ctx.upload_primitives(prims)
        if heights[bits] <= 64:
            ctx.build(flags)
            _same_tree(ctx.tree(), ob["tree"][:2 * n])
        else:
            with pytest.raises(ptamd.PtError) as e:
                ctx.build(flags)
            assert "higher than" in str(e.value)
            ctx.build(ptamd.PT_BUILD_MORTON_FULL)                 # the reference's tree still builds
            _same_tree(ctx.tree(), orc.build_full_codes(prims.view(orc.PRIM_RENDER))["tree"][:2 * n])
    print("tree heights", heights)
    assert heights[40] <= 64 < heights[61]


def test_morton_keyed_tree_of_equal_codes(gpu_ctx_factory):
    """All codes equal (n copies of one triangle): the keys are told apart by position only, which gives the
    balanced tree over the positions; 3000 keys span three chunks of the refit.  Tree and a rendered frame equal
    the oracle's."""
    n, W, H = 3000, 64, 48
    prims = np.zeros(n, dtype=ptamd.PRIM_RENDER)
    prims["vertices"]["pos"][:, 0] = (40.0, 40.0, 50.0)
    prims["vertices"]["pos"][:, 1] = (60.0, 40.0, 50.0)
    prims["vertices"]["pos"][:, 2] = (50.0, 60.0, 50.0)
    prims["vertices"]["normal"][:, :, 2] = 1.0
    prims["pad"] = -1
    ctx = gpu_ctx_factory()
    cb, alb = _soup_setup(ctx, prims, W, H, ptamd.PT_BUILD_TOPOLOGY_MORTON)
    ob = orc.build_codes_topology(prims.view(orc.PRIM_RENDER))
    assert len(np.unique(ob["rays"]["morton_code"])) == 1
    _same_tree(ctx.tree(), ob["tree"][:2 * n])
    assert orc.tree_height(ob["tree"], n) == 12
    ocam = orc.default_camera(**CAM); ocam.zfar = 1000.0
    sc = orc.OracleScene(prims.view(orc.PRIM_RENDER), ob["tree"], cb, stack_size=120, max_segments=8)
    gb = orc.primary(sc, ocam, W, H, alb)
    d, nn, w = ctx.download_gbuffer()
    assert np.array_equal(d, gb[0]) and np.array_equal(_bits(nn), _bits(gb[1])) and np.array_equal(_bits(w), _bits(gb[2]))
    assert (gb[1][..., 3].view(np.int32) > 0).any()                  # the triangle is in view
    ctx.set_detailed_counters(True)
    seeds = ptamd.frame_seeds(1, 0, 1)
    ctx.trace_frames(0, seeds, ptamd.PT_ACCUM_LINEAR)
    acc = np.zeros((H, W, 4), np.float32)
    c = orc.trace_frame_linear(sc, gb, 0, seeds[0], acc)
    assert np.array_equal(_bits(ctx.linear_sum()), _bits(acc)) and ctx.counters() == c


@pytest.mark.parametrize("n", [2, 3, 5, 64, 1025])
def test_morton_keyed_tiny_trees(gpu_ctx_factory, n):
    """The smallest trees of PT_BUILD_TOPOLOGY_MORTON (two keys: one internal node; 1025 keys: the root crosses the
    border of the two chunks of the refit): tree and one traced frame against the oracle."""
    W, H = 48, 32
    rng = np.random.default_rng(n)
    prims = make_soup(n, seed=n, dtype=ptamd.PRIM_RENDER)
    prims["vertices"]["pos"] = (rng.random((n, 3, 3), dtype=np.float32) * np.float32(60) + np.float32(20))
    prims["matid"] = np.arange(n) % 4
    ctx = gpu_ctx_factory()
    cb, alb = _soup_setup(ctx, prims, W, H, ptamd.PT_BUILD_TOPOLOGY_MORTON | ptamd.PT_BUILD_MORTON_FULL)
    ob = orc.build_codes_topology(prims.view(orc.PRIM_RENDER), True)
    _same_tree(ctx.tree(), ob["tree"][:2 * n])
    ocam = orc.default_camera(**CAM); ocam.zfar = 1000.0
    sc = orc.OracleScene(prims.view(orc.PRIM_RENDER), ob["tree"], cb, stack_size=120, max_segments=8)
    gb = orc.primary(sc, ocam, W, H, alb)
    d, nn, w = ctx.download_gbuffer()
    assert np.array_equal(d, gb[0]) and np.array_equal(_bits(nn), _bits(gb[1])) and np.array_equal(_bits(w), _bits(gb[2]))
    ctx.set_detailed_counters(True)
    seeds = ptamd.frame_seeds(1, 0, 2)
    ctx.trace_frames(0, seeds, ptamd.PT_ACCUM_LINEAR)
    acc = np.zeros((H, W, 4), np.float32)
    tot = dict(paths=0, rays=0, node_visits=0, box_tests=0, tri_tests=0)
    for f in range(2):
        c = orc.trace_frame_linear(sc, gb, f, seeds[f], acc)
        tot = {k: tot[k] + c[k] for k in tot}
    assert np.array_equal(_bits(ctx.linear_sum()), _bits(acc)) and ctx.counters() == tot
    # one primitive has no internal node: refused
    ctx.upload_primitives(prims[:1])
    with pytest.raises(ptamd.PtError):
        ctx.build(ptamd.PT_BUILD_TOPOLOGY_MORTON)
