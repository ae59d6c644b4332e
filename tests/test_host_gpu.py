"""GPU: the RayTracer host class (same method names as the reference's) drives the C ABI and
gives the oracle's results; the headless driver writes a snapshot."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import ptamd
from oracle import orc
from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
LIB = os.path.join(ROOT, "path-tracer_amd", "libpthost.so")


def test_raytracer_class_matches_oracle(golden, monkeypatch):
    monkeypatch.chdir(os.path.join(GOLDEN, "scenes"))
    L = C.CDLL(LIB)
    L.pth_tracer_new.restype = C.c_void_p
    L.pth_tracer_error.restype = C.c_char_p
    for f in ("pth_tracer_free", "pth_tracer_error"):
        getattr(L, f).argtypes = [C.c_void_p]
    L.pth_tracer_load.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    L.pth_tracer_run.argtypes = [C.c_void_p, C.c_int]
    L.pth_tracer_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.pth_tracer_members.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    W, H, frames = 144, 120, 3
    r = L.pth_tracer_new(0)
    try:
        g = golden["scene04"]
        n = L.pth_tracer_load(r, b"scene04.obj", W, H)
        assert n == len(g["prims"]), L.pth_tracer_error(r)
        # public members filled from the device build: bb_, ray_primitives (sorted), tree
        bb = np.zeros(1, dtype=orc.BOUND)
        rays = np.zeros(n, dtype=orc.PRIM_RAY)
        tree = np.zeros(2 * n, dtype=orc.TREENODE)
        assert L.pth_tracer_members(r, bb.ctypes.data_as(C.c_void_p), rays.ctypes.data_as(C.c_void_p),
                                    tree.ctypes.data_as(C.c_void_p)) == 0
        ob = orc.build(g["prims"])
        assert np.array_equal(bb["min"], ob["bb"]["min"]) and np.array_equal(bb["max"], ob["bb"]["max"])
        assert np.array_equal(rays["morton_code"], ob["sorted_rays"]["morton_code"])
        assert np.array_equal(rays["index"], ob["sorted_rays"]["index"])
        assert tree.tobytes() == ob["tree"][:2 * n].tobytes()
        # run(): first call builds, every call traces one frame with the fixed-seed convention
        assert L.pth_tracer_run(r, frames) == frames, L.pth_tracer_error(r)
        out = np.zeros((H, W, 4), np.float32)
        disp = np.zeros((H, W, 4), np.uint8)
        assert L.pth_tracer_read(r, out.ctypes.data_as(C.c_void_p), disp.ctypes.data_as(C.c_void_p)) == 0
        sc = orc.OracleScene(g["prims"], ob["tree"], g["cb"])          # golden cb = CS.cpp material mapping
        kd = [list(m[0:3]) for m in g["mat_table"]] + [[0, 0, 0]] * (16 - len(g["mat_table"]))
        gb = orc.primary(sc, orc.default_camera(), W, H, orc.albedo_table(g["cb"], True, kd_override=kd))
        old = np.zeros((H, W, 4), np.float32)
        seeds = orc.frame_seeds(1, 0, frames)
        for f in range(frames):
            old, outu, _, _ = orc.trace_frame(sc, gb, f, seeds[f], old)
        assert out.tobytes() == old.tobytes()
        assert np.array_equal(disp, outu)
        # run() queues frames and traces them in batches (here 32 + 8): same images as one launch per frame
        more = 40
        assert L.pth_tracer_run(r, more) == frames + more, L.pth_tracer_error(r)
        assert L.pth_tracer_read(r, out.ctypes.data_as(C.c_void_p), disp.ctypes.data_as(C.c_void_p)) == 0
        seeds = orc.frame_seeds(1, 0, frames + more)
        for f in range(frames, frames + more):
            old, outu, _, _ = orc.trace_frame(sc, gb, f, seeds[f], old)
        assert out.tobytes() == old.tobytes()
        assert np.array_equal(disp, outu)
    finally:
        L.pth_tracer_free(r)


def _tracer_lib():
    L = C.CDLL(LIB)
    L.pth_tracer_new.restype = C.c_void_p
    L.pth_tracer_error.restype = C.c_char_p
    for f in ("pth_tracer_free", "pth_tracer_error", "pth_tracer_frame_count"):
        getattr(L, f).argtypes = [C.c_void_p]
    L.pth_tracer_load.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    L.pth_tracer_run.argtypes = [C.c_void_p, C.c_int]
    L.pth_tracer_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.pth_tracer_set_frame_count.argtypes = [C.c_void_p, C.c_int]
    L.pth_tracer_resize.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.pth_tracer_set_tiles.argtypes = [C.c_void_p, C.c_int, C.c_int]
    return L


def _read(L, r, W, H):
    out = np.zeros((H, W, 4), np.float32)
    disp = np.zeros((H, W, 4), np.uint8)
    assert L.pth_tracer_read(r, out.ctypes.data_as(C.c_void_p), disp.ctypes.data_as(C.c_void_p)) == 0, L.pth_tracer_error(r)
    return out, disp


def test_raytracer_resize_acc_toggle_and_tiles(golden, monkeypatch):
    """The interactive scaffold of the reference on the facade: `resize` recreates the images and resets
    frame_count (ray_tracer.cpp:188-261); the application's ACC toggle writes frame_count = 0 between frames
    (CS.cpp:524-527), after which the next frame is blended with weight 0 for the old image -- no clear; the tile
    loop (ray_tracer.cpp:298-332) with tiles_x * tiles_y dispatches per frame gives the image of one dispatch."""
    monkeypatch.chdir(os.path.join(GOLDEN, "scenes"))
    L = _tracer_lib()
    g = golden["scene01"]
    W, H = 112, 96
    ob = orc.build(g["prims"])
    sc = orc.OracleScene(g["prims"], ob["tree"], g["cb"])
    kd = [list(m[0:3]) for m in g["mat_table"]] + [[0, 0, 0]] * (16 - len(g["mat_table"]))
    alb = orc.albedo_table(g["cb"], True, kd_override=kd)
    r = L.pth_tracer_new(0)
    try:
        assert L.pth_tracer_load(r, b"scene01.obj", 64, 48) == len(g["prims"]), L.pth_tracer_error(r)
        assert L.pth_tracer_run(r, 5) == 5
        # resize: counter back to 0, images recreated (cleared), the G-buffer is rendered again at the new size
        assert L.pth_tracer_resize(r, W, H) == 0 and L.pth_tracer_frame_count(r) == 0
        out, disp = _read(L, r, W, H)
        assert not out.any() and not disp.any()
        gb = orc.primary(sc, orc.default_camera(), W, H, alb)
        seeds = orc.frame_seeds(1, 0, 6)
        assert L.pth_tracer_run(r, 4) == 4, L.pth_tracer_error(r)
        old = np.zeros((H, W, 4), np.float32)
        for f in range(4):
            old = orc.trace_frame(sc, gb, f, seeds[f], old)[0]
        assert _read(L, r, W, H)[0].tobytes() == old.tobytes()
        # ACC toggle: frame_count = 0, then one frame: framecount 0 and seed 0 again, blended over the old image with weight 0
        L.pth_tracer_set_frame_count(r, 0)
        assert L.pth_tracer_run(r, 1) == 1
        old, outu, _, _ = orc.trace_frame(sc, gb, 0, seeds[0], old)
        out, disp = _read(L, r, W, H)
        assert out.tobytes() == old.tobytes() and np.array_equal(disp, outu)
        # tiles: 3 x 2 dispatches per frame (112 / 3 = 37: a fourth, one-pixel column of tiles, as in the reference's loop)
        L.pth_tracer_set_tiles(r, 3, 2)
        assert L.pth_tracer_run(r, 2) == 3, L.pth_tracer_error(r)
        for f in (1, 2):
            old, outu, _, _ = orc.trace_frame(sc, gb, f, seeds[f], old)
        out, disp = _read(L, r, W, H)
        assert out.tobytes() == old.tobytes() and np.array_equal(disp, outu)
        # the reference's literal seeding: a new seed per DISPATCH of the tile loop (ray_tracer.cpp:315-318).  Frame 3 is
        # dispatches 3 * 8 .. 3 * 8 + 7 (4 columns x 2 rows, columns outermost); a tile's pixels are those of a whole frame
        # traced with the tile's seed
        L.pth_tracer_set_tile_seed_per_dispatch.argtypes = [C.c_void_p, C.c_int]
        L.pth_tracer_set_tile_seed_per_dispatch(r, 1)
        assert L.pth_tracer_run(r, 1) == 4, L.pth_tracer_error(r)
        tw, th = W // 3, H // 2
        new, t = old.copy(), 0
        for w in range(0, W, tw):
            for h in range(0, H, th):
                full = orc.trace_frame(sc, gb, 3, orc.frame_seeds(1, 3 * 8 + t, 1)[0], old)[0]
                new[h:h + th, w:w + tw] = full[h:h + th, w:w + tw]
                t += 1
        assert t == 8 and _read(L, r, W, H)[0].tobytes() == new.tobytes()
    finally:
        L.pth_tracer_free(r)


def test_raytracer_class_build_flags(golden, monkeypatch):
    """`build_flags` reaches pt_build: with the tree over the Morton codes the facade's public tree is the oracle's
    for that option and the frames are the oracle's over that tree; back to the default, the reference's tree."""
    monkeypatch.chdir(os.path.join(GOLDEN, "scenes"))
    L = _tracer_lib()
    L.pth_tracer_set_build_flags.argtypes = [C.c_void_p, C.c_uint]
    L.pth_tracer_rebuild.argtypes = [C.c_void_p]
    L.pth_tracer_members.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    W, H, frames = 96, 80, 2
    g = golden["scene04"]
    n = len(g["prims"])
    r = L.pth_tracer_new(0)
    try:
        L.pth_tracer_set_build_flags(r, ptamd.PT_BUILD_TOPOLOGY_MORTON)
        assert L.pth_tracer_load(r, b"scene04.obj", W, H) == n, L.pth_tracer_error(r)
        tree = np.zeros(2 * n, dtype=orc.TREENODE)
        assert L.pth_tracer_members(r, None, None, tree.ctypes.data_as(C.c_void_p)) == 0
        ob = orc.build_codes_topology(g["prims"])
        for f in ("left", "right", "parent", "index"):
            assert np.array_equal(tree[f], ob["tree"][:2 * n][f]), f
        assert np.array_equal(tree["bound"]["min"], ob["tree"][:2 * n]["bound"]["min"])
        assert np.array_equal(tree["bound"]["max"], ob["tree"][:2 * n]["bound"]["max"])
        assert L.pth_tracer_run(r, frames) == frames, L.pth_tracer_error(r)
        out, disp = _read(L, r, W, H)
        sc = orc.OracleScene(g["prims"], ob["tree"], g["cb"], stack_size=120)
        kd = [list(m[0:3]) for m in g["mat_table"]] + [[0, 0, 0]] * (16 - len(g["mat_table"]))
        gb = orc.primary(sc, orc.default_camera(), W, H, orc.albedo_table(g["cb"], True, kd_override=kd))
        old = np.zeros((H, W, 4), np.float32)
        seeds = orc.frame_seeds(1, 0, frames)
        for f in range(frames):
            old, outu, _, _ = orc.trace_frame(sc, gb, f, seeds[f], old)
        assert out.tobytes() == old.tobytes()
        assert np.array_equal(disp, outu)
        # the default again: the reference's tree
        L.pth_tracer_set_build_flags(r, ptamd.PT_BUILD_DEFAULT)
        assert L.pth_tracer_rebuild(r) == 0, L.pth_tracer_error(r)
        assert L.pth_tracer_members(r, None, None, tree.ctypes.data_as(C.c_void_p)) == 0
        assert np.array_equal(tree["left"], g["tree"][:2 * n]["left"])
        assert np.array_equal(tree["parent"], g["tree"][:2 * n]["parent"])
    finally:
        L.pth_tracer_free(r)


def test_headless_driver_writes_snapshot(tmp_path):
    exe = os.path.join(ROOT, "path-tracer_amd", "host", "pt_headless")
    out = tmp_path / "snap.bmp"
    res = subprocess.run([exe, "scene01.obj", "160", "128", "8", str(out)], cwd=os.path.join(GOLDEN, "scenes"),
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    assert "862 triangles" in res.stdout
    data = out.read_bytes()
    assert data[:2] == b"BM" and len(data) == 54 + 160 * 3 * 128
    assert max(data[54:]) > 100                          # the light is visible
    # scheduled snapshots under the reference's file names (CS.cpp:446-479,507-514)
    snaps = tmp_path / "snaps"
    snaps.mkdir()
    res = subprocess.run([exe, "scene01.obj", "48", "40", "100", str(out), str(snaps)], cwd=os.path.join(GOLDEN, "scenes"),
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    assert sorted(p.name for p in snaps.iterdir()) == ["48X40frame_100.bmp", "48X40frame_50.bmp"]
    assert (snaps / "48X40frame_100.bmp").read_bytes() == out.read_bytes()
    bad = subprocess.run([exe, "missing.obj"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 1 and "cannot open" in bad.stderr
