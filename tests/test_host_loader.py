"""CPU: the host-side OBJ/MTL loader + Scene flattening (own code) against the fixtures made with
the reference's own parser (oracle/_ref/tinyobj_dump) and the oracle's flatten restatement."""
import ctypes as C
import os

import numpy as np
import pytest

import ptamd
from conftest import GOLDEN, ROOT, SCENES

LIB = os.path.join(ROOT, "path-tracer_amd", "libpthost.so")


@pytest.fixture(scope="module")
def host():
    L = C.CDLL(LIB)
    L.pth_scene_load.restype = C.c_void_p
    L.pth_scene_load.argtypes = [C.c_char_p]
    for f in ("pth_scene_free", "pth_scene_triangle_count", "pth_scene_shape_count", "pth_scene_material_count",
              "pth_scene_flatten", "pth_scene_materials"):
        getattr(L, f).argtypes = [C.c_void_p] + ([C.c_void_p] if f in ("pth_scene_flatten", "pth_scene_materials") else [])
    L.pth_parse_real.restype = C.c_float
    L.pth_parse_real.argtypes = [C.c_char_p]
    return L


@pytest.mark.parametrize("scene", SCENES)
def test_loader_matches_reference_parser_fixtures(host, golden, scene, monkeypatch):
    monkeypatch.chdir(os.path.join(GOLDEN, "scenes"))           # `mtllib` resolves against the cwd
    s = host.pth_scene_load((scene + ".obj").encode())
    assert s
    g = golden[scene]
    try:
        n = host.pth_scene_triangle_count(s)
        assert n == len(g["prims"])
        assert host.pth_scene_shape_count(s) == g["parse_counts"][2]
        assert host.pth_scene_material_count(s) == g["parse_counts"][3]
        prims = np.zeros(n, dtype=ptamd.PRIM_RENDER)
        assert host.pth_scene_flatten(s, prims.ctypes.data_as(C.c_void_p)) == n
        assert prims.tobytes() == g["prims"].tobytes()           # every float bit, matid and pad
        mats = np.zeros((g["parse_counts"][3], 12), dtype=np.float32)
        host.pth_scene_materials(s, mats.ctypes.data_as(C.c_void_p))
        assert mats.tobytes() == g["mat_table"].astype(np.float32).tobytes()
    finally:
        host.pth_scene_free(s)


def test_mtl_found_next_to_obj_from_another_cwd(host, golden, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    s = host.pth_scene_load(os.path.join(GOLDEN, "scenes", "scene01.obj").encode())
    assert s
    assert host.pth_scene_material_count(s) == 7
    host.pth_scene_free(s)
    assert not host.pth_scene_load(b"does_not_exist.obj")


def test_number_reader_follows_the_digit_accumulation_rule(host):
    # the parser accumulates decimal digits in double and narrows to float
    def expect(text):
        sign = -1.0 if text.startswith("-") else 1.0
        body = text.lstrip("+-")
        mant, _, ex = body.lower().partition("e")
        ip, _, fp = mant.partition(".")
        m = 0.0
        for ch in ip:
            m = m * 10 + int(ch)
        lut = [1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001]
        for k, ch in enumerate(fp, start=1):
            m += int(ch) * (lut[k] if k < 8 else 10.0 ** (-k))
        if ex:
            e = int(ex)
            if e:
                import math
                m = math.ldexp(m * 5.0 ** e, e)
        return np.float32(sign * m)
    for text in ["0", "-0", "1", "-4.2140", "6.5840", "0.1", "123.456789", "3.1417e+2", "-0.0E-3", "11e2", "1e-3", "0.00001234"]:
        assert np.float32(host.pth_parse_real(text.encode())).tobytes() == expect(text).tobytes(), text
    assert host.pth_parse_real(b"abc") == 0.0


def test_polygon_and_material_rules(host, tmp_path, monkeypatch):
    """Hand-made file: a concave pentagon (ear clipping), negative indices, a face without normals
    inside a smoothing group, `usemtl` of an unknown name (default material), `d` winning over `Tr`."""
    monkeypatch.chdir(tmp_path)
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 0.1 0.2 0.3\nTr 0.25\nd 0.5\nTr 0.75\nnewmtl b\nKe 2 2 2\nNs 40\nNi 1.33\n")
    (tmp_path / "t.obj").write_text(
        "mtllib m.mtl\nv 0 0 0\nv 2 0 0\nv 2 2 0\nv 1 0.5 0\nv 0 2 0\nvn 0 0 1\n"
        "g poly\nusemtl a\ns 1\nf 1 2 3 4 5\n"
        "usemtl nosuch\nf -5//1 -4//1 -3//1\n"
        "g second\nusemtl b\ns off\nf 1/1 2/1 3/1\n")
    s = host.pth_scene_load(b"t.obj")
    assert s
    try:
        n = host.pth_scene_triangle_count(s)
        assert n == 3 + 1 + 1 and host.pth_scene_shape_count(s) == 2 and host.pth_scene_material_count(s) == 3
        prims = np.zeros(n, dtype=ptamd.PRIM_RENDER)
        host.pth_scene_flatten(s, prims.ctypes.data_as(C.c_void_p))
        assert list(prims["matid"]) == [0, 0, 0, 2, 1]                 # unknown name -> default (last) material
        assert np.all(prims["pad"] == -1)
        # total area of the pentagon's triangles = polygon area (2.5, shoelace); x is mirrored
        p = prims["vertices"]["pos"][:3].astype(np.float64)
        area = 0.5 * np.abs(np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])[:, 2]).sum()
        assert area == pytest.approx(2.5)
        assert p[..., 0].min() == -2.0 and p[..., 0].max() == 0.0
        # no vn on the pentagon corners but a smoothing group: averaged face normals; the face normal is (c-a) x (b-a), i.e. (0,0,-1) here
        assert np.allclose(prims["vertices"]["normal"][0], [[0, 0, -1]] * 3)
        mats = np.zeros((3, 12), dtype=np.float32)
        host.pth_scene_materials(s, mats.ctypes.data_as(C.c_void_p))
        assert mats[0, 11] == np.float32(0.5)                           # dissolve: `d 0.5` wins
        assert mats[1, 9] == 40 and mats[1, 10] == np.float32(1.33) and mats[1, 6] == 2
        assert not mats[2].any()                                        # default material: all zero
    finally:
        host.pth_scene_free(s)


@pytest.mark.parametrize("face", ["f 9999 1 2", "f 1 2 5", "f -4 1 2", "f 1//3 2//1 3//1", "f 1//-2 2//1 3//1", "f 0 1 2"])
def test_face_indices_outside_the_arrays_are_errors(host, tmp_path, monkeypatch, face):
    """Too large, too negative and zero indices fail the load instead of reading outside the arrays: a vertex must
    exist when the face's group is closed (that is when the face is triangulated), a normal by the end of the file."""
    monkeypatch.chdir(tmp_path)
    (tmp_path / "bad.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\n" + face + "\nv 5 5 5\nvn 0 1 0\n")
    assert not host.pth_scene_load(b"bad.obj")
    (tmp_path / "good.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf -3//-1 -2//1 3//1\n")
    s = host.pth_scene_load(b"good.obj")
    assert s and host.pth_scene_triangle_count(s) == 1
    host.pth_scene_free(s)


def test_forward_references_are_accepted_as_the_reference_parser_does(host, tmp_path, monkeypatch):
    """tinyobjloader's fixIndex does no range check and the reference reads the arrays only after the parse
    (scene.cpp:134-290), so a face may name a vertex or normal that the file defines later."""
    monkeypatch.chdir(tmp_path)
    (tmp_path / "fwd.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//2 2//1 4//1\nv 5 5 5\nvn 0 1 0\n")
    s = host.pth_scene_load(b"fwd.obj")
    assert s and host.pth_scene_triangle_count(s) == 1
    host.pth_scene_free(s)
    # ... but a vertex defined only after the face's group has been closed is not there when the face is triangulated
    (tmp_path / "late.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\ng next\nv 5 5 5\nf 1 2 3\n")
    assert not host.pth_scene_load(b"late.obj")


def test_more_than_sixteen_materials_are_refused_by_the_facade(tmp_path, monkeypatch):
    """The constant buffer holds 16 materials, the appended default one included (structs.hlsli; the
    reference indexes past the table silently).  RayTracer::loadScene reports it instead of truncating."""
    monkeypatch.chdir(tmp_path)
    L = C.CDLL(LIB)
    L.pth_tracer_new.restype = C.c_void_p
    L.pth_tracer_error.restype = C.c_char_p
    L.pth_tracer_error.argtypes = [C.c_void_p]
    L.pth_tracer_free.argtypes = [C.c_void_p]
    L.pth_tracer_load.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    (tmp_path / "m.mtl").write_text("".join(f"newmtl m{k}\nKd 0.{k % 10} 0.5 0.5\n" for k in range(16)))
    (tmp_path / "many.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\n" +
                                       "".join(f"usemtl m{k}\nf 1 2 3\n" for k in range(16)))
    r = L.pth_tracer_new(0)
    try:
        assert L.pth_tracer_load(r, b"many.obj", 32, 32) == -1             # 16 + the default = 17: before any GPU call
        assert b"17 materials" in L.pth_tracer_error(r)
    finally:
        L.pth_tracer_free(r)


def test_snapshot_schedule_is_the_reference_list(host):
    """CS.cpp:446-479: the frame counts at which the reference writes "<W>X<H>frame_<n>.bmp"."""
    due = [f for f in range(0, 400001) if host.pth_snapshot_due(f)]
    assert due == [50, 100, 200, 500, 1000, 1500, 2000, 3000, 5000, 8000, 10000, 12000, 16000, 20000, 30000, 35000,
                   40000, 50000, 60000, 70000, 80000, 100000, 120000, 150000, 200000, 300000, 400000]
