#!/usr/bin/env python3
"""Regenerates tests/golden/ from the reference checkout (run in the build container only).

  python tests/gen_golden.py [/root/reference]

For every sceneNN.obj it
  1. parses the OBJ/MTL with the reference's own vendored parser (oracle/_ref/tinyobj_dump,
     compiled from the reference tree by oracle/Makefile) -- this is the pinned part;
  2. flattens the parse with the oracle's restatement of scene.cpp / ray_tracer.cpp
     (orc_flatten) and maps materials as CS.cpp:414-437 does;
  3. runs the oracle build and stores Morton codes, the stable permutation and the tree.
It also copies the scene assets and the reference's one shipped snapshot (data files) so
the GPU box, which has no reference checkout, can load them.
"""
import os
import shutil
import subprocess
import sys
import ctypes as C

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
SCENES = ["scene01", "scene02", "scene03", "scene04"]


def parse_dump(path):
    raw = np.fromfile(path, dtype="<i4")
    fl = raw.view("<f4")
    assert raw[0] == 0x4a424f54
    nv, nn, ns, nm = (int(x) for x in raw[1:5])
    o = 5
    verts = fl[o:o + 3 * nv].copy(); o += 3 * nv
    norms = fl[o:o + 3 * nn].copy(); o += 3 * nn
    shape_off = [0]
    fv, fvn, fmat, fsm = [], [], [], []
    for _ in range(ns):
        nf = int(raw[o]); o += 1
        idx = raw[o:o + 6 * nf].reshape(nf * 3, 2); o += 6 * nf
        fv.append(idx[:, 0].copy()); fvn.append(idx[:, 1].copy())
        fmat.append(raw[o:o + nf].copy()); o += nf
        fsm.append(raw[o:o + nf].copy()); o += nf
        shape_off.append(shape_off[-1] + nf)
    mats = fl[o:o + 12 * nm].reshape(nm, 12).copy(); o += 12 * nm
    assert o == len(raw)
    return dict(verts=verts, norms=norms, shape_off=np.array(shape_off, dtype=np.int32),
                fv=np.concatenate(fv).astype(np.int32), fvn=np.concatenate(fvn).astype(np.int32),
                fmat=np.concatenate(fmat).astype(np.int32), fsm=np.concatenate(fsm).astype(np.int32),
                mats=mats)


def flatten(d):
    m = orc.Mesh()
    m.n_vertices = len(d["verts"]) // 3
    m.n_normals = len(d["norms"]) // 3
    m.vertices = d["verts"].ctypes.data
    m.normals = d["norms"].ctypes.data
    m.n_shapes = len(d["shape_off"]) - 1
    m.shape_face_offset = d["shape_off"].ctypes.data
    m.face_v = d["fv"].ctypes.data
    m.face_vn = d["fvn"].ctypes.data
    m.face_mat = d["fmat"].ctypes.data
    m.face_smooth = d["fsm"].ctypes.data
    m.n_materials = len(d["mats"])
    total = int(d["shape_off"][-1])
    prims = np.zeros(total, dtype=orc.PRIM_RENDER)
    n = orc.lib().orc_flatten(C.byref(m), prims.ctypes.data_as(C.c_void_p))
    assert n == total
    return prims


def material_cb(mats, n, eye=(0.0, 5.0, 23.0)):
    """CS.cpp:405-437: eye, node_count, and per material ke = emission[0]*10, ns = shininess,
    kd = diffuse (or 1,1,1 when ke > 0), ks = specular, tr = 1 - dissolve, ni = ior."""
    cb = np.zeros(1, dtype=orc.CB_RADIX)
    cb["node_count"] = n
    cb["pos"][0] = eye
    assert len(mats) <= 16
    for i, m in enumerate(mats.astype(np.float32)):
        diffuse, specular, emission = m[0:3], m[3:6], m[6:9]
        shininess, ior, dissolve = m[9], m[10], m[11]
        M = cb["mats"][0][i]
        M["ke"] = np.float32(emission[0]) * np.float32(10)
        M["ns"] = shininess
        if M["ke"] <= 0:
            M["kd"][:3] = diffuse
        else:
            M["kd"][:3] = (1, 1, 1)
        M["ks"][:3] = specular
        M["tr"] = np.float32(1) - np.float32(dissolve)
        M["ni"] = ior
    return cb


def main():
    os.makedirs(os.path.join(OUT, "scenes"), exist_ok=True)
    tool = os.path.join(ROOT, "oracle", "_ref", "tinyobj_dump")
    for s in SCENES:
        for ext in (".obj", ".mtl"):
            shutil.copyfile(os.path.join(REF, s + ext), os.path.join(OUT, "scenes", s + ext))
        dump = os.path.join("/tmp", s + ".tinyobj.bin")
        subprocess.run([tool, s + ".obj", dump], cwd=REF, check=True)
        d = parse_dump(dump)
        prims = flatten(d)
        cb = material_cb(d["mats"], len(prims))
        b = orc.build(prims)
        np.savez_compressed(
            os.path.join(OUT, s + ".npz"),
            prims=prims.view(np.uint8), cb=cb.view(np.uint8), mat_table=d["mats"],
            morton=b["rays"]["morton_code"], perm=b["sorted_rays"]["index"],
            bb=b["bb"].view(np.uint8), tree=b["tree"].view(np.uint8),
            parse_counts=np.array([len(d["verts"]) // 3, len(d["norms"]) // 3, len(d["shape_off"]) - 1,
                                   len(d["mats"])], dtype=np.int32),
            shape_faces=np.diff(d["shape_off"]).astype(np.int32))
        codes = b["rays"]["morton_code"]
        print(s, "n =", len(prims), "shapes =", len(d["shape_off"]) - 1, "materials =", len(d["mats"]),
              "distinct codes =", len(np.unique(codes)), "max code =", int(codes.max()),
              "bb =", b["bb"]["min"][0], b["bb"]["max"][0])
    shutil.copyfile(os.path.join(REF, "1162X1008frame_24440.png"),
                    os.path.join(OUT, "reference_snapshot_1162x1008_frame24440.png"))


if __name__ == "__main__":
    main()
