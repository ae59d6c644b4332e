"""CPU: the C-ABI library loads and exports exactly what include/ptamd.h declares; layouts match."""
import ctypes as C
import os
import re

import numpy as np

import ptamd
from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "ptamd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_list_agree():
    assert _declared() == sorted(ptamd.EXPORTS)


def test_library_exports_every_declared_symbol():
    L = ptamd.load()
    for name in _declared():
        assert hasattr(L, name), f"libptamd.so lacks {name}"
    assert b"gfx950" in L.pt_version()


def test_struct_sizes_match_reference_layouts():
    # sizes of the reference's shared structs (structs.hpp / structs.hlsli)
    assert ptamd.BOUND.itemsize == 28 and ptamd.TREENODE.itemsize == 48
    assert ptamd.PRIM_RENDER.itemsize == 80 and ptamd.PRIM_RAY_CS.itemsize == 32
    assert ptamd.MATERIAL.itemsize == 48 and ptamd.CB_RADIX.itemsize == 784
    assert ptamd.HIT.itemsize == 24
    assert ptamd.TREENODE.fields["bound"][1] == 16 and ptamd.PRIM_RENDER.fields["matid"][1] == 72
    assert ptamd.CB_RADIX.fields["mats"][1] == 16


def test_host_only_entry_points_need_no_gpu():
    # fixed-seed convention and colour-target bytes are pure host functions
    bg = np.random.MT19937()
    bg._legacy_seeding(1)
    raw = bg.random_raw(16 * 8)
    assert np.array_equal(ptamd.frame_seeds(1, 0, 8), raw[::16].astype(np.uint32))
    assert np.array_equal(ptamd.frame_seeds(1, 3, 2), raw[48::16][:2].astype(np.uint32))
    tab = ptamd.albedo_table([[0.5, 0.5, 0.5], [1.0, 0.0, 0.25]], srgb_quirk=True)
    assert list(tab[0]) == [188, 188, 188, 255] and list(tab[1][:2]) == [255, 0]
    tab = ptamd.albedo_table([[0.5, 0.5, 0.5]], srgb_quirk=False)
    assert list(tab[0]) == [128, 128, 128, 255]


def test_calls_fail_loudly_without_context():
    L = ptamd.load()
    assert L.pt_build(C.c_void_p(None), C.c_uint(0)) != 0
    assert L.pt_trace(C.c_void_p(None), C.c_uint32(0), C.c_uint32(0)) != 0
    assert L.pt_synchronize(C.c_void_p(None)) != 0
