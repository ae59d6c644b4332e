import os
import sys

import numpy as np
import pytest

try:
    # PyTorch-ROCm brings its own HIP runtime under the soname libptamd.so links against; whichever is loaded first
    # serves the whole process, and torch does not find the GPU through the other one: load torch's first
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENES = ("scene01", "scene02", "scene03", "scene04")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (gfx950) device; run with -m gpu on the GPU box")


def load_golden(name):
    from oracle import orc
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {
        "prims": g["prims"].view(orc.PRIM_RENDER),
        "cb": g["cb"].view(orc.CB_RADIX),
        "mat_table": g["mat_table"],
        "morton": g["morton"],
        "perm": g["perm"],
        "bb": g["bb"].view(orc.BOUND),
        "tree": g["tree"].view(orc.TREENODE),
        "parse_counts": g["parse_counts"],
        "shape_faces": g["shape_faces"],
    }


@pytest.fixture(scope="session")
def golden():
    return {s: load_golden(s) for s in SCENES}


def make_soup(n, seed=12345, dtype=None):
    """Random triangle soup: centroids uniform in [0,1)^3, edges uniform in [-e,e]^3 with
    e = 0.5 n^(-1/3); matid 0; pad -1 (the reference's PrimitiveRender layout)."""
    from oracle import orc
    rng = np.random.default_rng(seed)
    c = rng.random((n, 3), dtype=np.float32)
    e = np.float32(0.5 * max(n, 1) ** (-1.0 / 3.0))
    prims = np.zeros(n, dtype=dtype or orc.PRIM_RENDER)
    prims["vertices"]["pos"][:, 0] = c
    prims["vertices"]["pos"][:, 1] = c + (rng.random((n, 3), dtype=np.float32) * 2 - 1) * e
    prims["vertices"]["pos"][:, 2] = c + (rng.random((n, 3), dtype=np.float32) * 2 - 1) * e
    nrm = rng.random((n, 3, 3), dtype=np.float32) * 2 - 1
    prims["vertices"]["normal"] = nrm
    prims["matid"] = 0
    prims["pad"] = -1
    return prims


def device_soup(n, seed=12345, scale=1.0, emitter_every=0, device="cuda"):
    """The same kind of soup generated ON the GPU with torch (no host copy: 50M-100M triangles), as an (n, 20)
    float32 tensor in the PrimitiveRender layout.  emitter_every = k: every k-th triangle gets material 4 (the
    emitter of soup_cb5), the others cycle through materials 0..3.  Not the same numbers as make_soup."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    e = 0.5 * max(n, 1) ** (-1.0 / 3.0)
    out = torch.zeros((n, 20), dtype=torch.float32, device=device)
    c = torch.rand((n, 3), generator=g, device=device)
    out[:, 0:3] = c * scale
    out[:, 6:9] = (c + (torch.rand((n, 3), generator=g, device=device) * 2 - 1) * e) * scale
    out[:, 12:15] = (c + (torch.rand((n, 3), generator=g, device=device) * 2 - 1) * e) * scale
    del c
    for k in (3, 9, 15):
        out[:, k:k + 3] = torch.rand((n, 3), generator=g, device=device) * 2 - 1
    ids = out.view(torch.int32)
    i = torch.arange(n, device=device, dtype=torch.int32)
    ids[:, 18] = i % 4
    if emitter_every:
        ids[:, 18] = torch.where(i % emitter_every == 0, torch.full_like(i, 4), i % 4)
    ids[:, 19] = -1
    torch.cuda.synchronize()
    return out


def soup_cb5(n, eye=(0.5, 0.5, 3.0)):
    """Materials of the large synthetic configuration: 4 diffuse + 1 emissive."""
    from oracle import orc
    cb = np.zeros(1, dtype=orc.CB_RADIX)
    cb["node_count"] = n
    cb["pos"][0] = eye
    m = cb["mats"][0]
    for k, kd in enumerate(((0.7, 0.6, 0.5), (0.3, 0.6, 0.7), (0.6, 0.7, 0.3), (0.75, 0.75, 0.75))):
        m[k]["ns"] = 1; m[k]["ni"] = 1; m[k]["kd"][:3] = kd
    m[4]["ke"] = 10; m[4]["ns"] = 1; m[4]["ni"] = 1; m[4]["kd"][:3] = (1, 1, 1)
    return cb


def soup_cb(n, eye=(0.5, 0.5, 3.0)):
    """4 diffuse-ish materials + 1 emitter, for soups."""
    from oracle import orc
    cb = np.zeros(1, dtype=orc.CB_RADIX)
    cb["node_count"] = n
    cb["pos"][0] = eye
    m = cb["mats"][0]
    m[0]["ns"] = 1; m[0]["ni"] = 1; m[0]["kd"][:3] = (0.7, 0.6, 0.5)
    m[1]["ns"] = 50; m[1]["ni"] = 1; m[1]["kd"][:3] = (0.1, 0.1, 0.1); m[1]["ks"][:3] = (0.8, 0.8, 0.8)
    m[2]["ns"] = 1; m[2]["ni"] = 1.5; m[2]["tr"] = 0.9; m[2]["kd"][:3] = (1, 1, 1)
    m[3]["ke"] = 10; m[3]["ns"] = 1; m[3]["ni"] = 1; m[3]["kd"][:3] = (1, 1, 1)
    return cb


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    """Creates contexts on device 0; fails (never skips) when the library or device is missing."""
    import ptamd
    made = []

    def make():
        c = ptamd.Context(0)
        made.append(c)
        return c
    yield make
    for c in made:
        c.close()
