"""TEST INFRASTRUCTURE: ctypes/numpy binding of oracle/_build/liboracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# numpy views of the interchange structs (include/ptamd_structs.h)
BOUND = np.dtype([("min", "<f4", 3), ("pad", "<f4"), ("max", "<f4", 3)])
TREENODE = np.dtype([("left", "<i4"), ("right", "<i4"), ("parent", "<i4"), ("index", "<i4"),
                     ("bound", BOUND), ("padding", "<i4")])
XYZN = np.dtype([("pos", "<f4", 3), ("normal", "<f4", 3)])
PRIM_RENDER = np.dtype([("vertices", XYZN, 3), ("matid", "<i4"), ("pad", "<i4")])
PRIM_RAY = np.dtype([("morton_code", "<u8"), ("bound", BOUND), ("index", "<i4")])
PRIM_RAY_CS = np.dtype([("bound", BOUND), ("index", "<i4")])
MATERIAL = np.dtype([("ke", "<f4"), ("ns", "<f4"), ("tr", "<f4"), ("ni", "<f4"), ("kd", "<f4", 4), ("ks", "<f4", 4)])
CB_RADIX = np.dtype([("node_count", "<i4"), ("pos", "<f4", 3), ("mats", MATERIAL, 16)])
assert BOUND.itemsize == 28 and TREENODE.itemsize == 48 and PRIM_RENDER.itemsize == 80
assert PRIM_RAY.itemsize == 40 and PRIM_RAY_CS.itemsize == 32 and MATERIAL.itemsize == 48 and CB_RADIX.itemsize == 784


class Mesh(C.Structure):
    _fields_ = [("n_vertices", C.c_int), ("n_normals", C.c_int),
                ("vertices", C.c_void_p), ("normals", C.c_void_p),
                ("n_shapes", C.c_int), ("shape_face_offset", C.c_void_p),
                ("face_v", C.c_void_p), ("face_vn", C.c_void_p), ("face_mat", C.c_void_p),
                ("face_smooth", C.c_void_p), ("n_materials", C.c_int)]


class Camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("at", C.c_float * 3), ("up", C.c_float * 3),
                ("fovy", C.c_float), ("znear", C.c_float), ("zfar", C.c_float)]


class Scene(C.Structure):
    _fields_ = [("n", C.c_int), ("prims", C.c_void_p), ("tree", C.c_void_p), ("cb", C.c_void_p),
                ("stack_size", C.c_int), ("max_segments", C.c_int), ("near_first", C.c_int)]


class Counters(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("node_visits", C.c_uint64),
                ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}

    def add(self, o):
        for k, _ in self._fields_:
            setattr(self, k, getattr(self, k) + getattr(o, k))


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle not built: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.orc_morton.restype = C.c_uint64
        L.orc_rand_init.restype = C.c_uint32
        L.orc_rand_init.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_rand_next.restype = C.c_float
        for f in ("orc_sinf", "orc_cosf", "orc_exp2f", "orc_log2f"):
            getattr(L, f).restype = C.c_float
            getattr(L, f).argtypes = [C.c_float]
        L.orc_powf.restype = C.c_float
        L.orc_powf.argtypes = [C.c_float, C.c_float]
        L.orc_flatten.restype = C.c_int
        L.orc_intersect_one.restype = C.c_int
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def n_threads():
    return max(1, min(int(os.environ.get("ORC_THREADS", "8")), os.cpu_count() or 1))


def _rows(H, parts):
    parts = max(1, min(parts, H))
    edges = np.linspace(0, H, parts + 1).astype(int)
    return [(int(edges[i]), int(edges[i + 1])) for i in range(parts) if edges[i + 1] > edges[i]]


# ------------------------------------------------------------------ build ---
def default_camera(eye=(0.0, 5.0, 23.0), at=(0.0, 5.0, 0.0)):
    """CS.cpp:49-50,201,319: eye (0,5,23), at (0,5,0), up (0,1,0), fovY = D3DX_PI/6, zn .005, zf 100."""
    cam = Camera()
    cam.eye[:] = eye
    cam.at[:] = at
    cam.up[:] = (0.0, 1.0, 0.0)
    cam.fovy = np.float32(np.float32(3.141592654) / np.float32(6))
    cam.znear = 0.005
    cam.zfar = 100.0
    return cam


def build(prims):
    """Full CPU build. Returns dict(rays(unsorted, with codes), bb, sorted_rays, sorted_cs, tree[2n+1])."""
    L = lib()
    n = len(prims)
    prims = np.ascontiguousarray(prims, dtype=PRIM_RENDER)
    rays = np.zeros(n, dtype=PRIM_RAY)
    L.orc_bounds(_p(prims), C.c_int(n), _p(rays))
    bb = np.zeros(1, dtype=BOUND)
    L.orc_scene_bound(_p(rays), C.c_int(n), _p(bb))
    L.orc_morton_all(_p(rays), C.c_int(n), _p(bb))
    srt = rays.copy()
    L.orc_sort(_p(srt), C.c_int(n))
    cs = np.zeros(n, dtype=PRIM_RAY_CS)
    L.orc_strip(_p(srt), C.c_int(n), _p(cs))
    tree = np.zeros(2 * n + 1, dtype=TREENODE)
    L.orc_radix_tree(C.c_int(n), _p(tree))
    L.orc_refit(C.c_int(n), _p(cs), _p(tree))
    return {"rays": rays, "bb": bb, "sorted_rays": srt, "sorted_cs": cs, "tree": tree}


def build_full_codes(prims):
    """NON-PARITY build of PT_BUILD_MORTON_FULL: the reference's code arithmetic with 2^21 - 1 cells per axis
    (float32, strict IEEE) instead of its 21, stable order, then the reference's index-keyed tree and refit."""
    L = lib()
    n = len(prims)
    prims = np.ascontiguousarray(prims, dtype=PRIM_RENDER)
    rays = np.zeros(n, dtype=PRIM_RAY)
    L.orc_bounds(_p(prims), C.c_int(n), _p(rays))
    bb = np.zeros(1, dtype=BOUND)
    L.orc_scene_bound(_p(rays), C.c_int(n), _p(bb))
    f = np.float32
    mn, mx = rays["bound"]["min"], rays["bound"]["max"]
    mid = ((mn + mx).astype(f) / f(2)).astype(f)
    q = ((((mid - bb["min"][0]).astype(f) / (bb["max"][0] - bb["min"][0]).astype(f)).astype(f)) * f(2097151.0)).astype(f)
    q = np.where(np.isnan(q), f(0), q).astype(np.uint64)
    code = np.zeros(n, dtype=np.uint64)
    for bit in range(21):
        for axis, sh in ((0, 2), (1, 1), (2, 0)):
            code |= ((q[:, axis] >> np.uint64(bit)) & np.uint64(1)) << np.uint64(3 * bit + sh)
    rays["morton_code"] = code
    srt = rays[np.argsort(code, kind="stable")]
    cs = np.zeros(n, dtype=PRIM_RAY_CS)
    L.orc_strip(_p(srt), C.c_int(n), _p(cs))
    tree = np.zeros(2 * n + 1, dtype=TREENODE)
    L.orc_radix_tree(C.c_int(n), _p(tree))
    L.orc_refit(C.c_int(n), _p(cs), _p(tree))
    return {"rays": rays, "bb": bb, "sorted_rays": srt, "sorted_cs": cs, "tree": tree}


def build_codes_topology(prims, full=False):
    """NON-PARITY build of PT_BUILD_TOPOLOGY_MORTON: codes and order as build() (or build_full_codes() with `full`),
    then the binary radix tree over the sorted codes (equal codes told apart by position) instead of the
    reference's index-keyed tree: n - 1 internal nodes, slots n - 1 and 2n of the array stay zero."""
    L = lib()
    b = build_full_codes(prims) if full else build(prims)
    n = len(prims)
    assert n >= 2
    codes = np.ascontiguousarray(b["sorted_rays"]["morton_code"])
    tree = np.zeros(2 * n + 1, dtype=TREENODE)
    L.orc_radix_tree_codes(C.c_int(n), _p(codes), _p(tree))
    L.orc_refit(C.c_int(n), _p(b["sorted_cs"]), _p(tree))
    b["tree"] = tree
    return b


def tree_height(tree, n):
    """internal nodes on the longest root-to-leaf path"""
    h = np.zeros(len(tree), dtype=np.int32)
    order, stack = [], [0]
    while stack:
        i = stack.pop()
        order.append(i)
        for c in (int(tree["left"][i]), int(tree["right"][i])):
            if c < n:
                stack.append(c)
    for i in reversed(order):
        h[i] = 1 + max(h[tree["left"][i]], h[tree["right"][i]])
    return int(h[0])


def radix_tree(n, literal=False):
    tree = np.zeros(2 * n + 1, dtype=TREENODE)
    (lib().orc_radix_tree_literal if literal else lib().orc_radix_tree)(C.c_int(n), _p(tree))
    return tree


def frame_seeds(base_seed, first, count):
    out = np.zeros(count, dtype=np.uint32)
    lib().orc_frame_seeds(C.c_uint32(base_seed), C.c_uint32(first), C.c_uint32(count), _p(out))
    return out


def albedo_table(cb, srgb_quirk=True, kd_override=None):
    """[16][4] bytes the colour target holds per material (alpha 255)."""
    tab = np.zeros((16, 4), dtype=np.uint8)
    out = (C.c_uint8 * 3)()
    for m in range(16):
        kd = (C.c_float * 3)(*(kd_override[m] if kd_override is not None else cb["mats"][0][m]["kd"][:3]))
        lib().orc_albedo_bytes(kd, C.c_int(1 if srgb_quirk else 0), out)
        tab[m, :3] = list(out)
        tab[m, 3] = 255
    return tab


class OracleScene:
    """Holds numpy buffers alive and exposes the C scene struct."""

    def __init__(self, prims, tree, cb, stack_size=15, max_segments=0, near_first=False):
        self.prims = np.ascontiguousarray(prims, dtype=PRIM_RENDER)
        self.tree = np.ascontiguousarray(tree, dtype=TREENODE)
        self.cb = np.ascontiguousarray(cb, dtype=CB_RADIX)
        self.n = len(self.prims)
        assert len(self.tree) == 2 * self.n + 1
        self.c = Scene(self.n, self.prims.ctypes.data, self.tree.ctypes.data, self.cb.ctypes.data, stack_size, max_segments,
                       1 if near_first else 0)


def primary(scene, cam, W, H, albedo8, threads=None):
    L = lib()
    diffuse = np.zeros((H, W, 4), dtype=np.uint8)
    normal = np.zeros((H, W, 4), dtype=np.float32)
    world = np.zeros((H, W, 4), dtype=np.float32)
    albedo8 = np.ascontiguousarray(albedo8, dtype=np.uint8)

    def work(r):
        L.orc_primary(C.byref(scene.c), C.byref(cam), C.c_int(W), C.c_int(H), _p(albedo8),
                      _p(diffuse), _p(normal), _p(world), C.c_int(r[0]), C.c_int(r[1]))
    t = threads or n_threads()
    with ThreadPoolExecutor(t) as ex:
        list(ex.map(work, _rows(H, t * 4)))
    return diffuse, normal, world


def trace_frame(scene, gbuf, framecount, seed, old, want_radiance=False, threads=None):
    """One literal frame. Returns (out_rgba, out_unorm, radiance|None, counters dict)."""
    L = lib()
    diffuse, normal, world = gbuf
    H, W = diffuse.shape[:2]
    out = np.zeros((H, W, 4), dtype=np.float32)
    outu = np.zeros((H, W, 4), dtype=np.uint8)
    rad = np.zeros((H, W, 3), dtype=np.float32) if want_radiance else None
    old = np.ascontiguousarray(old, dtype=np.float32)
    t = threads or n_threads()
    rows = _rows(H, t * 4)
    cnts = [Counters() for _ in rows]

    def work(i):
        r = rows[i]
        L.orc_trace_frame(C.byref(scene.c), C.c_int(W), C.c_int(H), _p(diffuse), _p(normal), _p(world),
                          C.c_uint32(framecount), C.c_uint32(int(seed)), _p(old), _p(out), _p(outu),
                          _p(rad) if rad is not None else None, C.byref(cnts[i]), C.c_int(r[0]), C.c_int(r[1]))
    with ThreadPoolExecutor(t) as ex:
        list(ex.map(work, range(len(rows))))
    tot = Counters()
    for c in cnts:
        tot.add(c)
    return out, outu, rad, tot.as_dict()


def trace_frame_linear(scene, gbuf, framecount, seed, acc, threads=None):
    """acc (H,W,4 float32) is updated in place; returns counters dict."""
    L = lib()
    diffuse, normal, world = gbuf
    H, W = diffuse.shape[:2]
    assert acc.dtype == np.float32 and acc.flags.c_contiguous
    t = threads or n_threads()
    rows = _rows(H, t * 4)
    cnts = [Counters() for _ in rows]

    def work(i):
        r = rows[i]
        L.orc_trace_frame_linear(C.byref(scene.c), C.c_int(W), C.c_int(H), _p(diffuse), _p(normal), _p(world),
                                 C.c_uint32(framecount), C.c_uint32(int(seed)), _p(acc), C.byref(cnts[i]),
                                 C.c_int(r[0]), C.c_int(r[1]))
    with ThreadPoolExecutor(t) as ex:
        list(ex.map(work, range(len(rows))))
    tot = Counters()
    for c in cnts:
        tot.add(c)
    return tot.as_dict()


def resolve(acc, frames):
    H, W = acc.shape[:2]
    out = np.zeros((H, W, 4), dtype=np.float32)
    outu = np.zeros((H, W, 4), dtype=np.uint8)
    lib().orc_resolve(C.c_int(W), C.c_int(H), _p(acc), C.c_uint32(frames), _p(out), _p(outu))
    return out, outu


def intersect_one(scene, origin, direction, t_min=1e-4, t_max=10000.0):
    o = (C.c_float * 3)(*origin)
    d = (C.c_float * 3)(*direction)
    prim = C.c_int(-1)
    bary = (C.c_float * 3)()
    t = C.c_float(0)
    cnt = Counters()
    hit = lib().orc_intersect_one(C.byref(scene.c), o, d, C.c_float(t_min), C.c_float(t_max),
                                  C.byref(prim), bary, C.byref(t), C.byref(cnt))
    return bool(hit), prim.value, tuple(bary), t.value, cnt.as_dict()
