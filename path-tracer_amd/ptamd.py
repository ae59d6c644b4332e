"""ctypes binding of libptamd.so (C ABI in include/ptamd.h).

The library is the product; this module is plumbing for tests, bench.py and Python
callers.  It never falls back to a CPU path: if libptamd.so is missing or no gfx950
device is present, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PTAMD_LIB selects another build of the same library (A/B measurements of kernel variants)
LIB_PATH = os.environ.get("PTAMD_LIB") or os.path.join(_HERE, "libptamd.so")

# numpy views of the interchange structs (include/ptamd_structs.h)
BOUND = np.dtype([("min", "<f4", 3), ("pad", "<f4"), ("max", "<f4", 3)])
TREENODE = np.dtype([("left", "<i4"), ("right", "<i4"), ("parent", "<i4"), ("index", "<i4"),
                     ("bound", BOUND), ("padding", "<i4")])
XYZN = np.dtype([("pos", "<f4", 3), ("normal", "<f4", 3)])
PRIM_RENDER = np.dtype([("vertices", XYZN, 3), ("matid", "<i4"), ("pad", "<i4")])
PRIM_RAY_CS = np.dtype([("bound", BOUND), ("index", "<i4")])
MATERIAL = np.dtype([("ke", "<f4"), ("ns", "<f4"), ("tr", "<f4"), ("ni", "<f4"), ("kd", "<f4", 4), ("ks", "<f4", 4)])
CB_RADIX = np.dtype([("node_count", "<i4"), ("pos", "<f4", 3), ("mats", MATERIAL, 16)])
HIT = np.dtype([("hit", "<i4"), ("primitive", "<i4"), ("t", "<f4"), ("bary", "<f4", 3)])

PT_BUILD_DEFAULT = 0
PT_BUILD_MORTON_FULL = 1
PT_BUILD_KARRAS_LINKS = 4
PT_BUILD_TREE_ONLY = 8
PT_BUILD_TOPOLOGY_MORTON = 16
PT_NOALLOC_MAX_PRIMS = 131072
PT_ACCUM_LITERAL = 0
PT_ACCUM_LINEAR = 1
PT_SCHEDULE_PERSISTENT = 0
PT_SCHEDULE_WAVEFRONT = 1
PT_TRAVERSAL_REFERENCE = 0
PT_TRAVERSAL_NEAR_FIRST = 1
STAGES = ("bounds", "morton", "sort", "topology", "refit", "pack")

# every symbol include/ptamd.h declares
EXPORTS = (
    "pt_create", "pt_destroy", "pt_last_error", "pt_set_stream", "pt_synchronize", "pt_version",
    "pt_upload_primitives", "pt_upload_primitives_device", "pt_set_constants",
    "pt_build", "pt_build_times", "pt_build_tree_noalloc",
    "pt_download_scene_bound", "pt_download_morton", "pt_download_permutation",
    "pt_download_sorted_primitives", "pt_download_tree",
    "pt_resize", "pt_upload_gbuffer", "pt_download_gbuffer", "pt_primary_hits", "pt_albedo_bytes",
    "pt_trace", "pt_trace_region", "pt_trace_frames", "pt_frame_seeds", "pt_get_frame_count", "pt_reset_accumulation",
    "pt_linear_sum_device_ptr", "pt_resolve",
    "pt_comm_unique_id", "pt_comm_init_rank", "pt_comm_destroy", "pt_comm_count", "pt_reduce_linear_sum",
    "pt_download_output", "pt_download_display", "pt_output_device_ptr",
    "pt_set_stack_size", "pt_set_max_segments", "pt_set_traversal_order", "pt_set_trace_overlap", "pt_set_trace_schedule", "pt_set_trace_timing", "pt_get_trace_times", "pt_set_detailed_counters", "pt_get_counters", "pt_reset_counters", "pt_intersect",
)


class Camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("at", C.c_float * 3), ("up", C.c_float * 3),
                ("fovy", C.c_float), ("znear", C.c_float), ("zfar", C.c_float)]


class Counters(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("node_visits", C.c_uint64),
                ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class PtError(RuntimeError):
    pass


_lib = None


def load():
    """dlopen libptamd.so (needs the HIP runtime, not a GPU)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PtError(f"{LIB_PATH} not built: run __graft_entry__.build()")
        L = C.CDLL(LIB_PATH)
        L.pt_last_error.restype = C.c_char_p
        L.pt_last_error.argtypes = [C.c_void_p]
        L.pt_version.restype = C.c_char_p
        L.pt_destroy.restype = None
        L.pt_destroy.argtypes = [C.c_void_p]
        L.pt_albedo_bytes.restype = None
        L.pt_frame_seeds.restype = None
        _lib = L
    return _lib


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


def frame_seeds(base_seed, first, count):
    out = np.zeros(count, dtype=np.uint32)
    load().pt_frame_seeds(C.c_uint32(base_seed), C.c_uint32(first), C.c_uint32(count), out.ctypes.data_as(C.c_void_p))
    return out


def unique_comm_id():
    """128-byte id for comm_init_rank (ncclGetUniqueId through the library); made on ONE rank."""
    buf = (C.c_char * 128)()
    rc = load().pt_comm_unique_id(buf)
    if rc != 0:
        raise PtError(f"pt_comm_unique_id failed ({rc}): {load().pt_last_error(None).decode()}")
    return bytes(buf.raw)


def albedo_table(kd_rows, srgb_quirk=True):
    """[16][4] bytes of the colour target per material from raw Kd rows (CS.cpp:357-359)."""
    tab = np.zeros((16, 4), dtype=np.uint8)
    out = (C.c_uint8 * 3)()
    for m, kd in enumerate(kd_rows):
        load().pt_albedo_bytes((C.c_float * 3)(*[float(x) for x in kd[:3]]), C.c_int(1 if srgb_quirk else 0), out)
        tab[m, :3] = list(out)
    tab[:, 3] = 255
    return tab


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One GPU's tracer state (pt_context)."""

    def __init__(self, device=0, stream=None):
        self.L = load()
        self.h = C.c_void_p()
        rc = self.L.pt_create(C.c_int(device), C.byref(self.h))
        if rc != 0:
            raise PtError(f"pt_create({device}) failed ({rc}): {self.L.pt_last_error(None).decode()}")
        self.n = 0
        self.W = self.H = 0
        if stream is not None:
            self.set_stream(stream)

    def _ck(self, rc):
        if rc != 0:
            raise PtError(f"libptamd error {rc}: {self.L.pt_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            self.L.pt_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle):
        self._ck(self.L.pt_set_stream(self.h, C.c_void_p(int(stream_handle))))

    def synchronize(self):
        self._ck(self.L.pt_synchronize(self.h))

    # -- scene ------------------------------------------------------------
    def upload_primitives(self, prims):
        prims = np.ascontiguousarray(prims, dtype=PRIM_RENDER)
        self._ck(self.L.pt_upload_primitives(self.h, _p(prims), C.c_int(len(prims))))
        self.n = len(prims)

    def upload_primitives_device(self, dev_ptr, n):
        self._ck(self.L.pt_upload_primitives_device(self.h, C.c_void_p(int(dev_ptr)), C.c_int(n)))
        self.n = n

    def set_constants(self, cb):
        cb = np.ascontiguousarray(cb, dtype=CB_RADIX)
        self._ck(self.L.pt_set_constants(self.h, _p(cb)))

    def build(self, flags=PT_BUILD_DEFAULT):
        self._ck(self.L.pt_build(self.h, C.c_uint(flags)))

    def build_tree_noalloc(self, dev_prims, n, dev_morton, dev_perm, dev_tree, flags=0):
        """Device pointers in, device pointers out; nothing is allocated (pt_build_tree_noalloc)."""
        self._ck(self.L.pt_build_tree_noalloc(self.h, C.c_void_p(int(dev_prims)), C.c_int(n), C.c_uint(flags),
                                              C.c_void_p(int(dev_morton) if dev_morton else None),
                                              C.c_void_p(int(dev_perm)), C.c_void_p(int(dev_tree))))

    def build_times(self):
        ms = (C.c_float * len(STAGES))()
        tot = C.c_float()
        passes = C.c_int()
        self._ck(self.L.pt_build_times(self.h, ms, C.byref(tot), C.byref(passes)))
        d = {k: float(ms[i]) for i, k in enumerate(STAGES)}
        d["total"] = float(tot.value)
        d["sort_passes"] = passes.value
        return d

    def scene_bound(self):
        bb = np.zeros(1, dtype=BOUND)
        self._ck(self.L.pt_download_scene_bound(self.h, _p(bb)))
        return bb

    def morton(self):
        a = np.zeros(self.n, dtype=np.uint64)
        self._ck(self.L.pt_download_morton(self.h, _p(a)))
        return a

    def permutation(self):
        a = np.zeros(self.n, dtype=np.int32)
        self._ck(self.L.pt_download_permutation(self.h, _p(a)))
        return a

    def sorted_primitives(self):
        a = np.zeros(self.n, dtype=PRIM_RAY_CS)
        self._ck(self.L.pt_download_sorted_primitives(self.h, _p(a)))
        return a

    def tree(self):
        a = np.zeros(2 * self.n, dtype=TREENODE)
        self._ck(self.L.pt_download_tree(self.h, _p(a)))
        return a

    # -- images -----------------------------------------------------------
    def resize(self, W, H):
        self._ck(self.L.pt_resize(self.h, C.c_int(W), C.c_int(H)))
        self.W, self.H = W, H

    def upload_gbuffer(self, diffuse, normal, world):
        diffuse = np.ascontiguousarray(diffuse, dtype=np.uint8)
        normal = np.ascontiguousarray(normal, dtype=np.float32)
        world = np.ascontiguousarray(world, dtype=np.float32)
        assert diffuse.size == self.W * self.H * 4 and normal.size == self.W * self.H * 4 and world.size == self.W * self.H * 4
        self._ck(self.L.pt_upload_gbuffer(self.h, _p(diffuse), _p(normal), _p(world)))

    def download_gbuffer(self):
        d = np.zeros((self.H, self.W, 4), dtype=np.uint8)
        n = np.zeros((self.H, self.W, 4), dtype=np.float32)
        w = np.zeros((self.H, self.W, 4), dtype=np.float32)
        self._ck(self.L.pt_download_gbuffer(self.h, _p(d), _p(n), _p(w)))
        return d, n, w

    def primary_hits(self, cam, albedo8):
        albedo8 = np.ascontiguousarray(albedo8, dtype=np.uint8)
        assert albedo8.size == 64
        self._ck(self.L.pt_primary_hits(self.h, C.byref(cam), _p(albedo8)))

    # -- trace ------------------------------------------------------------
    def trace(self, framecount, seed):
        self._ck(self.L.pt_trace(self.h, C.c_uint32(framecount), C.c_uint32(int(seed))))

    def trace_region(self, framecount, seed, x0, y0, x1, y1):
        self._ck(self.L.pt_trace_region(self.h, C.c_uint32(framecount), C.c_uint32(int(seed)),
                                        C.c_int(x0), C.c_int(y0), C.c_int(x1), C.c_int(y1)))

    def trace_frames(self, first, seeds, mode=PT_ACCUM_LITERAL):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        self._ck(self.L.pt_trace_frames(self.h, C.c_uint32(first), C.c_uint32(len(seeds)), _p(seeds), C.c_int(mode)))

    def frame_count(self):
        v = C.c_uint32()
        self._ck(self.L.pt_get_frame_count(self.h, C.byref(v)))
        return v.value

    def reset_accumulation(self):
        self._ck(self.L.pt_reset_accumulation(self.h))

    def linear_sum_ptr(self):
        p = C.c_void_p()
        nbytes = C.c_size_t()
        self._ck(self.L.pt_linear_sum_device_ptr(self.h, C.byref(p), C.byref(nbytes)))
        return p.value, nbytes.value

    def resolve(self):
        self._ck(self.L.pt_resolve(self.h))

    # -- multi-GPU merge (RCCL behind the C ABI) ------------------------------
    def comm_init_rank(self, n_ranks, rank, comm_id):
        """comm_id: the 128 bytes of unique_comm_id() made on rank 0."""
        buf = (C.c_char * 128).from_buffer_copy(bytes(comm_id))
        self._ck(self.L.pt_comm_init_rank(self.h, C.c_int(n_ranks), C.c_int(rank), buf))

    def comm_destroy(self):
        self._ck(self.L.pt_comm_destroy(self.h))

    def comm_count(self):
        """Ranks of this context's RCCL communicator (ncclCommCount), 0 without one."""
        n = C.c_int(0)
        self._ck(self.L.pt_comm_count(self.h, C.byref(n)))
        return n.value

    def reduce_linear_sum(self, nccl_comm=None):
        self._ck(self.L.pt_reduce_linear_sum(self.h, C.c_void_p(nccl_comm)))

    def output(self):
        a = np.zeros((self.H, self.W, 4), dtype=np.float32)
        self._ck(self.L.pt_download_output(self.h, _p(a)))
        return a

    def display(self):
        a = np.zeros((self.H, self.W, 4), dtype=np.uint8)
        self._ck(self.L.pt_download_display(self.h, _p(a)))
        return a

    def linear_sum(self):
        """Host copy of the linear sum (through the runtime's memcpy, via torch-free ctypes)."""
        import ctypes.util  # noqa: F401
        ptr, nbytes = self.linear_sum_ptr()
        hip = C.CDLL("libamdhip64.so")
        a = np.zeros((self.H, self.W, 4), dtype=np.float32)
        self.synchronize()
        rc = hip.hipMemcpy(_p(a), C.c_void_p(ptr), C.c_size_t(nbytes), C.c_int(2))  # hipMemcpyDeviceToHost
        if rc != 0:
            raise PtError(f"hipMemcpy failed: {rc}")
        return a

    # -- options / counters -------------------------------------------------
    def set_stack_size(self, entries):
        self._ck(self.L.pt_set_stack_size(self.h, C.c_int(entries)))

    def set_max_segments(self, segments):
        self._ck(self.L.pt_set_max_segments(self.h, C.c_int(segments)))

    def set_traversal_order(self, order):
        self._ck(self.L.pt_set_traversal_order(self.h, C.c_int(order)))

    def set_trace_overlap(self, enable):
        self._ck(self.L.pt_set_trace_overlap(self.h, C.c_int(1 if enable else 0)))

    def set_trace_schedule(self, schedule):
        self._ck(self.L.pt_set_trace_schedule(self.h, C.c_int(schedule)))

    def set_trace_timing(self, on):
        self._ck(self.L.pt_set_trace_timing(self.h, C.c_int(1 if on else 0)))

    def trace_times(self, capacity=65536):
        """Device-side duration (ms) of the trace kernel of every launch since timing was enabled."""
        ms = np.zeros(capacity, dtype=np.float32)
        cnt = C.c_int(0)
        self._ck(self.L.pt_get_trace_times(self.h, _p(ms), C.c_int(capacity), C.byref(cnt)))
        return ms[:min(cnt.value, capacity)].copy()

    def set_detailed_counters(self, on):
        self._ck(self.L.pt_set_detailed_counters(self.h, C.c_int(1 if on else 0)))

    def counters(self):
        c = Counters()
        self._ck(self.L.pt_get_counters(self.h, C.byref(c)))
        return c.as_dict()

    def reset_counters(self):
        self._ck(self.L.pt_reset_counters(self.h))

    def intersect(self, rays8):
        rays8 = np.ascontiguousarray(rays8, dtype=np.float32).reshape(-1, 8)
        out = np.zeros(len(rays8), dtype=HIT)
        self._ck(self.L.pt_intersect(self.h, _p(rays8), C.c_int(len(rays8)), _p(out)))
        return out
