"""CPU: the oracle's trace half -- elementary functions, RNG, traversal against brute force,
accumulation modes."""
import ctypes as C

import numpy as np
import pytest

from oracle import orc


def _ulps(a, ref):
    a = np.asarray(a, np.float32)
    ref64 = np.asarray(ref, np.float64)
    spacing = np.spacing(np.abs(ref64).astype(np.float32)).astype(np.float64)
    return np.abs(a.astype(np.float64) - ref64) / spacing


def _map(fn, xs):
    f = getattr(orc.lib(), fn)
    return np.array([f(C.c_float(float(x))) for x in xs], dtype=np.float32)


def test_sin_cos_accuracy():
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.random(4000, dtype=np.float32) * np.float32(2 * 3.1415926),
                         np.float32([0.0, 1e-8, 0.5, 1.5707963, 3.1415926, 4.712389, 6.2831852])])
    s, c = _map("orc_sinf", xs), _map("orc_cosf", xs)
    # absolute error: a few 1e-8 near zero crossings, otherwise ~1 ulp of a value near 1
    assert np.max(np.abs(s - np.sin(xs.astype(np.float64)))) < 1.5e-7
    assert np.max(np.abs(c - np.cos(xs.astype(np.float64)))) < 1.5e-7
    np.testing.assert_allclose(s * s + c * c, 1.0, atol=5e-7)


def test_exp2_log2_pow_accuracy_and_edges():
    rng = np.random.default_rng(2)
    xs = (rng.random(3000, dtype=np.float32) * 60 - 40).astype(np.float32)
    assert _ulps(_map("orc_exp2f", xs), np.exp2(xs.astype(np.float64))).max() < 2.5
    ys = np.exp2((rng.random(3000) * 40 - 30)).astype(np.float32)
    lg = _map("orc_log2f", ys)
    assert np.max(np.abs(lg - np.log2(ys.astype(np.float64))) / np.maximum(1e-3, np.abs(np.log2(ys.astype(np.float64))))) < 5e-7
    L = orc.lib()
    inf = float("inf")
    assert L.orc_exp2f(C.c_float(-inf)) == 0.0 and L.orc_exp2f(C.c_float(-200.0)) == 0.0
    assert L.orc_exp2f(C.c_float(200.0)) == inf
    assert L.orc_exp2f(C.c_float(0.0)) == 1.0 and L.orc_exp2f(C.c_float(3.0)) == 8.0
    assert L.orc_exp2f(C.c_float(-140.0)) == pytest.approx(2.0 ** -140, rel=1e-6)      # subnormal result
    assert L.orc_log2f(C.c_float(0.0)) == -inf and np.isnan(L.orc_log2f(C.c_float(-1.0)))
    assert L.orc_log2f(C.c_float(1.0)) == 0.0 and L.orc_log2f(C.c_float(8.0)) == 3.0
    assert L.orc_log2f(C.c_float(1e-40)) == pytest.approx(np.log2(np.float32(1e-40).astype(np.float64)), rel=1e-6)
    # HLSL pow semantics: pow(0, y > 0) = 0, negative base = NaN, pow(x, 1) ~ x
    assert L.orc_powf(C.c_float(0.0), C.c_float(2.2)) == 0.0
    assert np.isnan(L.orc_powf(C.c_float(-0.5), C.c_float(2.2)))
    v = np.float32(rng.random(2000, dtype=np.float32))
    p = np.array([L.orc_powf(C.c_float(float(x)), C.c_float(2.2)) for x in v], dtype=np.float32)
    np.testing.assert_allclose(p, v.astype(np.float64) ** np.float64(np.float32(2.2)), rtol=3e-6, atol=1e-12)


def _tea_py(v0, v1):
    s0 = 0
    M = 0xFFFFFFFF
    for _ in range(16):
        s0 = (s0 + 0x9e3779b9) & M
        v0 = (v0 + ((((v1 << 4) & M) + 0xa341316c & M) ^ ((v1 + s0) & M) ^ (((v1 >> 5) + 0xc8013ea4) & M))) & M
        v1 = (v1 + ((((v0 << 4) & M) + 0xad90777d & M) ^ ((v0 + s0) & M) ^ (((v0 >> 5) + 0x7e95761e) & M))) & M
    return v0


def test_rng_bit_exact():
    L = orc.lib()
    rng = np.random.default_rng(3)
    for v0, v1 in rng.integers(0, 2 ** 32, size=(200, 2), dtype=np.uint64):
        assert L.orc_rand_init(int(v0), int(v1)) == _tea_py(int(v0), int(v1))
    assert L.orc_rand_init(0, 0) == _tea_py(0, 0)
    s = C.c_uint32(12345)
    st = 12345
    for _ in range(64):
        f = L.orc_rand_next(C.byref(s))
        st = (1664525 * st + 1013904223) & 0xFFFFFFFF
        assert s.value == st
        assert f == np.float32(st & 0xFFFFFF) / np.float32(1 << 24)
        assert 0.0 <= f < 1.0


def test_frame_seeds_are_mt19937_every_16th():
    # first output of the 32-bit Mersenne twister with its standard default seed
    assert orc.frame_seeds(5489, 0, 1)[0] == 3499211612
    bg = np.random.MT19937()
    bg._legacy_seeding(1)                       # init_genrand(1), as std::mt19937(1)
    raw = bg.random_raw(16 * 40)
    assert np.array_equal(orc.frame_seeds(1, 0, 40), raw[::16].astype(np.uint32))
    assert np.array_equal(orc.frame_seeds(1, 7, 5), raw[7 * 16::16][:5].astype(np.uint32))


def _brute_force(prims, o, d, t_min, t_max):
    """Closest accepted hit over ALL triangles with the shader's triangle test in float32
    numpy (ray_tracing.hlsl:167-189), lowest primitive index on exact ties excluded from the check."""
    f = np.float32
    a = prims["vertices"]["pos"][:, 0].astype(f)
    e1 = prims["vertices"]["pos"][:, 1].astype(f) - a
    e2 = prims["vertices"]["pos"][:, 2].astype(f) - a
    d = d.astype(f)
    o = o.astype(f)

    def cross(u, v):
        return np.stack([u[..., 1] * v[..., 2] - u[..., 2] * v[..., 1],
                         u[..., 2] * v[..., 0] - u[..., 0] * v[..., 2],
                         u[..., 0] * v[..., 1] - u[..., 1] * v[..., 0]], -1).astype(f)

    def dot(u, v):
        return ((u[..., 0] * v[..., 0]).astype(f) + (u[..., 1] * v[..., 1]).astype(f)).astype(f) + (u[..., 2] * v[..., 2]).astype(f)
    q = cross(np.broadcast_to(d, e2.shape), e2)
    a1 = dot(e1, q)
    ok = ~(np.abs(a1) <= f(1e-4))
    with np.errstate(all="ignore"):
        s = ((o - a) / a1[:, None]).astype(f)
        r = cross(s, e1)
        t = dot(e2, r)
        bx = dot(s, q)
        by = dot(r, np.broadcast_to(d, r.shape))
        bz = (f(1) - bx - by).astype(f)
    ok &= (bx >= f(-1e-4)) & (by >= f(-1e-4)) & (bz >= f(-1e-4)) & (t >= f(1e-4)) & (t > f(t_min)) & (t < f(t_max))
    if not ok.any():
        return False, -1, None
    tt = np.where(ok, t, np.inf)
    i = int(np.argmin(tt))
    return True, i, float(tt[i])


@pytest.mark.parametrize("scene", ["scene01", "scene04"])
def test_traversal_equals_brute_force(golden, scene):
    g = golden[scene]
    sc = orc.OracleScene(g["prims"], g["tree"], g["cb"])
    rng = np.random.default_rng(7)
    lo, hi = g["bb"]["min"][0], g["bb"]["max"][0]
    n_hit = 0
    for k in range(300):
        o = (lo + (hi - lo) * rng.random(3)).astype(np.float32)
        d = rng.normal(size=3).astype(np.float32)
        d /= np.linalg.norm(d)
        if k % 10 == 0:
            d[k % 3] = 0.0                                   # axis-parallel component: 1/0 slabs
        hit, prim, bary, t, cnt = orc.intersect_one(sc, o, d)
        bh, bprim, bt = _brute_force(g["prims"], o, d, 1e-4, 10000.0)
        assert hit == bh
        if hit:
            n_hit += 1
            assert t == pytest.approx(bt, rel=0, abs=0)      # same float arithmetic -> same t
            assert cnt["node_visits"] >= 1 and cnt["rays"] == 1
    assert n_hit > 100


def _small_scene(golden, name, W, H):
    g = golden[name]
    sc = orc.OracleScene(g["prims"], g["tree"], g["cb"])
    alb = orc.albedo_table(g["cb"], True, kd_override=[list(m[0:3]) for m in g["mat_table"]] + [[0, 0, 0]] * (16 - len(g["mat_table"])))
    gb = orc.primary(sc, orc.default_camera(), W, H, alb)
    return sc, gb


def test_gbuffer_properties(golden):
    W, H = 96, 80
    sc, (dif, nrm, wpos) = _small_scene(golden, "scene01", W, H)
    matid = nrm[..., 3].view(np.int32)
    assert matid.min() >= 0 and matid.max() <= 7
    covered = matid > 0
    assert covered.mean() > 0.9                               # the box fills the view
    ln = np.linalg.norm(nrm[..., :3][covered], axis=-1)
    np.testing.assert_allclose(ln, 1.0, atol=1e-5)
    assert np.all(dif[..., 3][covered] == 255) and np.all(dif[~covered] == 0)
    # sRGB quirk: Kd 0.5 is stored as the sRGB-encoded byte 188, not 128
    out = (C.c_uint8 * 3)()
    orc.lib().orc_albedo_bytes((C.c_float * 3)(0.5, 0.5, 0.5), 1, out)
    assert list(out) == [188, 188, 188]
    orc.lib().orc_albedo_bytes((C.c_float * 3)(0.5, 0.5, 0.5), 0, out)
    assert list(out) == [128, 128, 128]


def test_literal_vs_linear_accumulation(golden):
    """The per-frame gamma round trip (reference) and the linear sum + one resolve agree to float
    rounding; the measured deviation is the documented tolerance between the two modes."""
    W, H, N = 64, 56, 24
    sc, gb = _small_scene(golden, "scene01", W, H)
    seeds = orc.frame_seeds(1, 0, N)
    old = np.zeros((H, W, 4), np.float32)
    acc = np.zeros((H, W, 4), np.float32)
    rays_a = rays_b = 0
    for f in range(N):
        old, outu, rad, ca = orc.trace_frame(sc, gb, f, seeds[f], old, want_radiance=True)
        cb_ = orc.trace_frame_linear(sc, gb, f, seeds[f], acc)
        rays_a += ca["rays"]; rays_b += cb_["rays"]
        assert np.all(rad >= 0) and np.isfinite(rad).all()
    assert rays_a == rays_b and rays_a > W * H * N
    lin, linu = orc.resolve(acc, N)
    assert np.all(acc[..., 3] == N)
    dev = np.abs(old[..., :3] - lin[..., :3]).max()
    assert dev < 2e-4, dev
    assert np.abs(outu.astype(int) - linu.astype(int)).max() <= 1
    assert np.all(old[..., 3] == 1.0)                         # alpha stays 1 through the recurrence
    assert old[..., :3].mean() > 0.05                         # the light is found


def test_frame_is_deterministic_and_seed_sensitive(golden):
    W, H = 48, 40
    sc, gb = _small_scene(golden, "scene04", W, H)
    z = np.zeros((H, W, 4), np.float32)
    a = orc.trace_frame(sc, gb, 0, 123, z, threads=1)[0]
    b = orc.trace_frame(sc, gb, 0, 123, z, threads=4)[0]
    c = orc.trace_frame(sc, gb, 0, 124, z)[0]
    assert a.tobytes() == b.tobytes()
    assert a.tobytes() != c.tobytes()


def test_max_segments_limits_the_path_length(golden):
    """max_segments = 0 and 14 are the reference's limit (itr reaches MAX_ITR * 3 after 14 rays,
    ray_tracing.hlsl:245-247,372-387); a smaller value ends every path after that many rays."""
    W, H = 48, 40
    sc, gb = _small_scene(golden, "scene04", W, H)
    g = golden["scene04"]
    z = np.zeros((H, W, 4), np.float32)
    ref = orc.trace_frame(sc, gb, 0, 123, z)
    same = orc.trace_frame(orc.OracleScene(g["prims"], g["tree"], g["cb"], max_segments=14), gb, 0, 123, z)
    assert ref[0].tobytes() == same[0].tobytes() and ref[3] == same[3]
    for cap in (1, 3):
        out = orc.trace_frame(orc.OracleScene(g["prims"], g["tree"], g["cb"], max_segments=cap), gb, 0, 123, z)
        assert out[3]["rays"] <= cap * out[3]["paths"]
        assert out[3]["rays"] < ref[3]["rays"]
