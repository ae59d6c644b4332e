/*
 * Numerical contract of the path tracer: the elementary float functions whose
 * precision HLSL leaves implementation-defined (sin, cos, pow, normalize,
 * reflect, min/max), pinned here to ONE deterministic meaning built only from
 * IEEE-754 binary32 +, -, *, /, sqrt and fused multiply-add.  The same source
 * compiles for the host (gcc/g++) and for gfx950 (hipcc), so a CPU check and
 * the GPU kernels can agree bit for bit.
 *
 * Build every translation unit that includes this file with
 * -ffp-contract=off and without any fast-math option: a fused or re-associated
 * expression changes results.  Fusion happens only where fmaf() is written.
 *
 * Shading expressions that use these helpers follow the reference's
 * ray_tracing.hlsl / helpers.hlsli; each helper cites the HLSL construct it
 * gives a meaning to.
 */
#ifndef PTAMD_MATH_H
#define PTAMD_MATH_H

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define PT_HD __host__ __device__ __forceinline__
#else
#define PT_HD static inline
#endif

/* HLSL `#define MY_PI 3.1415926` (helpers.hlsli:2), a float literal. */
#define PT_MY_PI 3.1415926f
/* kEpsilon (ray_tracing.hlsl:44) */
#define PT_EPSILON 1e-4f

typedef struct pt_f3 { float x, y, z; } pt_f3;

PT_HD uint32_t pt_f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
PT_HD float    pt_u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

/* HLSL min/max: if one operand is NaN the other is returned. */
PT_HD float pt_min(float a, float b) { return fminf(a, b); }
PT_HD float pt_max(float a, float b) { return fmaxf(a, b); }

PT_HD pt_f3 pt_mk3(float x, float y, float z) { pt_f3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_HD pt_f3 pt_add3(pt_f3 a, pt_f3 b) { return pt_mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_HD pt_f3 pt_sub3(pt_f3 a, pt_f3 b) { return pt_mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_HD pt_f3 pt_mul3(pt_f3 a, pt_f3 b) { return pt_mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_HD pt_f3 pt_scale3(pt_f3 a, float s) { return pt_mk3(a.x * s, a.y * s, a.z * s); }
/* HLSL dot(): products summed left to right, no fusion. */
PT_HD float pt_dot3(pt_f3 a, pt_f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* HLSL cross(). */
PT_HD pt_f3 pt_cross3(pt_f3 a, pt_f3 b)
{
    return pt_mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* HLSL normalize(): v * rsqrt(dot(v,v)), with rsqrt(x) := 1/sqrt(x) in IEEE
 * arithmetic.  normalize(0) is NaN, as in HLSL; it is not guarded. */
PT_HD pt_f3 pt_normalize3(pt_f3 v)
{
    float inv = 1.0f / sqrtf(pt_dot3(v, v));
    return pt_scale3(v, inv);
}
/* HLSL reflect(i, n) = i - 2 * n * dot(i, n); n need not be unit length. */
PT_HD pt_f3 pt_reflect3(pt_f3 i, pt_f3 n)
{
    float d2 = 2.0f * pt_dot3(i, n);
    return pt_mk3(i.x - d2 * n.x, i.y - d2 * n.y, i.z - d2 * n.z);
}

/* ---- sin / cos ---------------------------------------------------------
 * Cody-Waite reduction by pi/2 (three-term split, exact for |k| < 2^9) and
 * the classic single-precision minimax kernels on [-pi/4, pi/4].  Intended
 * range |x| <= ~800; the shading code only passes angles in [0, 2*pi].
 * Max error ~1 ulp of the result magnitude near 1 (checked against libm in
 * the CPU test-suite). */
PT_HD void pt_sincos_reduce(float x, float *r, int *q)
{
    float kf = floorf(fmaf(x, 0.636619772367581343f, 0.5f));
    float t = fmaf(-kf, 1.57073974609375f, x);            /* pi/2 hi  */
    t = fmaf(-kf, 5.657970905303955078125e-05f, t);       /* pi/2 mid */
    t = fmaf(-kf, 9.920936294705029468e-10f, t);          /* pi/2 lo  */
    *r = t;
    *q = ((int)kf) & 3;
}
PT_HD float pt_sin_kernel(float r)
{
    float s = r * r;
    float p = fmaf(-1.9515295891e-4f, s, 8.3321608736e-3f);
    p = fmaf(p, s, -1.6666654611e-1f);
    return fmaf(p * s, r, r);
}
PT_HD float pt_cos_kernel(float r)
{
    float s = r * r;
    float p = fmaf(2.443315711809948e-5f, s, -1.388731625493765e-3f);
    p = fmaf(p, s, 4.166664568298827e-2f);
    return fmaf(p, s * s, fmaf(-0.5f, s, 1.0f));
}
PT_HD float pt_sinf(float x)
{
    float r; int q;
    pt_sincos_reduce(x, &r, &q);
    float v = (q & 1) ? pt_cos_kernel(r) : pt_sin_kernel(r);
    return (q & 2) ? -v : v;
}
PT_HD float pt_cosf(float x)
{
    float r; int q;
    pt_sincos_reduce(x, &r, &q);
    float v = (q & 1) ? pt_sin_kernel(r) : pt_cos_kernel(r);
    return ((q + 1) & 2) ? -v : v;
}

/* ---- exp2 / log2 / pow -------------------------------------------------- */
/* 2^k for integer k in [-126, 127] */
PT_HD float pt_pow2i(int k) { return pt_u2f((uint32_t)(k + 127) << 23); }

PT_HD float pt_exp2f(float x)
{
    if (!(x == x)) return x;                      /* NaN */
    if (x >= 128.0f) return pt_u2f(0x7f800000u);  /* +inf */
    if (x < -150.0f) return 0.0f;                 /* includes -inf */
    float kf = floorf(x + 0.5f);
    float r = x - kf;                             /* exact, in [-0.5, 0.5] */
    float p = 1.535920892e-4f;
    p = fmaf(p, r, 1.339262701e-3f);
    p = fmaf(p, r, 9.618384764e-3f);
    p = fmaf(p, r, 5.550347269e-2f);
    p = fmaf(p, r, 2.402264476e-1f);
    p = fmaf(p, r, 6.931471825e-1f);
    p = fmaf(p, r, 1.0f);
    int k = (int)kf;                              /* in [-150, 128] */
    int k1 = k >> 1;
    return (p * pt_pow2i(k1)) * pt_pow2i(k - k1);
}

PT_HD float pt_log2f(float x)
{
    if (!(x == x)) return x;                      /* NaN */
    if (x < 0.0f) return pt_u2f(0x7fc00000u);     /* NaN */
    if (x == 0.0f) return pt_u2f(0xff800000u);    /* -inf */
    if (x == pt_u2f(0x7f800000u)) return x;       /* +inf */
    int e = 0;
    if (x < 1.17549435e-38f) { x = x * 4294967296.0f; e = -32; }   /* subnormal */
    uint32_t u = pt_f2u(x);
    /* split x = m * 2^ex with m in [0.75, 1.5) */
    uint32_t adj = u - 0x3f400000u;               /* bits(0.75f) */
    int ex = (int)adj >> 23;                      /* floor, arithmetic shift */
    float m = pt_u2f(u - ((uint32_t)ex << 23));   /* in [0.75, 1.5) */
    float ef = (float)(e + ex);
    float t = (m - 1.0f) / (m + 1.0f);
    float t2 = t * t;
    float p = 0.4374088347e+0f;
    p = fmaf(p, t2, 0.5764843822e+0f);
    p = fmaf(p, t2, 0.9618024230e+0f);
    float lo = fmaf(t, 2.8853900432586669921875f, ef);
    return fmaf(t2 * t, p, lo);
}

/* HLSL pow(x, y) = exp2(y * log2(x)): pow(0, y>0) = 0, negative base = NaN. */
PT_HD float pt_powf(float x, float y) { return pt_exp2f(y * pt_log2f(x)); }

#endif /* PTAMD_MATH_H */
