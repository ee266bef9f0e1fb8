// The reference's CPU transcendentals, restated operation for operation (round 5): Sleef 3.x expf_u10 / tanhf_u10 (FMA build) behind ATen's GELU(tanh) /
// SiLU vector kernels, ATen's Vectorized<float>::exp_u20 and glibc 2.35's expf behind its fp32 flash attention.  How each was established and the exhaustive
// checks: oracle/encoder_exact.c, profiles/r5_cpu_fp32_orders.txt.  Shared by csrc/encoder_exact.hip and csrc/gemm_fp32.hip (the exact GELU epilogue).
// Sources including this header are compiled with -ffp-contract=off: every FMA below is explicit.
#pragma once
#include "common.h"

namespace selftok {

// ---------------------------------------------------------------------------------------------------------------------------------
// Sleef 3.x expf_u10 / tanhf_u10 (FMA build), ATen exp_u20, glibc expf
// ---------------------------------------------------------------------------------------------------------------------------------
#define XE_R_LN2f 1.442695040888963407359924681001892137426645954152985934135449406931f
#define XE_L2Uf 0.693145751953125f
#define XE_L2Lf 1.428606765330187045e-06f
__device__ __forceinline__ float xe_pow2if(int q) { return __int_as_float((q + 0x7f) << 23); }
__device__ __forceinline__ float xe_ldexp2kf(float d, int e) { return d * xe_pow2if(e >> 1) * xe_pow2if(e - (e >> 1)); }

__device__ __forceinline__ float xe_sleef_expf(float d)
{
    const int q = (int)rintf(d * XE_R_LN2f);
    const float qf = (float)q;
    float s = fmaf(qf, -XE_L2Uf, d);
    s = fmaf(qf, -XE_L2Lf, s);
    float u = 0.000198527617612853646278381f;
    u = fmaf(u, s, 0.00139304355252534151077271f);
    u = fmaf(u, s, 0.00833336077630519866943359f);
    u = fmaf(u, s, 0.0416664853692054748535156f);
    u = fmaf(u, s, 0.166666671633720397949219f);
    u = fmaf(u, s, 0.5f);
    u = 1.0f + fmaf(s * s, u, s);
    u = xe_ldexp2kf(u, q);
    if (d < -104.0f) u = 0.0f;
    if (d > 104.0f) u = __builtin_inff();
    return u;
}

struct xf2 { float x, y; };
__device__ __forceinline__ xf2 dfadd2_f2_f(xf2 x, float y) { xf2 r; r.x = x.x + y; const float v = r.x - x.x; r.y = (x.x - (r.x - v)) + (y - v); r.y = r.y + x.y; return r; }
__device__ __forceinline__ xf2 dfadd2_f2_f2(xf2 x, xf2 y) { xf2 r; r.x = x.x + y.x; const float v = r.x - x.x; r.y = (x.x - (r.x - v)) + (y.x - v); r.y = r.y + (x.y + y.y); return r; }
__device__ __forceinline__ xf2 dfadd_f_f2(float x, xf2 y) { xf2 r; r.x = x + y.x; r.y = ((x - r.x) + y.x) + y.y; return r; }
__device__ __forceinline__ xf2 dfadd_f2_f2(xf2 x, xf2 y) { xf2 r; r.x = x.x + y.x; r.y = (((x.x - r.x) + y.x) + x.y) + y.y; return r; }
__device__ __forceinline__ xf2 dfmul_f2_f(xf2 x, float y) { xf2 r; r.x = x.x * y; r.y = fmaf(x.x, y, -r.x); r.y = fmaf(x.y, y, r.y); return r; }
__device__ __forceinline__ xf2 dfmul_f2_f2(xf2 x, xf2 y) { xf2 r; r.x = x.x * y.x; r.y = fmaf(x.x, y.x, -r.x); r.y = fmaf(x.y, y.x, r.y); r.y = fmaf(x.x, y.y, r.y); return r; }
__device__ __forceinline__ xf2 dfsqu_f2(xf2 x) { xf2 r; r.x = x.x * x.x; r.y = fmaf(x.x, x.x, -r.x); r.y = fmaf(x.x + x.x, x.y, r.y); return r; }
__device__ __forceinline__ xf2 dfrec_f2(xf2 d) { xf2 r; const float s = 1.0f / d.x; r.x = s; r.y = s * fmaf(-d.y, s, fmaf(-d.x, s, 1.0f)); return r; }
__device__ __forceinline__ xf2 dfdiv_f2_f2(xf2 n, xf2 d)
{
    xf2 q; const float t = 1.0f / d.x; q.x = n.x * t;
    const float u = fmaf(t, n.x, -q.x), v = fmaf(-d.y, t, fmaf(-d.x, t, 1.0f));
    q.y = fmaf(q.x, v, fmaf(n.y, t, u));
    return q;
}
__device__ __forceinline__ xf2 xe_expk2f(xf2 d)
{
    float u = (d.x + d.y) * XE_R_LN2f;
    const int q = (int)rintf(u);
    const float qf = (float)q;
    xf2 s = dfadd2_f2_f(d, qf * -XE_L2Uf);
    s = dfadd2_f2_f(s, qf * -XE_L2Lf);
    u = __uint_as_float(0x394fb7ffu);
    u = fmaf(u, s.x, __uint_as_float(0x3ab6bf7cu));
    u = fmaf(u, s.x, __uint_as_float(0x3c08890du));
    u = fmaf(u, s.x, __uint_as_float(0x3d2aaa5cu));
    xf2 t = dfadd2_f2_f(dfmul_f2_f(s, u), __uint_as_float(0x3e2aaaaau));
    t = dfadd2_f2_f(dfmul_f2_f2(s, t), 0.5f);
    t = dfadd2_f2_f2(s, dfmul_f2_f2(dfsqu_f2(s), t));
    t = dfadd_f_f2(1.0f, t);
    t.x = xe_ldexp2kf(t.x, q); t.y = xe_ldexp2kf(t.y, q);
    if (d.x < -104.0f) { t.x = 0.0f; t.y = 0.0f; }
    return t;
}
__device__ __forceinline__ float xe_sleef_tanhf(float x)
{
    float y = fabsf(x);
    const xf2 d0 = {y, 0.0f};
    xf2 d = xe_expk2f(d0);
    const xf2 e = dfrec_f2(d);
    const xf2 ne = {-e.x, -e.y};
    d = dfdiv_f2_f2(dfadd_f2_f2(d, ne), dfadd_f2_f2(d, e));
    y = d.x + d.y;
    if (fabsf(x) > 8.664339742f || y != y) y = 1.0f;
    y = __uint_as_float(__float_as_uint(y) ^ (__float_as_uint(x) & 0x80000000u));
    if (x != x) y = __uint_as_float(0xffffffffu);
    return y;
}
__device__ __forceinline__ float xe_gelu_tanh1(float v)
{
    const float kBeta = (float)(1.4142135623730950488 * 1.1283791670955125739 * 0.5), kKappa = (float)0.044715;
    const float cube = v * v * v;
    const float inner = kBeta * fmaf(kKappa, cube, v);
    return 0.5f * v * (1.0f + xe_sleef_tanhf(inner));
}
__device__ __forceinline__ float xe_silu1(float v) { return v / (1.0f + xe_sleef_expf(-v)); }

__device__ __forceinline__ float xe_exp_u20(float x)       // Vectorized<float>::exp_u20 (ATen/cpu/vec/vec512/vec512_float.h)
{
    const float f1 = 0.999999701f, f2 = 0.499991506f, f3 = 0.166676521f, f4 = 0.0418978221f, f5 = 0.00828929059f;
    const float log2e = __uint_as_float(0x3fb8aa3bu), ln2f = __uint_as_float(0x3f317218u);
    const float lmin = __uint_as_float(0xc2aeac50u), lmax = __uint_as_float(0x42b17218u);
    float src = x < lmax ? x : lmax;
    src = src > lmin ? src : lmin;
    const float fx = floorf(fmaf(src, log2e, 0.5f));
    const float r = fmaf(-fx, ln2f, src);
    float res = fmaf(r, f5, f4);
    res = fmaf(r, res, f3);
    res = fmaf(r, res, f2);
    res = fmaf(r, res, f1);
    res = fmaf(r, res, 1.0f);
    const int n1 = (int)rintf(fx - 1.0f);
    float two = __int_as_float((n1 + 127) << 23);
    if (x < lmin) two = 0.0f;
    res = res * two;
    return res * 2.0f;
}

static __constant__ unsigned long long XE_EXP2F_T[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
__device__ __forceinline__ float xe_expf_glibc(float x)     // glibc 2.35 expf: `std::exp(float)` of the flash kernel's rescale
{
    if (x != x) return x;
    if (x > 0x1.62e42ep6f) return __builtin_inff();
    if (x < -0x1.9fe368p6f) return 0.f;
    const double InvLn2N = 0x1.71547652b82fep+0 * 32, Shift = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
    const double z = InvLn2N * (double)x;
    double kd = z + Shift;
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= Shift;
    const double r = z - kd;
    const unsigned long long t = XE_EXP2F_T[ki % 32] + (ki << 47);
    const double s = __longlong_as_double((long long)t);
    const double zz = fma(C0, r, C1), r2 = r * r;
    double y = fma(C2, r, 1.0);
    y = fma(zz, r2, y);
    return (float)(y * s);
}

}  // namespace selftok
