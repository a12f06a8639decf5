// Canonical elementary functions of the engine (device + host), see DESIGN.md "Canonical arithmetic".
//
// The reference evaluates these inside libtorch (Snake1d.cs:52 sin, Decoder.cs:46 Tanh,
// Utils/TorchUtils.cs:26-30 ELU).  To make results reproducible bit-for-bit between gfx950 and
// any IEEE-754 host, each function is ONE fixed sequence of binary32 +,-,*,/,fma,rint
// (no libm / ocml calls, compiled with -ffp-contract=off so nothing is re-associated or fused).
// Coefficients: tools/fit_math_poly.py.  Accuracy: sin <= 1.9 ulp(1) on |x|<=40; exp/tanh <= 2 ulp.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NC_HD __host__ __device__ __forceinline__
#else
#define NC_HD inline
#endif

NC_HD float nc_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// The order of ATen's argmin, which the reference's quantizers call on the distance matrix (Modules/DAC/VectorQuantizer.cs:121,
// Modules/SNAC/VectorQuantizer.cs:137, Modules/Encodec/EuclideanCodebook.cs:181): a NaN distance beats every number, and among equal
// distances (and among NaNs) the LOWER index wins -- so a partly-NaN row yields its first NaN, an all-equal row index 0.
// nc_argmin_before: (d, i) precedes (bd, bi) in that order (any visiting order).  nc_argmin_scan: the same test for a scan that visits
// indices in ASCENDING order, where an equal distance never replaces the incumbent.
NC_HD bool nc_argmin_before(float d, int i, float bd, int bi) {
    const bool dn = d != d, bn = bd != bd;
    return (dn || bn) ? (dn && (!bn || i < bi)) : (d < bd || (d == bd && i < bi));
}
NC_HD bool nc_argmin_scan(float d, float bd) { return !(d >= bd) && bd == bd; }

NC_HD float nc_sinf(float x) {
    float n = __builtin_rintf(x * 0x1.45f306p-2f);  // x * fl(1/pi)
    float r = nc_fma(n, -3.140625f, x);             // Cody-Waite, pi = A + B + C
    r = nc_fma(n, -9.67502593994140625e-4f, r);
    r = nc_fma(n, -1.509957990978376432e-7f, r);
    float u = r * r;
    float p = -0x1.9d5778p-26f;
    p = nc_fma(p, u, 0x1.71936ap-19f);
    p = nc_fma(p, u, -0x1.a018f4p-13f);
    p = nc_fma(p, u, 0x1.111110p-7f);
    p = nc_fma(p, u, -0x1.555556p-3f);
    float s = nc_fma(r * u, p, r);
    int ni = (int)n;
    return (ni & 1) ? -s : s;
}

NC_HD float nc_expf(float x) {
    if (x > 88.0f) x = 88.0f;
    if (x < -87.0f) x = -87.0f;
    float n = __builtin_rintf(x * 0x1.715476p+0f);  // x * fl(log2 e)
    float r = nc_fma(n, -0x1.62e400p-1f, x);
    r = nc_fma(n, -0x1.7f7d1cp-20f, r);
    float q = 0x1.6d5accp-10f;
    q = nc_fma(q, r, 0x1.121f36p-7f);
    q = nc_fma(q, r, 0x1.5554d8p-5f);
    q = nc_fma(q, r, 0x1.5554cap-3f);
    q = nc_fma(q, r, 0x1.000000p-1f);
    float e = nc_fma(r * r, q, r) + 1.0f;
    int32_t ni = (int32_t)n;
    uint32_t bits = (uint32_t)(ni + 127) << 23;
    float sc = __builtin_bit_cast(float, bits);
    return e * sc;
}

NC_HD float nc_tanhf(float x) {
    float ax = __builtin_fabsf(x);
    if (ax < 0.55f) {
        float u = x * x;
        float p = -0x1.ad2786p-8f;
        p = nc_fma(p, u, 0x1.5c97c6p-6f);
        p = nc_fma(p, u, -0x1.b99508p-5f);
        p = nc_fma(p, u, 0x1.110fc6p-3f);
        p = nc_fma(p, u, -0x1.555554p-2f);
        return nc_fma(x * u, p, x);
    }
    float t;
    if (ax > 9.0f) {
        t = 1.0f;
    } else {
        float e = nc_expf(2.0f * ax);
        t = 1.0f - 2.0f / (e + 1.0f);
    }
    return __builtin_copysignf(t, x);
}

// Snake1d.cs:52  where(alpha == 0, x, addcdiv(x, sin(alpha*x)^2, alpha, 1)), with the division folded into one
// correctly-rounded reciprocal per channel: inv = nc_snake_inv(alpha) = fl(1/alpha), 0 when alpha == 0 (then the
// product term is exactly 0 and the result is x).  Branch-free; <= 1 ulp of the quotient away from the true division.
NC_HD float nc_snake_inv(float alpha) { return alpha == 0.0f ? 0.0f : 1.0f / alpha; }
NC_HD float nc_snakef(float x, float alpha, float inv) {
    float s = nc_sinf(alpha * x);
    return x + (s * s) * inv;
}

// ELU (alpha = 1) and the logistic function of the Encodec layers, on the canonical exp (SEANetEncoder.cs ELU, SLSTM.cs gates)
NC_HD float nc_eluf(float x) { return x > 0.0f ? x : nc_expf(x) - 1.0f; }
NC_HD float nc_sigmoidf(float x) { return 1.0f / (1.0f + nc_expf(-x)); }

#if defined(__HIPCC__)
// Two-at-a-time forms for the device kernels: the SAME operation sequences on a pair of values, written on 2-vectors so that the
// multiplies / fmas / adds compile to the packed fp32 instructions of gfx950 (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two
// IEEE operations per lane and issue slot -- the Snake activation is ~25 vector instructions per element and bounds the depthwise
// kernel and the residual units' epilogues).  Bit-identical to nc_sinf / nc_snakef per element.
typedef float nc_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ nc_f2 nc_fma2(nc_f2 a, nc_f2 b, nc_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ nc_f2 nc_sinf2(nc_f2 x) {
    const nc_f2 n = __builtin_elementwise_rint(x * 0x1.45f306p-2f);
    nc_f2 r = nc_fma2(n, (nc_f2)(-3.140625f), x);
    r = nc_fma2(n, (nc_f2)(-9.67502593994140625e-4f), r);
    r = nc_fma2(n, (nc_f2)(-1.509957990978376432e-7f), r);
    const nc_f2 u = r * r;
    nc_f2 p = (nc_f2)(-0x1.9d5778p-26f);
    p = nc_fma2(p, u, (nc_f2)(0x1.71936ap-19f));
    p = nc_fma2(p, u, (nc_f2)(-0x1.a018f4p-13f));
    p = nc_fma2(p, u, (nc_f2)(0x1.111110p-7f));
    p = nc_fma2(p, u, (nc_f2)(-0x1.555556p-3f));
    nc_f2 s = nc_fma2(r * u, p, r);
    const int n0 = (int)n[0], n1 = (int)n[1];
    s[0] = (n0 & 1) ? -s[0] : s[0];
    s[1] = (n1 & 1) ? -s[1] : s[1];
    return s;
}
// exp / ELU of a pair: nc_expf's operation sequence per component (clamps, range reduction, polynomial, scale by 2^n), the multiplies /
// fmas / adds packed.  Bit-identical to nc_expf / nc_eluf per element, NaNs included (the clamps and the final choice are compare +
// select per component, exactly as in the scalar form).  The Encodec input mode applies an ELU to every element a convolution stages;
// next to a matrix-core stream those vector instructions are issue time the matrix pipe does not get back (DESIGN 8 rounds 4-6).
__device__ __forceinline__ nc_f2 nc_expf2(nc_f2 x) {
    x[0] = x[0] > 88.0f ? 88.0f : x[0];
    x[1] = x[1] > 88.0f ? 88.0f : x[1];
    x[0] = x[0] < -87.0f ? -87.0f : x[0];
    x[1] = x[1] < -87.0f ? -87.0f : x[1];
    const nc_f2 n = __builtin_elementwise_rint(x * 0x1.715476p+0f);
    nc_f2 r = nc_fma2(n, (nc_f2)(-0x1.62e400p-1f), x);
    r = nc_fma2(n, (nc_f2)(-0x1.7f7d1cp-20f), r);
    nc_f2 q = (nc_f2)(0x1.6d5accp-10f);
    q = nc_fma2(q, r, (nc_f2)(0x1.121f36p-7f));
    q = nc_fma2(q, r, (nc_f2)(0x1.5554d8p-5f));
    q = nc_fma2(q, r, (nc_f2)(0x1.5554cap-3f));
    q = nc_fma2(q, r, (nc_f2)(0x1.000000p-1f));
    const nc_f2 e = nc_fma2(r * r, q, r) + 1.0f;
    const int32_t n0 = (int32_t)n[0], n1 = (int32_t)n[1];
    nc_f2 sc;
    sc[0] = __builtin_bit_cast(float, (uint32_t)(n0 + 127) << 23);
    sc[1] = __builtin_bit_cast(float, (uint32_t)(n1 + 127) << 23);
    return e * sc;
}
__device__ __forceinline__ nc_f2 nc_eluf2(nc_f2 x) {
    const nc_f2 e = nc_expf2(x) - 1.0f;
    nc_f2 y;
    y[0] = x[0] > 0.0f ? x[0] : e[0];
    y[1] = x[1] > 0.0f ? x[1] : e[1];
    return y;
}
__device__ __forceinline__ nc_f2 nc_snakef2(nc_f2 x, nc_f2 alpha, nc_f2 inv) {
    const nc_f2 s = nc_sinf2(alpha * x);
    return x + (s * s) * inv;
}
// The same Snake with the sine taken up to its sign: (-s) * (-s) == s * s exactly in IEEE-754, so the parity select of nc_sinf2 (two
// conversions, masks, compares and selects: 8 of the ~30 vector instructions of a pair) is dead weight under the square -- bit-identical
// to nc_snakef2.  Used by the vector-ALU-bound SNAC kernels (depthwise convolution, fused residual unit) and, from round 5, by the
// conv template (XV-only instances: staging and epilogue, -0.35 ms on the DAC step; legacy instances: the in-loop staging, as
// nc_snakef2_m_rows); in the pointwise kernel's epilogue it is neutral (round 5) and nc_snakef2 stays.
__device__ __forceinline__ nc_f2 nc_snakef2_m(nc_f2 x, nc_f2 alpha, nc_f2 inv) {
    const nc_f2 ax = alpha * x;
    const nc_f2 n = __builtin_elementwise_rint(ax * 0x1.45f306p-2f);
    nc_f2 r = nc_fma2(n, (nc_f2)(-3.140625f), ax);
    r = nc_fma2(n, (nc_f2)(-9.67502593994140625e-4f), r);
    r = nc_fma2(n, (nc_f2)(-1.509957990978376432e-7f), r);
    const nc_f2 u = r * r;
    nc_f2 p = (nc_f2)(-0x1.9d5778p-26f);
    p = nc_fma2(p, u, (nc_f2)(0x1.71936ap-19f));
    p = nc_fma2(p, u, (nc_f2)(-0x1.a018f4p-13f));
    p = nc_fma2(p, u, (nc_f2)(0x1.111110p-7f));
    p = nc_fma2(p, u, (nc_f2)(-0x1.555556p-3f));
    const nc_f2 s = nc_fma2(r * u, p, r);
    return x + (s * s) * inv;
}
// N independent pairs, the chain written step by step ACROSS the pairs: every packed instruction is followed by N - 1 independent ones, so
// none waits on its predecessor (pair by pair the compiler puts an `s_nop` between each two instructions of a chain).  Same operations
// per element in the same order as nc_snakef2_m: bit-identical.
template <int N, bool SIGNED = false>
__device__ __forceinline__ void nc_snakef2_m_rows(nc_f2 (&x)[N], const nc_f2 (&alpha)[N], const nc_f2 (&inv)[N]) {
    nc_f2 ax[N], n[N], r[N], u[N], p[N], s[N];
#pragma unroll
    for (int i = 0; i < N; ++i) ax[i] = alpha[i] * x[i];
#pragma unroll
    for (int i = 0; i < N; ++i) n[i] = __builtin_elementwise_rint(ax[i] * 0x1.45f306p-2f);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = nc_fma2(n[i], (nc_f2)(-3.140625f), ax[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = nc_fma2(n[i], (nc_f2)(-9.67502593994140625e-4f), r[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = nc_fma2(n[i], (nc_f2)(-1.509957990978376432e-7f), r[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) u[i] = r[i] * r[i];
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = nc_fma2((nc_f2)(-0x1.9d5778p-26f), u[i], (nc_f2)(0x1.71936ap-19f));
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = nc_fma2(p[i], u[i], (nc_f2)(-0x1.a018f4p-13f));
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = nc_fma2(p[i], u[i], (nc_f2)(0x1.111110p-7f));
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = nc_fma2(p[i], u[i], (nc_f2)(-0x1.555556p-3f));
#pragma unroll
    for (int i = 0; i < N; ++i) s[i] = nc_fma2(r[i] * u[i], p[i], r[i]);
    if constexpr (SIGNED) {   // (the canonical sine's parity select, kept for A/B runs: dead under the square)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int n0 = (int)n[i][0], n1 = (int)n[i][1];
            s[i][0] = (n0 & 1) ? -s[i][0] : s[i][0];
            s[i][1] = (n1 & 1) ? -s[i][1] : s[i][1];
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = x[i] + (s[i] * s[i]) * inv[i];
}
// in-place on two scalars
__device__ __forceinline__ void nc_snake_pair_m(float& x0, float& x1, float a0, float i0, float a1, float i1) {
    const nc_f2 r = nc_snakef2_m(nc_f2{x0, x1}, nc_f2{a0, a1}, nc_f2{i0, i1});
    x0 = r[0];
    x1 = r[1];
}
__device__ __forceinline__ void nc_snake_pair(float& x0, float& x1, float a0, float i0, float a1, float i1) {
    const nc_f2 r = nc_snakef2(nc_f2{x0, x1}, nc_f2{a0, a1}, nc_f2{i0, i1});
    x0 = r[0];
    x1 = r[1];
}
#endif
