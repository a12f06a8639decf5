/* ORACLE (test infrastructure; never linked into the product library).
 *
 * Canonical elementary functions of the parity spec (DESIGN.md "Canonical arithmetic").
 * The reference evaluates sin/tanh/exp inside libtorch (Snake1d.cs:52 `sin`, Decoder.cs:46 `Tanh`,
 * TorchUtils.cs:26-30 `ELU`); libm/Sleef/ocml all differ from each other in the last ulp, so the
 * spec fixes ONE definition built only from IEEE-754 binary32 +,-,*,/,fma,rint — every conforming
 * implementation (this file on the host, the HIP kernels on gfx950) returns identical bits.
 * Accuracy vs the real functions: sin <= 1.9 ulp(1.0) for |x| <= 40, tanh/exp <= 2 ulp.
 * Coefficients: tools/fit_math_poly.py.
 */
#ifndef NC_REF_MATH_H
#define NC_REF_MATH_H
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float ref_sinf(float x) {
    float n = rintf(x * 0x1.45f306p-2f);                 /* x * fl(1/pi) */
    float r = fmaf(n, -3.140625f, x);                    /* 3-term Cody-Waite, pi = A+B+C */
    r = fmaf(n, -9.67502593994140625e-4f, r);
    r = fmaf(n, -1.509957990978376432e-7f, r);
    float u = r * r;
    float p = -0x1.9d5778p-26f;
    p = fmaf(p, u, 0x1.71936ap-19f);
    p = fmaf(p, u, -0x1.a018f4p-13f);
    p = fmaf(p, u, 0x1.111110p-7f);
    p = fmaf(p, u, -0x1.555556p-3f);
    float s = fmaf(r * u, p, r);
    int ni = (int)n;
    return (ni & 1) ? -s : s;
}

/* exp(x) for x in [-87, 88]; outside is clamped. */
static inline float ref_expf(float x) {
    if (x > 88.0f) x = 88.0f;
    if (x < -87.0f) x = -87.0f;
    float n = rintf(x * 0x1.715476p+0f);                 /* x * fl(log2 e) */
    float r = fmaf(n, -0x1.62e400p-1f, x);               /* ln2 hi (exact in 15 bits) */
    r = fmaf(n, -0x1.7f7d1cp-20f, r);                    /* ln2 lo */
    float q = 0x1.6d5accp-10f;
    q = fmaf(q, r, 0x1.121f36p-7f);
    q = fmaf(q, r, 0x1.5554d8p-5f);
    q = fmaf(q, r, 0x1.5554cap-3f);
    q = fmaf(q, r, 0x1.000000p-1f);
    float e = fmaf(r * r, q, r) + 1.0f;
    int32_t ni = (int32_t)n;
    uint32_t bits = (uint32_t)(ni + 127) << 23;          /* 2^n, n in [-126,127] */
    float sc;
    memcpy(&sc, &bits, 4);
    return e * sc;
}

static inline float ref_tanhf(float x) {
    float ax = fabsf(x);
    if (ax < 0.55f) {
        float u = x * x;
        float p = -0x1.ad2786p-8f;
        p = fmaf(p, u, 0x1.5c97c6p-6f);
        p = fmaf(p, u, -0x1.b99508p-5f);
        p = fmaf(p, u, 0x1.110fc6p-3f);
        p = fmaf(p, u, -0x1.555554p-2f);
        return fmaf(x * u, p, x);
    }
    float t;
    if (ax > 9.0f) {
        t = 1.0f;
    } else {
        float e = ref_expf(2.0f * ax);
        t = 1.0f - 2.0f / (e + 1.0f);
    }
    return copysignf(t, x);
}

/* Snake1d.cs:52  where(alpha == 0, x, addcdiv(x, sin(alpha*x)^2, alpha, 1)) */
/* Snake1d.cs:52  where(alpha == 0, x, x + sin(alpha*x)^2 / alpha), restated with the division folded into one
 * correctly-rounded reciprocal per channel (canonical arithmetic): r = fl(1/alpha), or 0 when alpha == 0
 * (then sin(0*x)^2 * 0 == 0 and the result is x); y = x + fl(fl(s*s) * r).  Differs from the true quotient by <= 1 ulp of
 * the quotient. */
static inline float ref_snake_inv(float alpha) { return alpha == 0.0f ? 0.0f : 1.0f / alpha; }
static inline float ref_snakef(float x, float alpha) {
    float s = ref_sinf(alpha * x);
    return x + (s * s) * ref_snake_inv(alpha);
}

#endif
