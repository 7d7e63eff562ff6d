/*
 * oracle_math.h -- TEST INFRASTRUCTURE (CPU oracle). Not part of the product path.
 *
 * Pinned, bit-reproducible replacements for the CUDA libdevice transcendentals the
 * reference kernel calls (rsqrtf, logf, cosf, expf, powf: GPURaytracer.cu:15,643,693,
 * 809,831-832).  NVIDIA's implementations are not reproducible off NVIDIA hardware
 * (SURVEY.md section 8a-Q13), so the build pins its own definition ("detmath v1",
 * DESIGN.md section 3) and implements it twice, independently: here in C for the
 * oracle, and in raytracer_amd/csrc/rt_detmath.h for the HIP kernels.  The two must
 * agree bit for bit; tests/test_detmath_gpu.py checks that on the GPU and
 * tests/test_oracle_math.py pins this copy against correctly rounded libm results.
 *
 * Definition: every function evaluates a fixed polynomial in IEEE binary64 using only
 * + - * / (no fused multiply-add: compile with -ffp-contract=off) and rounds once to
 * binary32.  All integer <-> floating conversions are exact.
 */
#ifndef ORACLE_MATH_H
#define ORACLE_MATH_H

#include <stdint.h>
#include <string.h>
#include <math.h>

static inline uint64_t om_bits64(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double om_from_bits64(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

#define OM_LN2_HI 6.93147180369123816490e-01 /* 0x3fe62e42fee00000 */
#define OM_LN2_LO 1.90821492927058770002e-10 /* 0x3dea39ef35793c76 */
#define OM_INV_LN2 1.44269504088896338700e+00
#define OM_PIO2_HI 1.57079632673412561417e+00 /* first 33 bits of pi/2 */
#define OM_PIO2_LO 6.07710050650619224932e-11 /* pi/2 - OM_PIO2_HI */
#define OM_2_OVER_PI 6.36619772367581382433e-01

/* natural log of a finite, strictly positive, NORMAL double */
static inline double om_log_core(double x)
{
    uint64_t b = om_bits64(x);
    int e = (int)(b >> 52) - 1023;
    double m = om_from_bits64((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull); /* [1,2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    /* 2*atanh(s) = 2*s*(1 + z/3 + z^2/5 + ... + z^9/19), Horner from the top */
    double p = 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    double r = (2.0 * s) * p;
    double de = (double)e;
    return de * OM_LN2_HI + (r + de * OM_LN2_LO);
}

/* exp of a double in [-745, 709] */
static inline double om_exp_core(double x)
{
    double kf = floor(x * OM_INV_LN2 + 0.5);
    int k = (int)kf;
    double r = (x - kf * OM_LN2_HI) - kf * OM_LN2_LO;
    /* Taylor to r^13/13!, Horner from the top */
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    /* scale by 2^k in two exact steps so that k down to -1074 stays representable */
    int k1 = k / 2;
    int k2 = k - k1;
    double s1 = om_from_bits64((uint64_t)(k1 + 1023) << 52);
    double s2 = om_from_bits64((uint64_t)(k2 + 1023) << 52);
    return (p * s1) * s2;
}

static inline float om_rsqrtf(float x) { return 1.0f / sqrtf(x); }

static inline float om_logf(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    return (float)om_log_core((double)x);
}

static inline float om_expf(float x)
{
    if (x != x) return x;
    if (x > 89.0f) return INFINITY;
    if (x < -104.0f) return 0.0f;
    return (float)om_exp_core((double)x);
}

/* cosine; argument reduction is exact enough for |x| <= 1e5 (the kernel only
 * ever passes theta in [0, 6.2832]) */
static inline float om_cosf(float x)
{
    if (x != x || x == INFINITY || x == -INFINITY) return NAN;
    double xd = (double)x;
    double kf = floor(xd * OM_2_OVER_PI + 0.5);
    double r = (xd - kf * OM_PIO2_HI) - kf * OM_PIO2_LO;
    int n = (int)((long long)kf & 3);
    double z = r * r;
    double v;
    if (n == 0 || n == 2) {
        /* cos r = 1 - z/2! + z^2/4! - ... + z^8/16! */
        double p = 1.0 / 20922789888000.0;
        p = p * z - 1.0 / 87178291200.0;
        p = p * z + 1.0 / 479001600.0;
        p = p * z - 1.0 / 3628800.0;
        p = p * z + 1.0 / 40320.0;
        p = p * z - 1.0 / 720.0;
        p = p * z + 1.0 / 24.0;
        p = p * z - 0.5;
        p = p * z + 1.0;
        v = p;
    } else {
        /* sin r = r*(1 - z/3! + z^2/5! - ... + z^8/17!) */
        double p = 1.0 / 355687428096000.0;
        p = p * z - 1.0 / 1307674368000.0;
        p = p * z + 1.0 / 6227020800.0;
        p = p * z - 1.0 / 39916800.0;
        p = p * z + 1.0 / 362880.0;
        p = p * z - 1.0 / 5040.0;
        p = p * z + 1.0 / 120.0;
        p = p * z - 1.0 / 6.0;
        p = p * z + 1.0;
        v = r * p;
    }
    if (n == 1 || n == 2) v = -v;
    return (float)v;
}

/* powf for the two call sites of the kernel (base in [0,1], exponent 5 or 1000);
 * defined for every base >= 0; a negative base returns NaN (never reached: both
 * call sites clamp the base with max(0, .)). */
static inline float om_powf(float x, float y)
{
    if (x != x || y != y) return NAN;
    if (y == 0.0f) return 1.0f;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return y > 0.0f ? 0.0f : INFINITY;
    if (x == INFINITY) return y > 0.0f ? INFINITY : 0.0f;
    if (x == 1.0f) return 1.0f;
    double p = (double)y * om_log_core((double)x);
    if (p > 89.0) return INFINITY;
    if (p < -104.0) return 0.0f;
    return (float)om_exp_core(p);
}

/* CUDA max()/min() on floats are fmaxf/fminf: a NaN operand is dropped
 * (GPURaytracer.cu:290-311).  Pinned select form, identical in the HIP copy. */
static inline float om_fmaxf(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    return a > b ? a : b;
}
static inline float om_fminf(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    return a < b ? a : b;
}

#endif
