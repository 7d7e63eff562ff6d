// rt_detmath.h -- pinned device math ("detmath v1").
//
// The reference kernel calls CUDA libdevice rsqrtf/logf/cosf/expf/powf
// (processor/GPU/GPURaytracer.cu:15,643,693,809,831-832).  Those are not reproducible off
// NVIDIA hardware, and a one-ulp difference in a bounce direction can flip a hit at 1 spp, so
// this build defines its own: fixed polynomials evaluated in binary64 with separate multiply
// and add (the library is compiled with -ffp-contract=off), rounded once to binary32.  The
// results agree with correctly rounded libm values, and bit for bit with the oracle's (oracle/oracle_math.h;
// tests/test_parity_gpu.py::test_detmath_matches_oracle_bit_for_bit).  MI355X runs
// FP64 vector math at half the FP32 rate, so these cost little next to traversal.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtd {

__device__ __forceinline__ double f64_from_bits(uint64_t u) { return __longlong_as_double((long long)u); }
__device__ __forceinline__ uint64_t f64_bits(double d) { return (uint64_t)__double_as_longlong(d); }

constexpr double kLn2Hi = 6.93147180369123816490e-01;
constexpr double kLn2Lo = 1.90821492927058770002e-10;
constexpr double kInvLn2 = 1.44269504088896338700e+00;
constexpr double kPio2Hi = 1.57079632673412561417e+00;
constexpr double kPio2Lo = 6.07710050650619224932e-11;
constexpr double kTwoOverPi = 6.36619772367581382433e-01;

// ln(x) for finite positive normal x: x = m*2^e with m in (sqrt(1/2), sqrt(2)], 2*atanh series
__device__ __forceinline__ double log_core(double x)
{
    const uint64_t b = f64_bits(x);
    int e = (int)(b >> 52) - 1023;
    double m = f64_from_bits((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
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
    const double r = (2.0 * s) * p;
    const double de = (double)e;
    return de * kLn2Hi + (r + de * kLn2Lo);
}

__device__ __forceinline__ double exp_core(double x)
{
    const double kf = floor(x * kInvLn2 + 0.5);
    const int k = (int)kf;
    const double r = (x - kf * kLn2Hi) - kf * kLn2Lo;
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
    const int k1 = k / 2;
    const int k2 = k - k1;
    const double s1 = f64_from_bits((uint64_t)(k1 + 1023) << 52);
    const double s2 = f64_from_bits((uint64_t)(k2 + 1023) << 52);
    return (p * s1) * s2;
}

__device__ __forceinline__ float rsqrt_f(float x) { return 1.0f / __builtin_sqrtf(x); }

__device__ __forceinline__ float log_f(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return __builtin_nanf("");
    if (x == 0.0f) return -__builtin_inff();
    if (x == __builtin_inff()) return x;
    return (float)log_core((double)x);
}

__device__ __forceinline__ float exp_f(float x)
{
    if (x != x) return x;
    if (x > 89.0f) return __builtin_inff();
    if (x < -104.0f) return 0.0f;
    return (float)exp_core((double)x);
}

__device__ __forceinline__ float cos_f(float x)
{
    if (x != x || x == __builtin_inff() || x == -__builtin_inff()) return __builtin_nanf("");
    const double xd = (double)x;
    const double kf = floor(xd * kTwoOverPi + 0.5);
    const double r = (xd - kf * kPio2Hi) - kf * kPio2Lo;
    const int n = (int)((long long)kf & 3);
    const double z = r * r;
    double v;
    if (n == 0 || n == 2) {
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

__device__ __forceinline__ float pow_f(float x, float y)
{
    if (x != x || y != y) return __builtin_nanf("");
    if (y == 0.0f) return 1.0f;
    if (x < 0.0f) return __builtin_nanf("");
    if (x == 0.0f) return y > 0.0f ? 0.0f : __builtin_inff();
    if (x == __builtin_inff()) return y > 0.0f ? __builtin_inff() : 0.0f;
    if (x == 1.0f) return 1.0f;
    const double p = (double)y * log_core((double)x);
    if (p > 89.0) return __builtin_inff();
    if (p < -104.0) return 0.0f;
    return (float)exp_core(p);
}

// pow_f(x, 1000.0f) for the sun term (GPURaytracer.cu:15), same bits: for 0 <= x <= 0.9f the definition above returns 0
// (1000 * ln(0.9f) = -105.4 < -104, and log_core errs by 1e-16), so a ray that points more than 25.8 degrees away from the
// sun skips both binary64 series; a wave of camera rays that all miss away from the sun never runs them.
__device__ __forceinline__ float pow1000_f(float x)
{
    if (x <= 0.9f) return 0.0f;   // x is fmax_f(0, .): never negative, never NaN
    return pow_f(x, 1000.0f);
}

// pow_f(x, 5.0f) for Schlick's term (:643), same bits, without the two binary64 series: x^5 by three binary64
// products is within 2 ulp64 of the exact power, as the definition's exp(5 ln x) is within a few, so the two round
// to the same binary32 unless the value sits within ~2^-47 (relative) of a binary32 rounding boundary; that case
// (about one input in 10^6) and results near the binary32 subnormals, where the boundary sits elsewhere, take the
// definition itself.  Checked against pow_f(x, 5.0f) for every binary32 in [0, 1] on the GPU (rt_detmath_eval op 9).
__device__ __forceinline__ float pow5_f(float x)
{
    const double d = (double)x, d2 = d * d, p = (d2 * d2) * d;
    const uint32_t frac = (uint32_t)f64_bits(p) & 0x1FFFFFFFu;   // the 29 bits below a binary32 mantissa
    const uint32_t dist = frac > 0x10000000u ? frac - 0x10000000u : 0x10000000u - frac;
    if (!(x >= 0.0f && x <= 1.0f) || dist < 4096u || p < 0x1p-120) return pow_f(x, 5.0f);
    return (float)p;
}

// b / 255.0f, correctly rounded, for a byte b -- the texel conversion (Material.cu:199-206 divides on the host;
// here the texels stay RGBA8 and every tap converts).  q = b * fl(1/255) is within an ulp; one residual step
// r = fma(-255, q, b), q + r * fl(1/255) lands on the correctly rounded quotient for all 256 bytes (checked
// exhaustively: tests/test_parity_gpu.py::test_detmath_matches_oracle_bit_for_bit and, in exact rational
// arithmetic, tests/test_oracle.py).  3 instructions instead of the ~10 of an IEEE division, 16 per texture tap set.
__device__ __forceinline__ float byte_unorm(uint32_t b)
{
    const float p = (float)b, c = 0x1.010102p-8f;
    const float q = p * c;
    const float r = __builtin_fmaf(-255.0f, q, p);
    return __builtin_fmaf(r, c, q);
}

// (float)(v / 4294967295.0), the RNG's last step (:824), without the binary64 division: the product with
// fl64(1 / 4294967295) is within 2^-52 of the quotient, and no quotient v / (2^32 - 1) lies that close to a
// binary32 rounding boundary (v * 2^-32 * (1 + 2^-32 + ...): past the 24 kept bits, an exact-looking half is
// always followed by the 2^-32 term).  Checked against the division for all 2^32 values of v on the GPU
// (tests/test_parity_gpu.py::test_detmath_matches_oracle_bit_for_bit).
__device__ __forceinline__ float u32_unorm(uint32_t v)
{
    return (float)((double)v * (1.0 / 4294967295.0));
}

// CUDA max()/min() on floats drop a NaN operand (fmaxf/fminf); pinned select form
__device__ __forceinline__ float fmax_f(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    return a > b ? a : b;
}
__device__ __forceinline__ float fmin_f(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    return a < b ? a : b;
}

} // namespace rtd
