// aej_devmath.h -- device math shared by the colour kernels of the encode (color.hip) and decode (decode.hip) paths.
// Every function is a fixed sequence of IEEE-754 operations (no contraction), restating the numerics contract of DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>

namespace aej {

// ------------------------------------------------------------------------------------------------
// deterministic pow (same recipe as the contract in DESIGN.md: atanh-series log2, Taylor exp2)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double dev_log2(double x)
{
    long long b = __double_as_longlong(x);
    int e = (int)(b >> 52) - 1023;
    double m = __longlong_as_double((b & 0x000FFFFFFFFFFFFFll) | 0x3FF0000000000000ll);
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    double z = (m - 1.0) / (m + 1.0);
    double z2 = z * z;
    double p = 2.0 / 25.0;
    p = fma(p, z2, 2.0 / 23.0);
    p = fma(p, z2, 2.0 / 21.0);
    p = fma(p, z2, 2.0 / 19.0);
    p = fma(p, z2, 2.0 / 17.0);
    p = fma(p, z2, 2.0 / 15.0);
    p = fma(p, z2, 2.0 / 13.0);
    p = fma(p, z2, 2.0 / 11.0);
    p = fma(p, z2, 2.0 / 9.0);
    p = fma(p, z2, 2.0 / 7.0);
    p = fma(p, z2, 2.0 / 5.0);
    p = fma(p, z2, 2.0 / 3.0);
    p = fma(p, z2, 2.0);
    double lnm = z * p;
    return fma(lnm, 1.4426950408889634, (double)e);
}

__device__ __forceinline__ double dev_exp2(double t)
{
    double n = rint(t);
    double r = (t - n) * 0.6931471805599453;
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    long long ni = (long long)n;
    if (ni < -1022) return 0.0;
    if (ni > 1023) return __longlong_as_double(0x7FF0000000000000ll);
    return p * __longlong_as_double((ni + 1023) << 52);
}

__device__ __forceinline__ double dev_pow(double x, double y)
{
    if (x == 0.0) return 0.0;
    if (!(x > 0.0)) return __longlong_as_double(0x7FF8000000000000ll);
    if (x < 2.2250738585072014e-308) return 0.0;
    return dev_exp2(y * dev_log2(x));
}

// ------------------------------------------------------------------------------------------------
// matrices: float32(value) of the Python literals (numpy: np.array([...], dtype=np.float32))
// ------------------------------------------------------------------------------------------------
#define F(x) ((float)(x))
__device__ __forceinline__ float dot3(float m0, float m1, float m2, float a, float b, float c)
{
    float acc = a * m0;               // np.dot float32: k-ordered fma chain (OpenBLAS sgemm)
    acc = __builtin_fmaf(b, m1, acc);
    return __builtin_fmaf(c, m2, acc);
}
__device__ __forceinline__ float lin3(float m0, float m1, float m2, float X, float Y, float Z)
{
    float t = m0 * X;
    float u = m1 * Y;
    t = t + u;
    u = m2 * Z;
    return t + u;
}
__device__ __forceinline__ double lin3d(float m0, float m1, float m2, double a, double b, double c)
{
    double t = (double)m0 * a;
    double u = (double)m1 * b;
    t = t + u;
    u = (double)m2 * c;
    return t + u;
}
__device__ __forceinline__ float srgb_to_linear(float v)   // common.py:34-60 (float64 under numba typing)
{
    double d = (double)v;
    if (d <= 0.04045) return (float)(d / 12.92);
    return (float)dev_pow((d + 0.055) / 1.055, 2.4);
}
__device__ __forceinline__ double pq_inverse_eotf(double v, double m2)   // common.py:131-159
{
    const double c1 = 3424.0 / 4096.0, c2 = 2413.0 / 128.0, c3 = 2392.0 / 128.0, m1 = 2610.0 / 16384.0;
    double tmp = dev_pow(v / 10000.0, m1);
    double num = c1 + c2 * tmp;
    double den = 1.0 + c3 * tmp;
    return dev_pow(num / den, m2);
}

}  // namespace aej
