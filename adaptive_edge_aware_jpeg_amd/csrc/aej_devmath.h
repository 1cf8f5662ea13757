// aej_devmath.h -- device math shared by the colour kernels of the encode (color.hip) and decode (decode.hip) paths.
// Every function is a fixed sequence of IEEE-754 operations (no contraction), restating the numerics contract of DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>

namespace aej {

// ------------------------------------------------------------------------------------------------
// deterministic pow: exp2(y * log2(x)) from table look-ups and fma chains only (no division), the same recipe and the same
// generated tables (tools/gen_pow_tables.py) as the CPU side of the contract in DESIGN.md:
//   log2(x): x = 2^e m, i = top 6 mantissa bits, r = fma(m, INVC[i], -1), |r| <= 2^-7; (e + LOGC[i]) + r P(r)
//   exp2(t): k = rint(64 t), r = t - k / 64 (exact), j = k & 63, n = k >> 6; 2^n (T[j] + T[j] (r Q(r)))
// ------------------------------------------------------------------------------------------------
}  // namespace aej
#include "pow_tables.h"
namespace aej {

// where the three 64-entry tables are read from: the __device__ arrays (default) or a workgroup's LDS copy (k_color_planes
// evaluates 3 to 9 pows per pixel; LDS look-ups are cheaper than per-lane global loads)
struct PowTabs { const double *invc, *logc, *exp2t; };
__device__ __forceinline__ PowTabs pow_tabs_global() { return PowTabs{ POW_INVC, POW_LOGC, POW_EXP2T }; }

__device__ __forceinline__ double dev_log2(double x, const PowTabs &pt)
{
    long long b = __double_as_longlong(x);
    int e = (int)(b >> 52) - 1023;
    int i = (int)((b >> 46) & 63ll);
    double m = __longlong_as_double((b & 0x000FFFFFFFFFFFFFll) | 0x3FF0000000000000ll);
    double r = fma(m, pt.invc[i], -1.0);
    double p = POW_L[7];
    p = fma(p, r, POW_L[6]);
    p = fma(p, r, POW_L[5]);
    p = fma(p, r, POW_L[4]);
    p = fma(p, r, POW_L[3]);
    p = fma(p, r, POW_L[2]);
    p = fma(p, r, POW_L[1]);
    p = fma(p, r, POW_L[0]);
    double lo = r * p;
    double hi = (double)e + pt.logc[i];
    return hi + lo;
}

__device__ __forceinline__ double dev_exp2(double t, const PowTabs &pt)
{
    double kd = rint(t * 64.0);
    double r = fma(kd, -0.015625, t);
    long long k = (long long)kd;
    int j = (int)(k & 63ll);
    long long ni = k >> 6;
    double q = POW_E[5];
    q = fma(q, r, POW_E[4]);
    q = fma(q, r, POW_E[3]);
    q = fma(q, r, POW_E[2]);
    q = fma(q, r, POW_E[1]);
    q = fma(q, r, POW_E[0]);
    double s = r * q;
    const double tj = pt.exp2t[j];
    double v = fma(tj, s, tj);
    if (ni < -1022) return 0.0;
    if (ni > 1023) return __longlong_as_double(0x7FF0000000000000ll);
    return v * __longlong_as_double((ni + 1023) << 52);
}

__device__ __forceinline__ double dev_pow(double x, double y, const PowTabs &pt)
{
    if (x == 0.0) return 0.0;
    if (!(x > 0.0)) return __longlong_as_double(0x7FF8000000000000ll);
    if (x < 2.2250738585072014e-308) return 0.0;
    return dev_exp2(y * dev_log2(x, pt), pt);
}
__device__ __forceinline__ double dev_pow(double x, double y) { return dev_pow(x, y, pow_tabs_global()); }

// oklab.py:73 -- np.power(lms float32, 1 / 3) = x^t with t = (double)(float)(1 / 3) = 1/3 + 9.93e-9, as cbrt(x) * x^(t - 1/3): a float32
// Newton iteration for x^(-1/3) from a bit-pattern seed (three steps: 2e-7), one float64 step (4e-14), and the excess exponent as
// 1 + (t - 1/3) ln 2 * (bit-pattern estimate of log2 x).  |rel err| <= 3e-10; sixteen float32 + eleven float64 operations, no table, no
// division.  The same sequence, operation for operation, as pow_third_f32 in the CPU restatement (round 4: the general dev_pow above
// made the OKLAB colour stage of an 8K batch arithmetic-bound at twice its memory time).
__device__ __forceinline__ float dev_pow_third_f32(float x, const PowTabs &pt)
{
    const double third = (double)(float)(1.0 / 3.0);
    if (x == 0.0f) return 0.0f;
    if (!(x > 0.0f)) return __uint_as_float(0x7FC00000u);
    if (x < 1e-30f || x > 1e30f) return (float)dev_pow((double)x, third, pt);      // never produced by 8-bit images; keeps the seed in range
    const unsigned ix = __float_as_uint(x);
    float r = __uint_as_float(0x54a2fa8cu - ix / 3u);
    const float x3 = x * 0x1.555556p-2f;
#pragma unroll
    for (int it = 0; it < 3; it++) {
        const float r2 = r * r;
        const float r3 = r2 * r;
        const float t = __builtin_fmaf(-x3, r3, 0x1.555556p+0f);
        r = r * t;
    }
    const double xd = (double)x;
    double rd = (double)r;
    const double r2d = rd * rd;
    const double r3d = r2d * rd;
    const double td = fma(-(xd * 0x1.5555555555555p-2), r3d, 0x1.5555555555555p+0);
    rd = rd * td;
    double c = xd * rd;
    c = c * rd;                                                   // cbrt(x)
    const double L = fma((double)(int)ix, 0x1p-23, -0x1.fbd3f7ced9168p+6);      // ~ log2(x): bit pattern / 2^23 - 126.957
    const double u = 0x1.d9303ff8f9009p-28 * L;                  // (t - 1/3) ln 2 * log2(x)
    return (float)fma(c, u, c);
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
__device__ __forceinline__ float srgb_to_linear(float v, const PowTabs &pt)   // common.py:34-60 (float64 under numba typing)
{
    double d = (double)v;
    if (d <= 0.04045) return (float)(d / 12.92);
    return (float)dev_pow((d + 0.055) / 1.055, 2.4, pt);
}
__device__ __forceinline__ float srgb_to_linear(float v) { return srgb_to_linear(v, pow_tabs_global()); }
__device__ __forceinline__ double pq_inverse_eotf(double v, double m2, const PowTabs &pt)   // common.py:131-159
{
    const double c1 = 3424.0 / 4096.0, c2 = 2413.0 / 128.0, c3 = 2392.0 / 128.0, m1 = 2610.0 / 16384.0;
    double tmp = dev_pow(v / 10000.0, m1, pt);
    double num = c1 + c2 * tmp;
    double den = 1.0 + c3 * tmp;
    return dev_pow(num / den, m2, pt);
}

}  // namespace aej
