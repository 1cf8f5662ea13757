/*
 * aej_oracle.c -- CPU ORACLE for the adaptive-JPEG encode hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is a scalar, order-defined restatement of the reference's algorithm
 * (fevzibabaoglu/adaptive-edge-aware-jpeg, Python) for the path
 *     colour convert -> chroma downsample -> Canny edge map -> quadtree -> DCT -> quantise -> zigzag
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (adaptive_edge_aware_jpeg_amd/) never does.
 *
 * PARITY STATUS (see DESIGN.md "Oracle"):
 *   - pinned by reference-generated golden vectors (tests/golden/, made by executing the reference's
 *     own Python): quadtree leaves/states/root, zigzag tables, quality law, layer shapes, colour
 *     forward transforms for all 7 spaces (bit-exact for the 3 matrix spaces, <= 1 float32 ulp for the
 *     transcendental ones), normalisation.
 *   - pinned against the real third-party library present in the container: np.dot float32 (bit
 *     exact == k-ordered fmaf chain), np.percentile, np.pad(reflect), np.round(f32/int32).
 *   - PARITY UNPINNED for everything the reference computes inside OpenCV (cv2 is not installed and
 *     the reference holds no golden vectors for it): INTER_AREA resize, CLAHE, GaussianBlur,
 *     bilateralFilter, Canny, dct.  Those stages restate OpenCV 4.x's published algorithm
 *     (opencv-python==4.11.0.86, requirements.txt:18) as cited per function below.
 *
 * Citations are path:line under /root/reference/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(__GNUC__)
#define ORC_API __attribute__((visibility("default")))
#else
#define ORC_API
#endif

/* ------------------------------------------------------------------------------------------
 * Deterministic pow(): only IEEE-754 +,-,*,fma,rint on doubles, integer bit operations and table
 * look-ups, in a fixed order, so that the HIP kernels (which restate the same recipe with the same
 * generated tables, tools/gen_pow_tables.py) produce the same bits.  |rel err| ~ 2e-16 for the
 * arguments of the colour code.
 *   log2(x): x = 2^e m, i = top 6 mantissa bits, r = fma(m, INVC[i], -1), |r| <= 2^-7;
 *            log2(x) = (e + LOGC[i]) + r P(r), P = degree-7 Taylor polynomial of log2(1 + r) / r
 *   exp2(t): k = rint(64 t), r = t - k / 64 (exact), j = k & 63, n = k >> 6;
 *            exp2(t) = 2^n (T[j] + T[j] (r Q(r))), Q = degree-5 Taylor polynomial of (2^r - 1) / r
 * ---------------------------------------------------------------------------------------- */
#include "aej_pow_tables.h"

static inline uint64_t d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

static double orc_log2(double x) /* x > 0, normal */
{
    uint64_t b = d2u(x);
    int e = (int)(b >> 52) - 1023;
    int i = (int)((b >> 46) & 63u);
    double m = u2d((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull); /* [1,2) */
    double r = fma(m, POW_INVC[i], -1.0);
    double p = POW_L[7];
    p = fma(p, r, POW_L[6]);
    p = fma(p, r, POW_L[5]);
    p = fma(p, r, POW_L[4]);
    p = fma(p, r, POW_L[3]);
    p = fma(p, r, POW_L[2]);
    p = fma(p, r, POW_L[1]);
    p = fma(p, r, POW_L[0]);
    double lo = r * p;
    double hi = (double)e + POW_LOGC[i];
    return hi + lo;
}

static double orc_exp2(double t)
{
    double kd = rint(t * 64.0);
    double r = fma(kd, -0.015625, t);
    int64_t k = (int64_t)kd;
    int j = (int)(k & 63);
    int64_t ni = k >> 6;                     /* arithmetic shift: floor(k / 64) */
    double q = POW_E[5];
    q = fma(q, r, POW_E[4]);
    q = fma(q, r, POW_E[3]);
    q = fma(q, r, POW_E[2]);
    q = fma(q, r, POW_E[1]);
    q = fma(q, r, POW_E[0]);
    double s = r * q;
    double v = fma(POW_EXP2T[j], s, POW_EXP2T[j]);
    if (ni < -1022) return 0.0;
    if (ni > 1023) return INFINITY;
    return v * u2d((uint64_t)(ni + 1023) << 52);
}

/* pow for the arguments the colour code feeds it: x >= 0 (0 -> 0), x < 0 or NaN -> NaN. */
static double orc_pow(double x, double y)
{
    if (x == 0.0) return 0.0;
    if (!(x > 0.0)) return NAN;
    if (x < 2.2250738585072014e-308) return 0.0; /* subnormal input: never produced by the path */
    return orc_exp2(y * orc_log2(x));
}

ORC_API void orc_pow_array(const double *x, double y, double *out, int64_t n)
{
    for (int64_t i = 0; i < n; i++) out[i] = orc_pow(x[i], y);
}

/* ------------------------------------------------------------------------------------------
 * a-1  colour:  color.convert("sRGB", space, x)  (src/color/conversion.py:95-124)
 * ---------------------------------------------------------------------------------------- */
enum { SP_YCBCR = 0, SP_YCOCG = 1, SP_YCOCG_R = 2, SP_OKLAB = 3, SP_ICTCP = 4, SP_ICACB = 5, SP_JZAZBZ = 6, SP_XYZ = 7 };

/* numpy builds these from Python doubles cast to float32 (np.array(..., dtype=np.float32)) */
#define F(x) ((float)(x))
static const float M_YCBCR[9] = { F(0.299000), F(0.587000), F(0.114000), F(-0.168736), F(-0.331264), F(0.500000),
                                  F(0.500000), F(-0.418688), F(-0.081312) };            /* ycbcr.py:25-30 */
static const float M_YCOCG[9] = { F(0.25), F(0.50), F(0.25), F(0.50), F(0.00), F(-0.50), F(-0.25), F(0.50), F(-0.25) }; /* ycocg.py:25-30 */
static const float M_YCOCG_R[9] = { F(0.25), F(0.50), F(0.25), F(1.00), F(0.00), F(-1.00), F(-0.50), F(1.00), F(-0.50) }; /* ycocg.py:46-51 */
static const float M_RGB_XYZ[9] = { F(0.4124564), F(0.3575761), F(0.1804375), F(0.2126729), F(0.7151522), F(0.0721750),
                                    F(0.0193339), F(0.1191920), F(0.9503041) };         /* xyz.py:27-32 */
static const float M_OK_LMS[9] = { F(0.8189330101), F(0.3618667424), F(-0.1288597137), F(0.0329845436), F(0.9293118715),
                                   F(0.0361456387), F(0.0482003018), F(0.2643662691), F(0.6338517070) }; /* oklab.py:27-32 */
static const float M_OK_LAB[9] = { F(0.2104542553), F(0.7936177850), F(-0.0040720468), F(1.9779984951), F(-2.4285922050),
                                   F(0.4505937099), F(0.0259040371), F(0.7827717662), F(-0.8086757660) }; /* oklab.py:39-44 */
static const float M_ICT_LMS[9] = { F(0.3592), F(0.6976), F(-0.0358), F(-0.1922), F(1.1004), F(0.0755), F(0.0070), F(0.0749), F(0.8434) }; /* ictcp.py:142-147 */
static const float M_ICT_OUT[9] = { F(0.5000), F(0.5000), F(0.0000), F(1.6137), F(-3.3234), F(1.7097), F(4.3781), F(-4.2455), F(-0.1325) }; /* ictcp.py:152-157 */
static const float M_ICA_RGB[9] = { F(0.37613), F(0.70431), F(-0.05675), F(-0.21649), F(1.14744), F(0.05356), F(0.02567), F(0.16713), F(0.74235) }; /* icacb.py:142-147 */
static const float M_ICA_OUT[9] = { F(0.4949), F(0.5037), F(0.0015), F(4.2854), F(-4.5462), F(0.2609), F(0.3605), F(1.1499), F(-1.5105) }; /* icacb.py:152-157 */
static const float M_JZ_LMS[9] = { F(0.41478972), F(0.579999), F(0.0146480), F(-0.2015100), F(1.120649), F(0.0531008),
                                   F(-0.0166008), F(0.264800), F(0.6684799) };          /* jzazbz.py:191-196 */
static const float M_JZ_OUT[9] = { F(0.500000), F(0.500000), F(0.000000), F(3.524000), F(-4.066708), F(0.542708),
                                   F(0.199076), F(1.096799), F(-1.295875) };            /* jzazbz.py:201-206 */

/* np.dot((N,3) f32, (3,3) f32): OpenBLAS sgemm == k-ordered fmaf chain (verified bit-exact against
 * numpy in this container, tests/test_oracle_pins.py). */
static inline float dot3(const float *m, float a, float b, float c)
{
    float acc = a * m[0];
    acc = fmaf(b, m[1], acc);
    return fmaf(c, m[2], acc);
}

/* common.py:34-60 -- under numba typing the float32 pixel is promoted to float64 */
static inline float srgb_to_linear(float v)
{
    double d = (double)v;
    if (d <= 0.04045) return (float)(d / 12.92);
    return (float)orc_pow((d + 0.055) / 1.055, 2.4);
}

/* oklab.py:73 -- np.power(lms float32, 1 / 3): the Python float is cast to float32, so the exponent is t = (double)(float)(1 / 3) = 1/3 + 9.93e-9.
 * x^t = cbrt(x) * x^(t - 1/3): the cube root from a float32 Newton iteration for x^(-1/3) (bit-pattern seed, 3.9 % off; three steps
 * r <- r (4/3 - (x/3) r^3) reach 2e-7) finished by one float64 step (4e-14), and the excess exponent as 1 + (t - 1/3) ln 2 * L with L the
 * bit-pattern estimate of log2 x (+-0.05: 3e-10 relative in the result; for x = 1e-5 the factor is 1 - 1.1e-7, two float32 ulps, so it is
 * not optional).  Total |rel err| <= 3e-10 against x^t: the float32 result differs from the correctly rounded one in 0.2 % of the inputs,
 * by one ulp (the reference's own vectorised powf differs from it in 20 %).  Sixteen float32 + eleven float64 operations, no table, no
 * division -- the general pow of this file (two table look-ups, a degree-7 and a degree-5 polynomial in float64) cost the OKLAB colour
 * stage of an 8K image as much again as its memory traffic.  The same sequence is in csrc/aej_devmath.h. */
static inline float pow_third_f32(float x)
{
    const double third = (double)(float)(1.0 / 3.0);
    if (x == 0.0f) return 0.0f;
    if (!(x > 0.0f)) return NAN;
    if (x < 1e-30f || x > 1e30f) return (float)orc_pow((double)x, third);      /* never produced by 8-bit images; keeps the seed in range */
    uint32_t ix;
    memcpy(&ix, &x, 4);
    const uint32_t ir = 0x54a2fa8cu - ix / 3u;
    float r;
    memcpy(&r, &ir, 4);
    const float x3 = x * 0x1.555556p-2f;
    for (int it = 0; it < 3; it++) {
        const float r2 = r * r;
        const float r3 = r2 * r;
        const float t = fmaf(-x3, r3, 0x1.555556p+0f);
        r = r * t;
    }
    const double xd = (double)x;
    double rd = (double)r;
    const double r2d = rd * rd;
    const double r3d = r2d * rd;
    const double td = fma(-(xd * 0x1.5555555555555p-2), r3d, 0x1.5555555555555p+0);
    rd = rd * td;
    double c = xd * rd;
    c = c * rd;                                                   /* cbrt(x) */
    const double L = fma((double)(int32_t)ix, 0x1p-23, -0x1.fbd3f7ced9168p+6);      /* ~ log2(x): bit pattern / 2^23 - 126.957 */
    const double u = 0x1.d9303ff8f9009p-28 * L;                  /* (t - 1/3) ln 2 * log2(x) */
    return (float)fma(c, u, c);
}

static int g_oklab_independent_pow = 0;
ORC_API void orc_set_oklab_independent_pow(int on) { g_oklab_independent_pow = on; }
/* x^(float)(1/3) both ways, for the test that pins the short sequence to the general pow */
ORC_API void orc_pow_third_array(const float *x, float *out, int64_t n, int independent)
{
    const double third = (double)(float)(1.0 / 3.0);
    for (int64_t i = 0; i < n; i++) out[i] = independent ? (x[i] == 0.0f ? 0.0f : (float)orc_pow((double)x[i], third)) : pow_third_f32(x[i]);
}

/* common.py:131-159 */
static inline double pq_inverse_eotf(double v, double m2)
{
    const double c1 = 3424.0 / 4096.0, c2 = 2413.0 / 128.0, c3 = 2392.0 / 128.0, m1 = 2610.0 / 16384.0;
    double tmp = orc_pow(v / 10000.0, m1);
    double num = c1 + c2 * tmp;
    double den = 1.0 + c3 * tmp;
    return orc_pow(num / den, m2);
}

/* float32 a*X + b*Y + c*Z as the Python expression evaluates it (ictcp.py:49-57): no fma */
static inline float lin3(const float *m, float X, float Y, float Z)
{
    float t = m[0] * X;
    float u = m[1] * Y;
    t = t + u;
    u = m[2] * Z;
    return t + u;
}
static inline double lin3d(const float *m, double a, double b, double c)
{
    double t = (double)m[0] * a;
    double u = (double)m[1] * b;
    t = t + u;
    u = (double)m[2] * c;
    return t + u;
}

static void color_px(int space, float r, float g, float b, float *o)
{
    switch (space) {
    case SP_YCBCR: /* ycbcr.py:61 */
        o[0] = dot3(M_YCBCR + 0, r, g, b); o[1] = dot3(M_YCBCR + 3, r, g, b); o[2] = dot3(M_YCBCR + 6, r, g, b); return;
    case SP_YCOCG: /* ycocg.py:82 */
        o[0] = dot3(M_YCOCG + 0, r, g, b); o[1] = dot3(M_YCOCG + 3, r, g, b); o[2] = dot3(M_YCOCG + 6, r, g, b); return;
    case SP_YCOCG_R: /* ycocg.py:121 */
        o[0] = dot3(M_YCOCG_R + 0, r, g, b); o[1] = dot3(M_YCOCG_R + 3, r, g, b); o[2] = dot3(M_YCOCG_R + 6, r, g, b); return;
    default: break;
    }
    /* xyz.py:63-64 */
    float lr = srgb_to_linear(r), lg = srgb_to_linear(g), lb = srgb_to_linear(b);
    float X = dot3(M_RGB_XYZ + 0, lr, lg, lb), Y = dot3(M_RGB_XYZ + 3, lr, lg, lb), Z = dot3(M_RGB_XYZ + 6, lr, lg, lb);
    if (space == SP_XYZ) { o[0] = X; o[1] = Y; o[2] = Z; return; } /* XYZ.srgb_to_xyz, xyz.py:63-64 */
    if (space == SP_OKLAB) { /* oklab.py:71-75 */
        float l = dot3(M_OK_LMS + 0, X, Y, Z), m = dot3(M_OK_LMS + 3, X, Y, Z), s = dot3(M_OK_LMS + 6, X, Y, Z);
        /* np.power(f32, python float) -> powf(x, (float)(1/3)).  g_oklab_independent_pow (tests only): the general float64 pow of this
         * file instead of the short sequence the HIP kernel shares -- a path with no code in common with the kernel */
        float lp, mp, sp;
        if (g_oklab_independent_pow) {
            const double third = (double)(float)(1.0 / 3.0);
            lp = (float)orc_pow((double)l, third); mp = (float)orc_pow((double)m, third); sp = (float)orc_pow((double)s, third);
        } else { lp = pow_third_f32(l); mp = pow_third_f32(m); sp = pow_third_f32(s); }
        o[0] = dot3(M_OK_LAB + 0, lp, mp, sp); o[1] = dot3(M_OK_LAB + 3, lp, mp, sp); o[2] = dot3(M_OK_LAB + 6, lp, mp, sp);
        return;
    }
    if (space == SP_ICTCP || space == SP_ICACB) { /* ictcp.py:45-81, icacb.py:45-81 */
        const float *m1 = space == SP_ICTCP ? M_ICT_LMS : M_ICA_RGB;
        const float *m2 = space == SP_ICTCP ? M_ICT_OUT : M_ICA_OUT;
        float L = lin3(m1 + 0, X, Y, Z), M = lin3(m1 + 3, X, Y, Z), S = lin3(m1 + 6, X, Y, Z);
        const double pm2 = 2523.0 / 32.0;
        double Lp = pq_inverse_eotf((double)L, pm2), Mp = pq_inverse_eotf((double)M, pm2), Sp = pq_inverse_eotf((double)S, pm2);
        o[0] = (float)lin3d(m2 + 0, Lp, Mp, Sp); o[1] = (float)lin3d(m2 + 3, Lp, Mp, Sp); o[2] = (float)lin3d(m2 + 6, Lp, Mp, Sp);
        return;
    }
    /* JzAzBz, jzazbz.py:54-99, constants :178-189 */
    {
        const double bb = 1.15, gg = 0.66, d = -0.56, d0 = 1.6295499532821566e-11, p = 1.7 * 2523.0 / 32.0;
        double Xp = bb * (double)X - (bb - 1.0) * (double)Z;
        double Yp = gg * (double)Y - (gg - 1.0) * (double)X;
        double Zp = (double)Z;
        /* numba typing: M[i,2] * Z_p is float32*float32 -> float32, the other two products are float64 */
        (void)Zp;
        double L = ((double)M_JZ_LMS[0] * Xp + (double)M_JZ_LMS[1] * Yp) + (double)(M_JZ_LMS[2] * Z);
        double M = ((double)M_JZ_LMS[3] * Xp + (double)M_JZ_LMS[4] * Yp) + (double)(M_JZ_LMS[5] * Z);
        double S = ((double)M_JZ_LMS[6] * Xp + (double)M_JZ_LMS[7] * Yp) + (double)(M_JZ_LMS[8] * Z);
        double Lp = pq_inverse_eotf(L, p), Mp = pq_inverse_eotf(M, p), Sp = pq_inverse_eotf(S, p);
        double Iz = lin3d(M_JZ_OUT + 0, Lp, Mp, Sp), Az = lin3d(M_JZ_OUT + 3, Lp, Mp, Sp), Bz = lin3d(M_JZ_OUT + 6, Lp, Mp, Sp);
        double Jz = ((1.0 + d) * Iz) / (1.0 + d * Iz) - d0;
        o[0] = (float)Jz; o[1] = (float)Az; o[2] = (float)Bz;
    }
}

ORC_API void orc_color_forward(int space, const float *rgb, float *out, int64_t n)
{
    for (int64_t i = 0; i < n; i++) color_px(space, rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], out + 3 * i);
}

/* ------------------------------------------------------------------------------------------
 * a-2  Jpeg._downsample (jpeg.py:323-338): cv.resize(layer, (W/rw, H/rh), INTER_AREA), integer ratios.
 *      OpenCV 4.x resize.cpp: same size -> copy; 2x2 -> ((r0e+r0o)+(r1e+r1o))*0.25f (ResizeAreaFastVec_SIMD_32f);
 *      other integer areas -> sequential sum over (row-major) taps, times 1.f/area (ResizeAreaFast_Invoker).
 *      conv is the (H,W,3) interleaved colour-converted image, ch selects the layer (jpeg.py:263-264 transposes).
 * ---------------------------------------------------------------------------------------- */
/* general INTER_AREA (non-integer scale): OpenCV 4.x resize.cpp computeResizeAreaTab + ResizeArea_Invoker.
 * tab entries for destination index d are contiguous; off[d]..off[d+1]. */
static int area_tab(int ssize, int dsize, double scale, int *off, int *si, float *alpha)
{
    int k = 0;
    for (int dx = 0; dx < dsize; dx++) {
        double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        double cellWidth = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        if (sx2 > ssize - 1) sx2 = ssize - 1;
        if (sx1 > sx2) sx1 = sx2;
        off[dx] = k;
        if (sx1 - fsx1 > 1e-3) { si[k] = sx1 - 1; alpha[k++] = (float)((sx1 - fsx1) / cellWidth); }
        for (int sx = sx1; sx < sx2; sx++) { si[k] = sx; alpha[k++] = (float)(1.0 / cellWidth); }
        if (fsx2 - sx2 > 1e-3) {
            double a = fsx2 - sx2; if (a > 1.) a = 1.; if (a > cellWidth) a = cellWidth;
            si[k] = sx2; alpha[k++] = (float)(a / cellWidth);
        }
    }
    off[dsize] = k;
    return k;
}

ORC_API int orc_downsample(const float *conv, int H, int W, int ch, int rh, int rw, float *out)
{
    if (rh < 1 || rw < 1) return -1;
    int Ho = H / rh, Wo = W / rw;
    if (Ho < 1 || Wo < 1) return -1;
    if (Ho == H && Wo == W) {                         /* dsize == ssize: copy */
        for (int64_t i = 0; i < (int64_t)H * W; i++) out[i] = conv[i * 3 + ch];
        return 0;
    }
    /* resize(): scale = 1 / (dsize / ssize) in double; "area fast" iff both scales are integers */
    double scale_x = 1.0 / ((double)Wo / (double)W), scale_y = 1.0 / ((double)Ho / (double)H);
    int isx = (int)lrint(scale_x), isy = (int)lrint(scale_y);
    int fast = fabs(scale_x - isx) < 2.220446049250313e-16 && fabs(scale_y - isy) < 2.220446049250313e-16;
    if (fast) {
        for (int y = 0; y < Ho; y++)
            for (int x = 0; x < Wo; x++) {
                float v;
                if (isy == 2 && isx == 2) {
                    const float *r0 = conv + ((int64_t)(2 * y) * W + 2 * x) * 3 + ch;
                    const float *r1 = r0 + (int64_t)W * 3;
                    v = ((r0[0] + r0[3]) + (r1[0] + r1[3])) * 0.25f;
                } else {
                    float sum = 0.f;
                    for (int dy = 0; dy < isy; dy++)
                        for (int dx = 0; dx < isx; dx++)
                            sum += conv[((int64_t)(y * isy + dy) * W + (x * isx + dx)) * 3 + ch];
                    v = sum * (1.f / (float)(isx * isy));
                }
                out[(int64_t)y * Wo + x] = v;
            }
        return 0;
    }
    int *xoff = (int *)malloc(sizeof(int) * (size_t)(Wo + 1)), *yoff = (int *)malloc(sizeof(int) * (size_t)(Ho + 1));
    int *xsi = (int *)malloc(sizeof(int) * (size_t)(W + 2 * Wo + 2)), *ysi = (int *)malloc(sizeof(int) * (size_t)(H + 2 * Ho + 2));
    float *xal = (float *)malloc(sizeof(float) * (size_t)(W + 2 * Wo + 2)), *yal = (float *)malloc(sizeof(float) * (size_t)(H + 2 * Ho + 2));
    area_tab(W, Wo, scale_x, xoff, xsi, xal);
    area_tab(H, Ho, scale_y, yoff, ysi, yal);
    for (int y = 0; y < Ho; y++)
        for (int x = 0; x < Wo; x++) {
            float sum = 0.f;
            for (int j = yoff[y]; j < yoff[y + 1]; j++) {
                const float *S = conv + (int64_t)ysi[j] * W * 3 + ch;
                float buf = 0.f;
                for (int k = xoff[x]; k < xoff[x + 1]; k++) { float t = S[(int64_t)xsi[k] * 3] * xal[k]; buf = buf + t; }
                float t = yal[j] * buf;
                sum = sum + t;
            }
            out[(int64_t)y * Wo + x] = sum;
        }
    free(xoff); free(yoff); free(xsi); free(ysi); free(xal); free(yal);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * a-3  (img * 255).astype(np.uint8)  (edge_detection.py:70): float32 multiply, truncation toward zero,
 *      x86 wrap mod 256 (cvttss2si then low byte).
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_to_u8(const float *plane, uint8_t *out, int64_t n)
{
    for (int64_t i = 0; i < n; i++) {
        float v = plane[i] * 255.0f;
        int32_t t = (v >= 2147483648.0f || v < -2147483648.0f || v != v) ? (int32_t)0x80000000 : (int32_t)v;
        out[i] = (uint8_t)(t & 0xFF);
    }
}

static inline int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * (n - 1) - i;
    }
    return i;
}

/* ------------------------------------------------------------------------------------------
 * a-4  CLAHE(clipLimit=0.75, tileGridSize=(4,4)).apply(u8)  (edge_detection.py:73-74)
 *      OpenCV 4.x clahe.cpp (CLAHE_CalcLut_Body / CLAHE_Interpolation_Body), 8-bit.
 * ---------------------------------------------------------------------------------------- */
static void clahe_impl(const uint8_t *src, uint8_t *dst, int H, int W, double clip_limit)
{
    const int TX = 4, TY = 4;
    int Hp = H, Wp = W;
    if (W % TX != 0 || H % TY != 0) { Hp = H + (TY - H % TY); Wp = W + (TX - W % TX); }
    int tw = Wp / TX, th = Hp / TY;
    int area = tw * th;
    float lutScale = 255.0f / (float)area;
    int clip = 0;                         /* clahe.cpp: clipLimit_ <= 0 means no clipping */
    if (clip_limit > 0.0) { clip = (int)(clip_limit * (double)area / 256.0); if (clip < 1) clip = 1; }
    uint8_t lut[16][256];
    for (int ty = 0; ty < TY; ty++)
        for (int tx = 0; tx < TX; tx++) {
            int hist[256];
            memset(hist, 0, sizeof hist);
            for (int y = ty * th; y < (ty + 1) * th; y++) {
                int sy = reflect101(y, H);
                for (int x = tx * tw; x < (tx + 1) * tw; x++) hist[src[(int64_t)sy * W + reflect101(x, W)]]++;
            }
            if (clip > 0) {
                int clipped = 0;
                for (int i = 0; i < 256; i++)
                    if (hist[i] > clip) { clipped += hist[i] - clip; hist[i] = clip; }
                int batch = clipped / 256, resid = clipped - batch * 256;
                for (int i = 0; i < 256; i++) hist[i] += batch;
                if (resid != 0) {
                    int step = 256 / resid; if (step < 1) step = 1;
                    for (int i = 0; i < 256 && resid > 0; i += step, resid--) hist[i]++;
                }
            }
            int sum = 0;
            for (int i = 0; i < 256; i++) {
                sum += hist[i];
                float f = (float)sum * lutScale;
                long r = lrintf(f); /* cvRound: round-half-even */
                lut[ty * TX + tx][i] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
            }
        }
    float inv_tw = 1.0f / (float)tw, inv_th = 1.0f / (float)th;
    for (int y = 0; y < H; y++) {
        float tyf = (float)y * inv_th - 0.5f;
        int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
        float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > TY - 1) ty2 = TY - 1;
        for (int x = 0; x < W; x++) {
            float txf = (float)x * inv_tw - 0.5f;
            int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
            float xa = txf - (float)tx1, xa1 = 1.0f - xa;
            if (tx1 < 0) tx1 = 0;
            if (tx2 > TX - 1) tx2 = TX - 1;
            int v = src[(int64_t)y * W + x];
            float a = (float)lut[ty1 * TX + tx1][v] * xa1;
            float b = (float)lut[ty1 * TX + tx2][v] * xa;
            float c = (float)lut[ty2 * TX + tx1][v] * xa1;
            float d = (float)lut[ty2 * TX + tx2][v] * xa;
            float top = a + b, bot = c + d;
            float res = top * ya1 + bot * ya; /* separate mul, mul, add: -ffp-contract=off */
            long r = lrintf(res);
            dst[(int64_t)y * W + x] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
        }
    }
}

ORC_API void orc_clahe(const uint8_t *src, uint8_t *dst, int H, int W) { clahe_impl(src, dst, H, W, 0.75); }

/* ------------------------------------------------------------------------------------------
 * a-5  GaussianBlur(u8, (3,3), 0)  (edge_detection.py:77): kernel [1 2 1]/4, 8-bit fixed point path
 *      => (sum [1 2 1;2 4 2;1 2 1]*p + 8) >> 4, BORDER_REFLECT_101.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_gauss3(const uint8_t *src, uint8_t *dst, int H, int W)
{
    for (int y = 0; y < H; y++) {
        int y0 = reflect101(y - 1, H), y2 = reflect101(y + 1, H);
        for (int x = 0; x < W; x++) {
            int x0 = reflect101(x - 1, W), x2 = reflect101(x + 1, W);
            const uint8_t *r0 = src + (int64_t)y0 * W, *r1 = src + (int64_t)y * W, *r2 = src + (int64_t)y2 * W;
            int s = (r0[x0] + 2 * r0[x] + r0[x2]) + 2 * (r1[x0] + 2 * r1[x] + r1[x2]) + (r2[x0] + 2 * r2[x] + r2[x2]);
            dst[(int64_t)y * W + x] = (uint8_t)((s + 8) >> 4);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * a-6  bilateralFilter(u8, 5, 75, 75)  (edge_detection.py:78).  OpenCV 4.x bilateral_filter:
 *      radius 2, circular mask (13 taps, row-major), weights (float)exp(double), w = sw*cw,
 *      wsum += w, sum = fma(val, w, sum), out = cvRound(sum / wsum), BORDER_REFLECT_101.
 * ---------------------------------------------------------------------------------------- */
static void bilateral_tables_impl(float *space_w /*13*/, int *dy /*13*/, int *dx /*13*/, float *color_w /*256*/, double sigma_color, double sigma_space)
{
    if (sigma_color <= 0) sigma_color = 1;      /* bilateral_filter.dispatch.cpp */
    if (sigma_space <= 0) sigma_space = 1;
    double cc = -0.5 / (sigma_color * sigma_color), sc = -0.5 / (sigma_space * sigma_space);
    for (int i = 0; i < 256; i++) color_w[i] = (float)exp((double)i * (double)i * cc);
    int k = 0;
    for (int i = -2; i <= 2; i++)
        for (int j = -2; j <= 2; j++) {
            double r = sqrt((double)i * i + (double)j * j);
            if (r > 2.0) continue;
            space_w[k] = (float)exp(r * r * sc);
            dy[k] = i; dx[k] = j; k++;
        }
}

ORC_API void orc_bilateral_tables(float *space_w, int *dy, int *dx, float *color_w) { bilateral_tables_impl(space_w, dy, dx, color_w, 75.0, 75.0); }

static void bilateral5_impl(const uint8_t *src, uint8_t *dst, int H, int W, double sigma_color, double sigma_space)
{
    float sw[13], cw[256];
    int dy[13], dx[13];
    bilateral_tables_impl(sw, dy, dx, cw, sigma_color, sigma_space);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int v0 = src[(int64_t)y * W + x];
            float sum = 0.f, wsum = 0.f;
            for (int k = 0; k < 13; k++) {
                int v = src[(int64_t)reflect101(y + dy[k], H) * W + reflect101(x + dx[k], W)];
                int dlt = v - v0; if (dlt < 0) dlt = -dlt;
                float w = sw[k] * cw[dlt];
                wsum += w;
                sum = fmaf((float)v, w, sum);
            }
            long r = lrintf(sum / wsum);
            dst[(int64_t)y * W + x] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
        }
}

ORC_API void orc_bilateral5(const uint8_t *src, uint8_t *dst, int H, int W) { bilateral5_impl(src, dst, H, W, 75.0, 75.0); }

/* ------------------------------------------------------------------------------------------
 * a-7  np.percentile(blur, 10) / (blur, 30)  (edge_detection.py:81-82): method 'linear' on the
 *      sorted values == histogram walk.  numpy _lerp: a + (b-a)*t, or b - (b-a)*(1-t) when t >= 0.5.
 * ---------------------------------------------------------------------------------------- */
static double percentile_from_hist(const int64_t *hist, int64_t n, double q)
{
    double virt = (double)(n - 1) * (q / 100.0);
    int64_t lo = (int64_t)floor(virt);
    double t = virt - (double)lo;
    int64_t hi = lo + 1; if (hi > n - 1) hi = n - 1;
    int a = 0, b = 0; int64_t cum = 0; int fa = 0, fb = 0;
    for (int v = 0; v < 256; v++) {
        cum += hist[v];
        if (!fa && cum > lo) { a = v; fa = 1; }
        if (!fb && cum > hi) { b = v; fb = 1; }
    }
    double diff = (double)(b - a);
    double r = (double)a + diff * t;
    if (t >= 0.5) r = (double)b - diff * (1.0 - t);
    return r;
}

static void percentiles_impl(const uint8_t *img, int64_t n, double q_lo, double q_hi, double *lo, double *hi)
{
    int64_t hist[256];
    memset(hist, 0, sizeof hist);
    for (int64_t i = 0; i < n; i++) hist[img[i]]++;
    *lo = percentile_from_hist(hist, n, q_lo);
    *hi = percentile_from_hist(hist, n, q_hi);
}

ORC_API void orc_percentiles(const uint8_t *img, int64_t n, double *lo, double *hi) { percentiles_impl(img, n, 0.10 * 100, 0.30 * 100, lo, hi); }

/* ------------------------------------------------------------------------------------------
 * a-8  cv.Canny(u8, lo, hi, apertureSize=3, L2gradient=True)  (edge_detection.py:85)
 *      OpenCV 4.x canny.cpp: Sobel 3x3 int16 BORDER_REPLICATE, mag = dx^2+dy^2 (int32), magnitude outside the
 *      image = 0, NMS with the integer tan(22.5) test, hysteresis over 8-neighbours.  dst is 255/0.
 *      The map (0 = weak candidate, 1 = suppressed, 2 = strong) is exported too for stage-level parity tests.
 * ---------------------------------------------------------------------------------------- */
static void canny_thresholds_impl(double lo, double hi, int l2, int *low, int *high)
{
    if (lo > hi) { double t = lo; lo = hi; hi = t; }
    if (l2) {                                /* L2gradient: squared magnitudes are compared */
        if (lo > 32767.0) lo = 32767.0;
        if (hi > 32767.0) hi = 32767.0;
        if (lo > 0) lo *= lo;
        if (hi > 0) hi *= hi;
    }
    *low = (int)floor(lo);
    *high = (int)floor(hi);
}

ORC_API void orc_canny_thresholds(double lo, double hi, int *low, int *high) { canny_thresholds_impl(lo, hi, 1, low, high); }

static void canny_impl(const uint8_t *src, uint8_t *dst, uint8_t *nms_out /* may be NULL */, int H, int W, double lo, double hi, int l2)
{
    int low, high;
    canny_thresholds_impl(lo, hi, l2, &low, &high);
    int64_t n = (int64_t)H * W;
    int16_t *gx = (int16_t *)malloc(n * 2), *gy = (int16_t *)malloc(n * 2);
    int32_t *mag = (int32_t *)calloc((size_t)(H + 2) * (W + 2), 4);
    uint8_t *map = (uint8_t *)malloc((size_t)(H + 2) * (W + 2));
    memset(map, 1, (size_t)(H + 2) * (W + 2));
#define SRC(y, x) src[(int64_t)((y) < 0 ? 0 : (y) >= H ? H - 1 : (y)) * W + ((x) < 0 ? 0 : (x) >= W ? W - 1 : (x))]
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int a = SRC(y - 1, x - 1), b = SRC(y - 1, x), c = SRC(y - 1, x + 1);
            int d = SRC(y, x - 1), f = SRC(y, x + 1);
            int g = SRC(y + 1, x - 1), h = SRC(y + 1, x), i = SRC(y + 1, x + 1);
            int dx = (c + 2 * f + i) - (a + 2 * d + g);
            int dy = (g + 2 * h + i) - (a + 2 * b + c);
            gx[(int64_t)y * W + x] = (int16_t)dx;
            gy[(int64_t)y * W + x] = (int16_t)dy;
            mag[(int64_t)(y + 1) * (W + 2) + (x + 1)] = l2 ? dx * dx + dy * dy : (dx < 0 ? -dx : dx) + (dy < 0 ? -dy : dy);
        }
#undef SRC
    int64_t *stack = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    int64_t sp = 0;
    const int TG22 = 13573;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int64_t mi = (int64_t)(y + 1) * (W + 2) + (x + 1);
            const int32_t *ma = mag + mi, *mp = ma - (W + 2), *mn = ma + (W + 2);
            int m = *ma;
            uint8_t res = 1;
            if (m > low) {
                int xs = gx[(int64_t)y * W + x], ys = gy[(int64_t)y * W + x];
                int ax = xs < 0 ? -xs : xs, ay = (ys < 0 ? -ys : ys) << 15;
                int tg22x = ax * TG22;
                int keep = 0;
                if (ay < tg22x) {
                    keep = (m > ma[-1] && m >= ma[1]);
                } else {
                    int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x) keep = (m > mp[0] && m >= mn[0]);
                    else {
                        int s = ((xs ^ ys) < 0) ? -1 : 1;
                        keep = (m > mp[-s] && m > mn[s]);
                    }
                }
                if (keep) res = (m > high) ? 2 : 0;
            }
            map[mi] = res;
            if (res == 2) stack[sp++] = mi;
        }
    if (nms_out)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) nms_out[(int64_t)y * W + x] = map[(int64_t)(y + 1) * (W + 2) + (x + 1)];
    const int64_t st = W + 2;
    const int64_t nb[8] = { -st - 1, -st, -st + 1, -1, 1, st - 1, st, st + 1 };
    while (sp > 0) {
        int64_t p = stack[--sp];
        for (int k = 0; k < 8; k++) {
            int64_t q = p + nb[k];
            if (map[q] == 0) { map[q] = 2; stack[sp++] = q; }
        }
    }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) dst[(int64_t)y * W + x] = (map[(int64_t)(y + 1) * (W + 2) + (x + 1)] == 2) ? 255 : 0;
    free(stack); free(map); free(mag); free(gx); free(gy);
}

ORC_API void orc_canny(const uint8_t *src, uint8_t *dst, uint8_t *nms_out, int H, int W, double lo, double hi) { canny_impl(src, dst, nms_out, H, W, lo, hi, 1); }

/* whole EdgeDetection.canny (edge_detection.py:70-86); edge01 gets 1 for edge (== 1.0f in the reference).
 * params (edge_detection.py:31-40; NULL = the defaults): { canny_low_ratio, canny_high_ratio, clahe_clip_limit,
 * bilateral_sigma_color, bilateral_sigma_space, use_L2_gradient }.  aperture_size 3, tile grid (4, 4), bilateral diameter 5
 * and the 3 x 3 Gaussian are structural in this restatement (as in the kernels). */
ORC_API void orc_edge_pipeline_ex(const float *plane, uint8_t *edge01, int H, int W, uint8_t *stages /* 4*H*W or NULL */, double *thr /*2 or NULL*/,
                                  const double *params /* 6 or NULL */)
{
    static const double defaults[6] = { 0.10, 0.30, 0.75, 75.0, 75.0, 1.0 };
    const double *p = params ? params : defaults;
    int64_t n = (int64_t)H * W;
    uint8_t *a = (uint8_t *)malloc(n), *b = (uint8_t *)malloc(n);
    orc_to_u8(plane, a, n);
    if (stages) memcpy(stages, a, n);
    clahe_impl(a, b, H, W, p[2]);
    if (stages) memcpy(stages + n, b, n);
    orc_gauss3(b, a, H, W);
    if (stages) memcpy(stages + 2 * n, a, n);
    bilateral5_impl(a, b, H, W, p[3], p[4]);
    if (stages) memcpy(stages + 3 * n, b, n);
    double lo, hi;
    percentiles_impl(b, n, p[0] * 100, p[1] * 100, &lo, &hi);
    if (thr) { thr[0] = lo; thr[1] = hi; }
    canny_impl(b, a, NULL, H, W, lo, hi, p[5] != 0.0);
    for (int64_t i = 0; i < n; i++) edge01[i] = a[i] ? 1 : 0;
    free(a); free(b);
}

ORC_API void orc_edge_pipeline(const float *plane, uint8_t *edge01, int H, int W, uint8_t *stages /* 4*H*W or NULL */, double *thr /*2 or NULL*/)
{
    int64_t n = (int64_t)H * W;
    uint8_t *a = (uint8_t *)malloc(n), *b = (uint8_t *)malloc(n);
    orc_to_u8(plane, a, n);
    if (stages) memcpy(stages, a, n);
    orc_clahe(a, b, H, W);
    if (stages) memcpy(stages + n, b, n);
    orc_gauss3(b, a, H, W);
    if (stages) memcpy(stages + 2 * n, a, n);
    orc_bilateral5(a, b, H, W);
    if (stages) memcpy(stages + 3 * n, b, n);
    double lo, hi;
    orc_percentiles(b, n, &lo, &hi);
    if (thr) { thr[0] = lo; thr[1] = hi; }
    orc_canny(b, a, NULL, H, W, lo, hi);
    for (int64_t i = 0; i < n; i++) edge01[i] = a[i] ? 1 : 0;
    free(a); free(b);
}

/* ------------------------------------------------------------------------------------------
 * a-9 / a-10  QuadTree._build_tree + get_leaves_and_states  (quadtree.py:93-165), utils.py:36-41.
 *      Restated as the reference does it: explicit-stack top-down DFS.  States: 0 leaf, 1 internal, 2 absent.
 * ---------------------------------------------------------------------------------------- */
ORC_API int orc_root_size(int H, int W)
{
    int n = H > W ? H : W;
    int lp;
    if (n <= 2) lp = n;
    else { lp = 1; while (lp * 2 < n) lp *= 2; } /* largest power of two strictly below n (utils.py:41) */
    return lp * 2;
}

static int has_edge(const uint8_t *edge, int H, int W, int x, int y, int s)
{
    int y1 = y + s > H ? H : y + s, x1 = x + s > W ? W : x + s;
    for (int yy = y; yy < y1; yy++)
        for (int xx = x; xx < x1; xx++)
            if (edge[(int64_t)yy * W + xx]) return 1;
    return 0;
}

typedef struct { int x, y, s; } qitem;

/* Pre-order walk producing leaves and states in one pass: equivalent to building the node tree and then
 * walking it (quadtree.py:136-165): a popped out-of-bounds child is the `None` entry -> state 2. */
ORC_API int orc_quadtree(const uint8_t *edge, int H, int W, int min_size, int max_size,
                         int32_t *leaves, int64_t leaf_cap, uint8_t *states, int64_t state_cap,
                         int64_t *n_leaves, int64_t *n_states, int *root_size)
{
    int root = orc_root_size(H, W);
    *root_size = root;
    int64_t cap = 64 * 4 + 16;
    qitem *stack = (qitem *)malloc(sizeof(qitem) * (size_t)cap);
    int64_t sp = 0, nl = 0, ns = 0;
    stack[sp++] = (qitem){ 0, 0, root };
    int rc = 0;
    while (sp > 0) {
        qitem it = stack[--sp];
        if (it.x >= W || it.y >= H) {
            if (ns >= state_cap) { rc = -2; break; }
            states[ns++] = 2;
            continue;
        }
        int split = it.s > max_size || (it.s > min_size && has_edge(edge, H, W, it.x, it.y, it.s));
        if (split) {
            if (ns >= state_cap) { rc = -2; break; }
            states[ns++] = 1;
            int h = it.s / 2;
            if (sp + 4 > cap) { cap *= 2; stack = (qitem *)realloc(stack, sizeof(qitem) * (size_t)cap); }
            stack[sp++] = (qitem){ it.x + h, it.y + h, h };
            stack[sp++] = (qitem){ it.x, it.y + h, h };
            stack[sp++] = (qitem){ it.x + h, it.y, h };
            stack[sp++] = (qitem){ it.x, it.y, h };
        } else {
            if (ns >= state_cap || nl >= leaf_cap) { rc = -2; break; }
            states[ns++] = 0;
            leaves[3 * nl] = it.x; leaves[3 * nl + 1] = it.y; leaves[3 * nl + 2] = it.s; nl++;
        }
    }
    free(stack);
    *n_leaves = nl; *n_states = ns;
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * a-11  normalise (common.py:161-174 via jpeg.py:387-390): (v - mid) * scale in float32.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_normalize(const float *plane, float *out, int64_t n, float mid, float scale)
{
    for (int64_t i = 0; i < n; i++) { float t = plane[i] - mid; out[i] = t * scale; }
}

/* np.pad(mode='reflect') index map (jpeg.py:402): i -> j = i mod 2(n-1); j >= n -> 2(n-1)-j; n==1 -> 0 */
static inline int reflect_pad_idx(int i, int n)
{
    if (n <= 1) return 0;
    int p = 2 * (n - 1);
    int j = i % p;
    return j >= n ? p - j : j;
}

/* ------------------------------------------------------------------------------------------
 * a-12  cv.dct (jpeg.py:471): orthonormal 2-D DCT-II.  OpenCV's factorisation is not reproducible, so the
 *       contract fixes the arithmetic: D = float32(alpha_k cos(pi (2n+1) k / 2s)) from float64,
 *       T = D.X and Y = T.D^T as k-ordered float32 fma chains starting from +0.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_dct_matrix(int s, float *D)
{
    for (int k = 0; k < s; k++) {
        double alpha = k == 0 ? sqrt(1.0 / (double)s) : sqrt(2.0 / (double)s);
        for (int n = 0; n < s; n++) D[k * s + n] = (float)(alpha * cos(3.14159265358979323846 * (double)(2 * n + 1) * (double)k / (2.0 * (double)s)));
    }
}

static void dct_block(const float *D, const float *X, float *T, float *Y, int s)
{
    /* T[i][j] = sum_k D[i][k] X[k][j] */
    for (int i = 0; i < s; i++) {
        float *t = T + i * s;
        for (int j = 0; j < s; j++) t[j] = 0.f;
        for (int k = 0; k < s; k++) {
            float d = D[i * s + k];
            const float *xr = X + k * s;
            for (int j = 0; j < s; j++) t[j] = fmaf(d, xr[j], t[j]);
        }
    }
    /* Y[i][j] = sum_k T[i][k] D[j][k] */
    for (int i = 0; i < s; i++) {
        const float *t = T + i * s;
        for (int j = 0; j < s; j++) {
            const float *dr = D + j * s;
            float acc = 0.f;
            for (int k = 0; k < s; k++) acc = fmaf(t[k], dr[k], acc);
            Y[i * s + j] = acc;
        }
    }
}

/* bare forward DCT contract on one float block: the per-leaf `cv.dct(block)` call of jpeg.py:471 (used by the
 * reference-structured CPU baseline, oracle/reference_structured.py) */
ORC_API void orc_dct_block(const float *D, const float *X, float *Y, int s)
{
    float *T = (float *)malloc(sizeof(float) * (size_t)s * s);
    dct_block(D, X, T, Y, s);
    free(T);
}

/* a-11 gather + a-12 DCT + a-13 quantise (np.round(f32 / int32) in float64, half-even, jpeg.py:501)
 * + a-15 zigzag gather (jpeg.py:579-588), per leaf, in leaf order.
 * norm: normalised plane (H,W); leaves (n,3) = x,y,s; qm_by_log2[l] / zz_by_log2[l]: tables for s = 1<<l.
 * coeffs receives sum(s*s) int32; dct_out (optional) receives the pre-quantisation floats in raster order. */
ORC_API int orc_blocks_encode(const float *norm, int H, int W, const int32_t *leaves, int64_t n_leaves,
                              const int32_t *const *qm_by_log2, const int32_t *const *zz_by_log2,
                              int32_t *coeffs, float *dct_out)
{
    float *Dm[16] = { 0 };
    int maxs = 1;
    for (int64_t i = 0; i < n_leaves; i++) if (leaves[3 * i + 2] > maxs) maxs = leaves[3 * i + 2];
    float *X = (float *)malloc(sizeof(float) * (size_t)maxs * maxs * 3);
    float *T = X + (size_t)maxs * maxs, *Y = T + (size_t)maxs * maxs;
    int64_t off = 0;
    int rc = 0;
    for (int64_t li = 0; li < n_leaves; li++) {
        int x = leaves[3 * li], y = leaves[3 * li + 1], s = leaves[3 * li + 2];
        int lg = 0; while ((1 << lg) < s) lg++;
        if ((1 << lg) != s || lg >= 16 || !qm_by_log2[lg] || !zz_by_log2[lg]) { rc = -3; break; }
        if (!Dm[lg]) { Dm[lg] = (float *)malloc(sizeof(float) * (size_t)s * s); orc_dct_matrix(s, Dm[lg]); }
        int hc = H - y < s ? H - y : s, wc = W - x < s ? W - x : s;
        for (int r = 0; r < s; r++) {
            int sr = y + reflect_pad_idx(r, hc);
            for (int c = 0; c < s; c++) X[r * s + c] = norm[(int64_t)sr * W + x + reflect_pad_idx(c, wc)];
        }
        dct_block(Dm[lg], X, T, Y, s);
        if (dct_out) memcpy(dct_out + off, Y, sizeof(float) * (size_t)s * s);
        const int32_t *qm = qm_by_log2[lg], *zz = zz_by_log2[lg];
        for (int i = 0; i < s * s; i++) {
            int src = zz[i];
            double q = (double)Y[src] / (double)qm[src];
            coeffs[off + i] = (int32_t)rint(q);
        }
        off += (int64_t)s * s;
    }
    for (int i = 0; i < 16; i++) free(Dm[i]);
    free(X);
    return rc;
}

/* ==========================================================================================
 * DECODE path (src/jpeg/jpeg.py:274-297) -- next-scope row, restated the same way.
 * ======================================================================================== */
#include "aej_inv_constants.h"

static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float idot3(const uint32_t *m, float a, float b, float c)   /* np.dot float32 row, k-ordered fma chain */
{
    float acc = a * bits2f(m[0]);
    acc = fmaf(b, bits2f(m[1]), acc);
    return fmaf(c, bits2f(m[2]), acc);
}
static inline float clip01(float v) { return v < 0.0f ? 0.0f : v > 1.0f ? 1.0f : v; }   /* np.clip (NaN stays NaN) */

/* common.py:62-92: float64 under numba typing, then max(0.0, min(1.0, x)) on the stored float32 (NaN -> 1.0) */
static inline float linear_to_srgb(float v)
{
    double d = (double)v, r;
    if (d <= 0.0031308) r = d * 12.92;
    else r = 1.055 * orc_pow(d, 1.0 / 2.4) - 0.055;
    float f = (float)r;
    float m = (f < 1.0f) ? f : 1.0f;      /* min(1.0, f): f if f < 1.0 else 1.0 */
    return (m > 0.0f) ? m : 0.0f;         /* max(0.0, m) */
}

/* common.py:94-129 */
static inline double pq_eotf(double v, double m2)
{
    const double c1 = 3424.0 / 4096.0, c2 = 2413.0 / 128.0, c3 = 2392.0 / 128.0, m1 = 2610.0 / 16384.0;
    double tmp = orc_pow(v, 1.0 / m2);
    double num = tmp - c1, den = c2 - c3 * tmp;
    if (num < 0.0) num = 0.0;
    if (den <= 0.0) den = 1e-12;
    return 10000.0 * orc_pow(num / den, 1.0 / m1);
}

static inline float ilin3(const uint32_t *m, float a, float b, float c)   /* float32 m0*a + m1*b + m2*c, no fma */
{
    float t = bits2f(m[0]) * a, u = bits2f(m[1]) * b;
    t = t + u;
    u = bits2f(m[2]) * c;
    return t + u;
}
static inline double ilin3d(const uint32_t *m, double a, double b, double c)
{
    double t = (double)bits2f(m[0]) * a, u = (double)bits2f(m[1]) * b;
    t = t + u;
    u = (double)bits2f(m[2]) * c;
    return t + u;
}

static void xyz_to_srgb(float X, float Y, float Z, float *o)   /* xyz.py:83-84 */
{
    float r = idot3(INV_XYZ_RGB_BITS + 0, X, Y, Z), g = idot3(INV_XYZ_RGB_BITS + 3, X, Y, Z), b = idot3(INV_XYZ_RGB_BITS + 6, X, Y, Z);
    o[0] = linear_to_srgb(r); o[1] = linear_to_srgb(g); o[2] = linear_to_srgb(b);
}

static void color_inv_px(int space, float a, float b, float c, float *o)
{
    const uint32_t *m = NULL;
    switch (space) {
    case SP_YCBCR: m = INV_YCBCR_BITS; break;       /* ycbcr.py:80-83 */
    case SP_YCOCG: m = INV_YCOCG_BITS; break;       /* ycocg.py:101-104 */
    case SP_YCOCG_R: m = INV_YCOCG_R_BITS; break;   /* ycocg.py:140-143 */
    default: break;
    }
    if (m) {
        o[0] = clip01(idot3(m + 0, a, b, c)); o[1] = clip01(idot3(m + 3, a, b, c)); o[2] = clip01(idot3(m + 6, a, b, c));
        return;
    }
    if (space == SP_XYZ) { xyz_to_srgb(a, b, c, o); return; }   /* XYZ.xyz_to_srgb, xyz.py:83-84 */
    if (space == SP_OKLAB) {   /* oklab.py:92-96 */
        float lp = idot3(INV_OK_LAB_LMSP_BITS + 0, a, b, c), mp = idot3(INV_OK_LAB_LMSP_BITS + 3, a, b, c), sp = idot3(INV_OK_LAB_LMSP_BITS + 6, a, b, c);
        /* np.power(f32, 3): x^3 rounded once from float64 */
        double dl = lp, dm = mp, ds = sp;
        float l = (float)(dl * dl * dl), mm = (float)(dm * dm * dm), s = (float)(ds * ds * ds);
        float X = idot3(INV_OK_LMS_XYZ_BITS + 0, l, mm, s), Y = idot3(INV_OK_LMS_XYZ_BITS + 3, l, mm, s), Z = idot3(INV_OK_LMS_XYZ_BITS + 6, l, mm, s);
        xyz_to_srgb(X, Y, Z, o);
        return;
    }
    if (space == SP_ICTCP || space == SP_ICACB) {   /* ictcp.py:101-137, icacb.py:101-137 */
        const uint32_t *m2 = space == SP_ICTCP ? INV_ICT_LMSP_BITS : INV_ICA_RGBP_BITS;
        const uint32_t *m1 = space == SP_ICTCP ? INV_ICT_LMS_XYZ_BITS : INV_ICA_RGB_XYZ_BITS;
        float Lp = ilin3(m2 + 0, a, b, c), Mp = ilin3(m2 + 3, a, b, c), Sp = ilin3(m2 + 6, a, b, c);
        const double pm2 = 2523.0 / 32.0;
        double L = pq_eotf((double)Lp, pm2), M = pq_eotf((double)Mp, pm2), S = pq_eotf((double)Sp, pm2);
        float X = (float)ilin3d(m1 + 0, L, M, S), Y = (float)ilin3d(m1 + 3, L, M, S), Z = (float)ilin3d(m1 + 6, L, M, S);
        xyz_to_srgb(X, Y, Z, o);
        return;
    }
    {   /* JzAzBz, jzazbz.py:136-176 */
        const double bb = 1.15, gg = 0.66, d = -0.56, d0 = 1.6295499532821566e-11, p = 1.7 * 2523.0 / 32.0;
        const uint32_t *m2 = INV_JZ_LMSP_BITS, *m1 = INV_JZ_LMS_XYZ_BITS;
        double Iz = ((double)a + d0) / (1.0 + d - d * ((double)a + d0));
        /* numba typing: M[i,0]*Iz is float64, M[i,1]*Az and M[i,2]*Bz are float32 products */
        double Lp = ((double)bits2f(m2[0]) * Iz + (double)(bits2f(m2[1]) * b)) + (double)(bits2f(m2[2]) * c);
        double Mp = ((double)bits2f(m2[3]) * Iz + (double)(bits2f(m2[4]) * b)) + (double)(bits2f(m2[5]) * c);
        double Sp = ((double)bits2f(m2[6]) * Iz + (double)(bits2f(m2[7]) * b)) + (double)(bits2f(m2[8]) * c);
        double L = pq_eotf(Lp, p), M = pq_eotf(Mp, p), S = pq_eotf(Sp, p);
        double Xp = ilin3d(m1 + 0, L, M, S), Yp = ilin3d(m1 + 3, L, M, S), Zp = ilin3d(m1 + 6, L, M, S);
        double X = (Xp + (bb - 1.0) * Zp) / bb;
        double Y = (Yp + (gg - 1.0) * X) / gg;
        xyz_to_srgb((float)X, (float)Y, (float)Zp, o);
    }
}

ORC_API void orc_color_inverse(int space, const float *in, float *out, int64_t n)
{
    for (int64_t i = 0; i < n; i++) color_inv_px(space, in[3 * i], in[3 * i + 1], in[3 * i + 2], out + 3 * i);
}

/* Jpeg._block_merge's walk (jpeg.py:424-448): positions of the leaves from their sizes + geometry */
ORC_API int64_t orc_leaf_positions(const int32_t *sizes, int64_t n, int root, int H, int W, int32_t *xy)
{
    int64_t cap = 64 * 4 + 16, sp = 0, li = 0;
    qitem *stack = (qitem *)malloc(sizeof(qitem) * (size_t)cap);
    stack[sp++] = (qitem){ 0, 0, root };
    while (sp > 0 && li < n) {
        qitem it = stack[--sp];
        if (it.x >= W || it.y >= H || it.s == 0) continue;
        if (it.s == sizes[li]) { xy[2 * li] = it.x; xy[2 * li + 1] = it.y; li++; }
        else {
            int h = it.s / 2;
            if (sp + 4 > cap) { cap *= 2; stack = (qitem *)realloc(stack, sizeof(qitem) * (size_t)cap); }
            stack[sp++] = (qitem){ it.x + h, it.y + h, h };
            stack[sp++] = (qitem){ it.x, it.y + h, h };
            stack[sp++] = (qitem){ it.x + h, it.y, h };
            stack[sp++] = (qitem){ it.x, it.y, h };
        }
    }
    free(stack);
    return li;
}

/* inverse zigzag + _dequantize (jpeg.py:663-670, 508-529) + cv.idct (jpeg.py:483) + merge/crop + _denormalize
 * (jpeg.py:441-455, common.py:176-189).  IDCT contract: T = D^T.Y, X = T.D as k-ordered float32 fma chains from +0. */
ORC_API int orc_blocks_decode(const int32_t *coeffs, const int32_t *leaves, int64_t n_leaves, const int32_t *const *qm_by_log2,
                              const int32_t *const *zz_by_log2, float mid, float scale, int H, int W, float *plane)
{
    float *Dm[16] = { 0 };
    int maxs = 1;
    for (int64_t i = 0; i < n_leaves; i++) if (leaves[3 * i + 2] > maxs) maxs = leaves[3 * i + 2];
    float *Y = (float *)malloc(sizeof(float) * (size_t)maxs * maxs * 3);
    float *T = Y + (size_t)maxs * maxs, *X = T + (size_t)maxs * maxs;
    int64_t off = 0;
    int rc = 0;
    for (int64_t li = 0; li < n_leaves; li++) {
        int x = leaves[3 * li], y = leaves[3 * li + 1], s = leaves[3 * li + 2];
        int lg = 0; while ((1 << lg) < s) lg++;
        if ((1 << lg) != s || lg >= 16 || !qm_by_log2[lg] || !zz_by_log2[lg]) { rc = -3; break; }
        if (!Dm[lg]) { Dm[lg] = (float *)malloc(sizeof(float) * (size_t)s * s); orc_dct_matrix(s, Dm[lg]); }
        const float *D = Dm[lg];
        const int32_t *qm = qm_by_log2[lg], *zz = zz_by_log2[lg];
        for (int i = 0; i < s * s; i++) Y[zz[i]] = (float)(coeffs[off + i] * qm[zz[i]]);
        for (int n = 0; n < s; n++) {          /* T[n][j] = sum_k D[k][n] Y[k][j] */
            float *t = T + n * s;
            for (int j = 0; j < s; j++) t[j] = 0.f;
            for (int k = 0; k < s; k++) {
                float d = D[k * s + n];
                const float *yr = Y + k * s;
                for (int j = 0; j < s; j++) t[j] = fmaf(d, yr[j], t[j]);
            }
        }
        for (int n = 0; n < s; n++) {          /* X[n][m] = sum_k T[n][k] D[k][m] */
            float *xr = X + n * s;
            for (int m = 0; m < s; m++) xr[m] = 0.f;
            for (int k = 0; k < s; k++) {
                float t = T[n * s + k];
                const float *dr = D + k * s;
                for (int m = 0; m < s; m++) xr[m] = fmaf(t, dr[m], xr[m]);
            }
        }
        for (int n = 0; n < s && y + n < H; n++)
            for (int m = 0; m < s && x + m < W; m++) {
                float v = X[n * s + m] / scale;
                plane[(int64_t)(y + n) * W + x + m] = v + mid;
            }
        off += (int64_t)s * s;
    }
    for (int i = 0; i < 16; i++) free(Dm[i]);
    free(Y);
    return rc;
}

/* Jpeg._upsample (jpeg.py:340-354): cv.resize(layer, (W, H), INTER_LINEAR), float32.  OpenCV 4.x resizeGeneric_ with
 * HResizeLinear / VResizeLinear: half-pixel centres; x index clamped with the weight forced to 0 at both ends, y rows
 * clipped with the weight left as computed; horizontal pass then vertical pass, separate mul / add. */
ORC_API void orc_upsample_linear(const float *src, int h, int w, float *dst, int H, int W)
{
    if (h == H && w == W) { memcpy(dst, src, sizeof(float) * (size_t)H * W); return; }
    double scale_x = 1.0 / ((double)W / (double)w), scale_y = 1.0 / ((double)H / (double)h);
    int *xofs = (int *)malloc(sizeof(int) * (size_t)W);
    float *xa = (float *)malloc(sizeof(float) * (size_t)W * 2);
    for (int dx = 0; dx < W; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= w - 1) { fx = 0.f; sx = w - 1; }
        xofs[dx] = sx; xa[2 * dx] = 1.f - fx; xa[2 * dx + 1] = fx;
    }
    float *r0 = (float *)malloc(sizeof(float) * (size_t)W * 2), *r1 = r0 + W;
    for (int dy = 0; dy < H; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= (float)sy;
        float b0 = 1.f - fy, b1 = fy;
        int y0 = sy < 0 ? 0 : sy > h - 1 ? h - 1 : sy, y1 = sy + 1 < 0 ? 0 : sy + 1 > h - 1 ? h - 1 : sy + 1;
        const float *s0 = src + (int64_t)y0 * w, *s1 = src + (int64_t)y1 * w;
        for (int dx = 0; dx < W; dx++) {
            int sx = xofs[dx];
            int sx1 = sx + 1 < w ? sx + 1 : sx;          /* weight is 0 there */
            float a0 = xa[2 * dx], a1 = xa[2 * dx + 1];
            if (sx + 1 < w) {
                float p = s0[sx] * a0, q = s0[sx1] * a1; r0[dx] = p + q;
                p = s1[sx] * a0; q = s1[sx1] * a1; r1[dx] = p + q;
            } else {                                     /* dx >= xmax: D = S[sx] * ONE */
                r0[dx] = s0[sx] * 1.0f; r1[dx] = s1[sx] * 1.0f;
            }
        }
        float *d = dst + (int64_t)dy * W;
        for (int dx = 0; dx < W; dx++) { float p = r0[dx] * b0, q = r1[dx] * b1; d[dx] = p + q; }
    }
    free(xofs); free(xa); free(r0);
}

/* bare IDCT contract on one float block (used by the fixture generator as the cv2.idct stand-in) */
ORC_API void orc_idct_block(const float *D, const float *Y, float *X, int s)
{
    float *T = (float *)malloc(sizeof(float) * (size_t)s * s);
    for (int n = 0; n < s; n++) {
        float *t = T + n * s;
        for (int j = 0; j < s; j++) t[j] = 0.f;
        for (int k = 0; k < s; k++) { float d = D[k * s + n]; const float *yr = Y + k * s; for (int j = 0; j < s; j++) t[j] = fmaf(d, yr[j], t[j]); }
    }
    for (int n = 0; n < s; n++) {
        float *xr = X + n * s;
        for (int m = 0; m < s; m++) xr[m] = 0.f;
        for (int k = 0; k < s; k++) { float t = T[n * s + k]; const float *dr = D + k * s; for (int m = 0; m < s; m++) xr[m] = fmaf(t, dr[m], xr[m]); }
    }
    free(T);
}
