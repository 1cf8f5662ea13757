// decode.hip -- the DECODE path of the codec (src/jpeg/jpeg.py:274-297) on gfx950: inverse zigzag + dequantise + inverse DCT +
// block merge + denormalise (jpeg.py:410-459, 473-483, 508-529, 663-670), bilinear up-sampling of the chroma layers
// (jpeg.py:340-354) and the inverse colour transforms (src/color/*.py).  Entropy decoding (JSON, bit unpacking, zlib) stays on
// the host.  Same numerics rules as the encode kernels: every float op written out, fma only where written.
//
// IDCT contract: with D the float32 DCT-II basis, Yq[k][j] = float(q[k][j] * Q[k][j]),
//     T[n][j] = fma-chain over k of D[k][n] * Yq[k][j],   X[n][m] = fma-chain over k of T[n][k] * D[k][m],
// then plane[y+n][x+m] = X[n][m] / scale + mid for the in-bounds part of the leaf.
#include "aej_common.h"
#include "aej_launch.h"
#include "aej_bigblock.h"
#include "aej_mfma.h"
#include "aej_devmath.h"
#include "inv_constants.h"

namespace aej {


// ------------------------------------------------------------------------------------------------
// leaf tables -> per-size work lists (per plane segment; order inside a segment is irrelevant for the result)
// ------------------------------------------------------------------------------------------------
struct WorkPtrsD { LeafWork *w[kMaxSizes]; };

// A leaf table that does not fit the plan (more leaves than the layer can hold, a size outside the settings' block range, an origin
// outside the layer, a coefficient offset outside the layer's span, more leaves of one size than its work list holds) sets *bad instead of writing out of bounds; aej_decode_batch reports it.
__global__ __launch_bounds__(256) void k_work_from_tables(Geom g, QtGeom q, const int *__restrict__ leaves, const long long *__restrict__ counts,
                                                          WorkPtrsD wp, int *__restrict__ work_count, int bmin_log2, int *__restrict__ bad)
{
    const int l = blockIdx.y, b = blockIdx.z, plane = b * 3 + l;
    long long n = counts[(long long)plane * 4 + 1];
    if (n < 0 || n > q.leaf_cap[l]) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *bad = 1;
        n = 0;
    }
    const int4 *tab = reinterpret_cast<const int4 *>(leaves) + (long long)b * q.leaf_stride + q.leaf_off[l];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        int4 lf = tab[i];
        int k = (31 - __clz(lf.z)) - bmin_log2;
        if (k < 0 || k >= q.nsizes || !wp.w[k] || lf.z != (q.bmin << k)) { k = -1; *bad = 1; }
        // the IDCT kernels clip a leaf at the layer's right / bottom edge only: its origin has to lie inside the layer, and its
        // coefficients inside the layer's span of `coeffs`
        else if (lf.x < 0 || lf.y < 0 || lf.x >= g.w[l] || lf.y >= g.h[l] || lf.w < 0 || (long long)lf.w + (long long)lf.z * lf.z > q.coeff_cap[l]) { k = -1; *bad = 1; }
        // one atomic per wave and block size instead of one per leaf (all lanes of a wave work on the same plane): a few
        // counters shared by 10^5 leaves per image serialise otherwise.  List order is irrelevant to the decode.
        const int lane = threadIdx.x & 63;
        int pos = 0;
        for (int kk = 0; kk < q.nsizes; kk++) {
            const unsigned long long m = __ballot(k == kk);
            if (m == 0) continue;
            const int leader = __ffsll((long long)m) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(&work_count[plane * kMaxSizes + kk], __popcll(m));
            base = __shfl(base, leader);
            if (k == kk) pos = base + __popcll(m & ((1ull << lane) - 1ull));
        }
        if (k >= 0) {
            const long long cap = (l < 2 ? q.work_off[l + 1][k] : q.work_stride[k]) - q.work_off[l][k];      // this plane's segment of the list
            if (pos < cap) wp.w[k][(long long)b * q.work_stride[k] + q.work_off[l][k] + pos] = pack_work(lf.x, lf.y, lf.w);
            else *bad = 1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// shared prologue helpers (same scheme as dct.hip)
// ------------------------------------------------------------------------------------------------
struct LayerTabD {
    int w[3], h[3];
    long long poff[3], coff[3], woff[3];
    float mid[3], scale[3];
};

__device__ __forceinline__ int wave_incl_scan_d(int v, int /*lane*/) { return wave_scan_incl(v); }      // (threads 0..63 of the workgroup: a whole wave)

__device__ __forceinline__ void idct_prologue(const Geom &g, const QtGeom &q, const IdctArgs &a, int *s_pref, LayerTabD &lt)
{
    if (threadIdx.x < 3) {
        const int l = threadIdx.x;
        lt.w[l] = g.w[l]; lt.h[l] = g.h[l];
        lt.poff[l] = g.poff[l]; lt.coff[l] = q.coeff_off[l]; lt.woff[l] = q.work_off[l][a.k];
        lt.mid[l] = a.mid[l]; lt.scale[l] = a.scale[l];
    }
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        int carry = 0;
        if (lane == 0) s_pref[0] = 0;
        for (int base = 0; base < a.nplanes; base += 64) {
            int p = base + lane;
            int v = p < a.nplanes ? a.work_count[(long long)p * kMaxSizes + a.k] : 0;
            int inc = wave_incl_scan_d(v, lane);
            if (p < a.nplanes) s_pref[p + 1] = carry + inc;
            carry += __builtin_amdgcn_readlane(inc, 63);
        }
    }
    __syncthreads();
}

__device__ __forceinline__ int4 fetch_item_d(const IdctArgs &a, long long work_stride, const LayerTabD &lt, const int *s_pref, long long item)
{
    int lo = 0, hi = a.nplanes;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if ((long long)s_pref[mid] <= item) lo = mid; else hi = mid;
    }
    const int b = lo / 3, l = lo - 3 * b;
    return unpack_work(lo, a.work[(long long)b * work_stride + lt.woff[l] + (item - s_pref[lo])]);
}

// ------------------------------------------------------------------------------------------------
// small blocks: S threads per leaf (column pass, LDS transpose, row pass)
// ------------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void k_idct_small(Geom g, QtGeom q, IdctArgs a)
{
    constexpr int LPB = 256 / S;
    constexpr int SS = S * S;
    __shared__ float sT[LPB * S * (S + 1)];
    __shared__ float sD[SS];
    __shared__ int sZi[SS];
    __shared__ int sQm[3 * SS];
    __shared__ LayerTabD lt;
    extern __shared__ int s_pref[];
    const int tid = threadIdx.x;
    for (int i = tid; i < SS; i += 256) { sD[i] = a.D[i]; sZi[i] = a.zzinv[i]; }
    for (int i = tid; i < 3 * SS; i += 256) sQm[i] = a.qm[i / SS] ? a.qm[i / SS][i % SS] : 1;
    idct_prologue(g, q, a, s_pref, lt);
    const long long count = s_pref[a.nplanes];
    const long long wstride = q.work_stride[a.k];
    const int slot = tid / S, j = tid % S;
    for (long long base = (long long)blockIdx.x * LPB; base < count; base += (long long)gridDim.x * LPB) {
        const bool active = base + slot < count;
        int4 wk = make_int4(0, 0, 0, 0);
        int layer = 0, b = 0;
        if (active) {
            wk = fetch_item_d(a, wstride, lt, s_pref, base + slot);
            b = wk.x / 3; layer = wk.x - 3 * b;
            const int *cf = a.coeffs + (long long)b * q.coeff_stride + lt.coff[layer] + wk.w;
            float y[S];                                  // column j of the dequantised block
#pragma unroll
            for (int k = 0; k < S; k++) y[k] = (float)(cf[sZi[k * S + j]] * sQm[layer * SS + k * S + j]);
#pragma unroll
            for (int n = 0; n < S; n++) {                // T[n][j] = sum_k D[k][n] Y[k][j]
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < S; k++) acc = __builtin_fmaf(sD[k * S + n], y[k], acc);
                sT[(slot * S + n) * (S + 1) + j] = acc;
            }
        }
        __syncthreads();
        if (active) {
            const int w = lt.w[layer], h = lt.h[layer];
            float *dst = a.planes + (long long)b * g.pstride + lt.poff[layer];
            const float mid = lt.mid[layer], scale = lt.scale[layer];
            float t[S];                                  // row n = j of T
#pragma unroll
            for (int k = 0; k < S; k++) t[k] = sT[(slot * S + j) * (S + 1) + k];
            const int yy = wk.z + j;
#pragma unroll
            for (int m = 0; m < S; m++) {                // X[n][m] = sum_k T[n][k] D[k][m]
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < S; k++) acc = __builtin_fmaf(t[k], sD[k * S + m], acc);
                if (yy < h && wk.y + m < w) { float v = acc / scale; dst[(long long)yy * w + wk.y + m] = v + mid; }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// large blocks: MFMA, same tiling as k_dct_mfma.  P = Yq^T.D (P[r][c] = T[c][r]); X = T.D reads A[i][k] = P[k][I0+i];
// B[k][j] = D[k][J0+j] lives in S/2 registers.
// ------------------------------------------------------------------------------------------------
template <int S>
struct IdctCfg {
    static constexpr int NT = S / 32;
    static constexpr int TPW = S == 128 ? 2 : 1;
    static constexpr int NWAVES = NT * NT / TPW;
    static constexpr int NTHREADS = NWAVES * 64;
};

template <int S>
__global__ __launch_bounds__(IdctCfg<S>::NTHREADS) void k_idct_mfma(Geom g, QtGeom q, IdctArgs a)
{
    using C = IdctCfg<S>;
    constexpr int NT = C::NT, TPW = C::TPW, NTHREADS = C::NTHREADS;
    constexpr int SS = S * S;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sY = smem, *sP = smem + SS;
    LayerTabD &lt = *reinterpret_cast<LayerTabD *>(smem + 2 * SS);
    int *s_pref = reinterpret_cast<int *>(smem + 2 * SS) + (sizeof(LayerTabD) + 3) / 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wj = wave % NT, wi0 = wave / NT, J0 = wj * 32, li = lane & 31, lh = lane >> 5;
    float dreg[S / 2];
#pragma unroll
    for (int s = 0; s < S / 2; s++) dreg[s] = a.D[(2 * s + lh) * S + J0 + li];
    idct_prologue(g, q, a, s_pref, lt);
    const long long count = s_pref[a.nplanes];
    const long long wstride = q.work_stride[a.k];
    for (long long item = blockIdx.x; item < count; item += gridDim.x) {
        const int4 wk = fetch_item_d(a, wstride, lt, s_pref, item);
        const int b = wk.x / 3, layer = wk.x - 3 * b;
        const int w = lt.w[layer], h = lt.h[layer];
        const int *cf = a.coeffs + (long long)b * q.coeff_stride + lt.coff[layer] + wk.w;
        const int *qm = a.qm[layer];
        for (int i = tid; i < SS; i += NTHREADS) {
            const int r = a.zz[i];
            sY[r] = (float)(cf[i] * qm[r]);
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TPW; t++) {
            const int I0 = (wi0 + t * (NT / TPW)) * 32;
            floatx16 acc1[1];
#pragma unroll
            for (int r = 0; r < 16; r++) acc1[0][r] = 0.f;
            mfma_chain<S, 1, kMfmaPF>(sY, 0, I0 + li, lh, dreg, acc1);
            const floatx16 acc = acc1[0];
#pragma unroll
            for (int r = 0; r < 16; r++) sP[(I0 + (r & 3) + 8 * (r >> 2) + 4 * lh) * S + J0 + li] = acc[r];
        }
        __syncthreads();
        float *dst = a.planes + (long long)b * g.pstride + lt.poff[layer];
        const float mid = lt.mid[layer], scale = lt.scale[layer];
#pragma unroll
        for (int t = 0; t < TPW; t++) {
            const int I0 = (wi0 + t * (NT / TPW)) * 32;
            floatx16 acc1[1];
#pragma unroll
            for (int r = 0; r < 16; r++) acc1[0][r] = 0.f;
            mfma_chain<S, 1, kMfmaPF>(sP, 0, I0 + li, lh, dreg, acc1);
            const floatx16 acc = acc1[0];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int yy = wk.z + I0 + (r & 3) + 8 * (r >> 2) + 4 * lh, xx = wk.y + J0 + li;
                if (yy < h && xx < w) { float v = acc[r] / scale; dst[(long long)yy * w + xx] = v + mid; }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// S = 256 (aej_bigblock.h): T = D^T.Y into this workgroup's scratch, then X = T.D, de-normalise, merge
// ------------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void k_idct_big(Geom g, QtGeom q, IdctArgs a)
{
    __shared__ BigTileLds L;
    __shared__ LayerTabD lt;
    extern __shared__ int s_pref[];
    idct_prologue(g, q, a, s_pref, lt);
    const long long count = s_pref[a.nplanes];
    const long long wstride = q.work_stride[a.k];
    float *T = a.scratch + (long long)blockIdx.x * S * S;
    for (long long item = blockIdx.x; item < count; item += gridDim.x) {
        const int4 wk = fetch_item_d(a, wstride, lt, s_pref, item);
        const int b = wk.x / 3, layer = wk.x - 3 * b;
        const int w = lt.w[layer], h = lt.h[layer];
        const int *cf = a.coeffs + (long long)b * q.coeff_stride + lt.coff[layer] + wk.w;
        const int *qm = a.qm[layer];
        const float *D = a.D;
        big_product<S>(L,
            [&](int i, int k) { return D[k * S + i]; },
            [&](int k, int j) { const int r = k * S + j; return (float)(cf[a.zzinv[r]] * qm[r]); },      // _dequantize, jpeg.py:508-529
            [&](int i, int j, float v) { T[i * S + j] = v; });
        big_scratch_sync();
        float *dst = a.planes + (long long)b * g.pstride + lt.poff[layer];
        const float mid = lt.mid[layer], scale = lt.scale[layer];
        big_product<S>(L,
            [&](int i, int k) { return T[i * S + k]; },
            [&](int k, int j) { return D[k * S + j]; },
            [&](int i, int j, float v) {
                const int yy = wk.z + i, xx = wk.y + j;
                if (yy < h && xx < w) { float t = v / scale; dst[(long long)yy * w + xx] = t + mid; }
            });
        big_scratch_sync();
    }
}

// ------------------------------------------------------------------------------------------------
// inverse colour transforms (conversion.py:122-124 and the per-space x_to_srgb functions)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bits2f(unsigned u) { return __uint_as_float(u); }
__device__ __forceinline__ float idot3(const unsigned *m, float a, float b, float c)
{
    float acc = a * bits2f(m[0]);
    acc = __builtin_fmaf(b, bits2f(m[1]), acc);
    return __builtin_fmaf(c, bits2f(m[2]), acc);
}
__device__ __forceinline__ float clip01(float v) { return v < 0.0f ? 0.0f : v > 1.0f ? 1.0f : v; }
__device__ __forceinline__ float linear_to_srgb(float v)          // common.py:62-92
{
    double d = (double)v, r;
    if (d <= 0.0031308) r = d * 12.92;
    else r = 1.055 * dev_pow(d, 1.0 / 2.4) - 0.055;
    float f = (float)r;
    float m = (f < 1.0f) ? f : 1.0f;
    return (m > 0.0f) ? m : 0.0f;
}
__device__ __forceinline__ double pq_eotf(double v, double m2)    // common.py:94-129
{
    const double c1 = 3424.0 / 4096.0, c2 = 2413.0 / 128.0, c3 = 2392.0 / 128.0, m1 = 2610.0 / 16384.0;
    double tmp = dev_pow(v, 1.0 / m2);
    double num = tmp - c1, den = c2 - c3 * tmp;
    if (num < 0.0) num = 0.0;
    if (den <= 0.0) den = 1e-12;
    return 10000.0 * dev_pow(num / den, 1.0 / m1);
}
__device__ __forceinline__ float ilin3(const unsigned *m, float a, float b, float c)
{
    float t = bits2f(m[0]) * a, u = bits2f(m[1]) * b;
    t = t + u;
    u = bits2f(m[2]) * c;
    return t + u;
}
__device__ __forceinline__ double ilin3d(const unsigned *m, double a, double b, double c)
{
    double t = (double)bits2f(m[0]) * a, u = (double)bits2f(m[1]) * b;
    t = t + u;
    u = (double)bits2f(m[2]) * c;
    return t + u;
}
__device__ __forceinline__ void xyz_to_srgb(float X, float Y, float Z, float &r, float &g, float &b)
{
    r = linear_to_srgb(idot3(INV_XYZ_RGB_BITS + 0, X, Y, Z));
    g = linear_to_srgb(idot3(INV_XYZ_RGB_BITS + 3, X, Y, Z));
    b = linear_to_srgb(idot3(INV_XYZ_RGB_BITS + 6, X, Y, Z));
}

template <int SPACE>
__device__ __forceinline__ void color_inv_px(float a, float b, float c, float &r, float &g, float &bl)
{
    if constexpr (SPACE <= 2) {
        const unsigned *m = SPACE == 0 ? INV_YCBCR_BITS : SPACE == 1 ? INV_YCOCG_BITS : INV_YCOCG_R_BITS;
        r = clip01(idot3(m + 0, a, b, c)); g = clip01(idot3(m + 3, a, b, c)); bl = clip01(idot3(m + 6, a, b, c));
    } else if constexpr (SPACE == 3) {
        float lp = idot3(INV_OK_LAB_LMSP_BITS + 0, a, b, c), mp = idot3(INV_OK_LAB_LMSP_BITS + 3, a, b, c), sp = idot3(INV_OK_LAB_LMSP_BITS + 6, a, b, c);
        double dl = lp, dm = mp, ds = sp;
        float l = (float)(dl * dl * dl), mm = (float)(dm * dm * dm), s = (float)(ds * ds * ds);
        xyz_to_srgb(idot3(INV_OK_LMS_XYZ_BITS + 0, l, mm, s), idot3(INV_OK_LMS_XYZ_BITS + 3, l, mm, s), idot3(INV_OK_LMS_XYZ_BITS + 6, l, mm, s), r, g, bl);
    } else if constexpr (SPACE == 4 || SPACE == 5) {
        const unsigned *m2 = SPACE == 4 ? INV_ICT_LMSP_BITS : INV_ICA_RGBP_BITS;
        const unsigned *m1 = SPACE == 4 ? INV_ICT_LMS_XYZ_BITS : INV_ICA_RGB_XYZ_BITS;
        float Lp = ilin3(m2 + 0, a, b, c), Mp = ilin3(m2 + 3, a, b, c), Sp = ilin3(m2 + 6, a, b, c);
        const double pm2 = 2523.0 / 32.0;
        double L = pq_eotf((double)Lp, pm2), M = pq_eotf((double)Mp, pm2), S = pq_eotf((double)Sp, pm2);
        xyz_to_srgb((float)ilin3d(m1 + 0, L, M, S), (float)ilin3d(m1 + 3, L, M, S), (float)ilin3d(m1 + 6, L, M, S), r, g, bl);
    } else if constexpr (SPACE == 7) {    // XYZ -> sRGB, xyz.py:83-84 (helper space of color.convert)
        xyz_to_srgb(a, b, c, r, g, bl);
    } else {
        const double bb = 1.15, gg = 0.66, d = -0.56, d0 = 1.6295499532821566e-11, p = 1.7 * 2523.0 / 32.0;
        const unsigned *m2 = INV_JZ_LMSP_BITS, *m1 = INV_JZ_LMS_XYZ_BITS;
        double Iz = ((double)a + d0) / (1.0 + d - d * ((double)a + d0));
        double Lp = ((double)bits2f(m2[0]) * Iz + (double)(bits2f(m2[1]) * b)) + (double)(bits2f(m2[2]) * c);
        double Mp = ((double)bits2f(m2[3]) * Iz + (double)(bits2f(m2[4]) * b)) + (double)(bits2f(m2[5]) * c);
        double Sp = ((double)bits2f(m2[6]) * Iz + (double)(bits2f(m2[7]) * b)) + (double)(bits2f(m2[8]) * c);
        double L = pq_eotf(Lp, p), M = pq_eotf(Mp, p), S = pq_eotf(Sp, p);
        double Xp = ilin3d(m1 + 0, L, M, S), Yp = ilin3d(m1 + 3, L, M, S), Zp = ilin3d(m1 + 6, L, M, S);
        double X = (Xp + (bb - 1.0) * Zp) / bb;
        double Y = (Yp + (gg - 1.0) * X) / gg;
        xyz_to_srgb((float)X, (float)Y, (float)Zp, r, g, bl);
    }
}

template <int SPACE>
__global__ __launch_bounds__(256) void k_color_inverse(const float *__restrict__ in, float *__restrict__ out, long long n)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float r, g, b;
        color_inv_px<SPACE>(in[3 * i], in[3 * i + 1], in[3 * i + 2], r, g, b);
        out[3 * i] = r; out[3 * i + 1] = g; out[3 * i + 2] = b;
    }
}

// cv.resize(layer, (W, H), INTER_LINEAR) sample at (dx, dy) (OpenCV resizeGeneric_, HResizeLinear + VResizeLinear, float32).
// scale_x / scale_y = 1.0 / ((double)W / w), 1.0 / ((double)H / h): formed once on the host (the same IEEE double operations).
struct UpScale { double sx[3], sy[3]; };

struct RowTaps { const float *s0, *s1; float b0, b1; };

__device__ __forceinline__ RowTaps row_taps(const float *__restrict__ src, int h, int w, double scale_y, int dy)
{
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy -= (float)sy;
    RowTaps t;
    t.b0 = 1.f - fy; t.b1 = fy;
    const int y0 = sy < 0 ? 0 : sy > h - 1 ? h - 1 : sy, y1 = sy + 1 < 0 ? 0 : sy + 1 > h - 1 ? h - 1 : sy + 1;
    t.s0 = src + (long long)y0 * w; t.s1 = src + (long long)y1 * w;
    return t;
}

struct ColTaps { int sx; float a0, a1; bool two; };

__device__ __forceinline__ ColTaps col_taps(int w, double scale_x, int dx)
{
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= w - 1) { fx = 0.f; sx = w - 1; }
    ColTaps c;
    c.sx = sx; c.a0 = 1.f - fx; c.a1 = fx; c.two = sx + 1 < w;
    return c;
}

__device__ __forceinline__ float bilinear_at(const RowTaps &t, const ColTaps &c)
{
    float r0, r1;
    if (c.two) {
        float p = t.s0[c.sx] * c.a0, q = t.s0[c.sx + 1] * c.a1; r0 = p + q;
        p = t.s1[c.sx] * c.a0; q = t.s1[c.sx + 1] * c.a1; r1 = p + q;
    } else {
        r0 = t.s0[c.sx] * 1.0f; r1 = t.s1[c.sx] * 1.0f;
    }
    float p = r0 * t.b0, q = r1 * t.b1;
    return p + q;
}

// up-sample the three layers to full resolution and apply the inverse colour transform: planes -> rgb [B][H][W][3].
// Thread = one column, walking kUpRows rows (the column taps are computed once); lanes along x so plane reads are coalesced;
// grid = (x groups, row groups, images): no index divisions.
constexpr int kUpRows = 16;

template <int SPACE>
__global__ __launch_bounds__(256) void k_upsample_color(Geom g, UpScale us, const float *__restrict__ planes, float *__restrict__ rgb)
{
    const int b = blockIdx.z;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= g.W) return;
    const float *pl = planes + (long long)b * g.pstride;
    bool same[3];
    ColTaps ct[3];
#pragma unroll
    for (int l = 0; l < 3; l++) {
        same[l] = g.h[l] == g.H && g.w[l] == g.W;
        if (!same[l]) ct[l] = col_taps(g.w[l], us.sx[l], x);
    }
    const int y_end = min(g.H, (int)(blockIdx.y + 1) * kUpRows);
    for (int y = blockIdx.y * kUpRows; y < y_end; y++) {
        float c[3];
#pragma unroll
        for (int l = 0; l < 3; l++) {
            if (same[l]) c[l] = pl[g.poff[l] + (long long)y * g.w[l] + x];
            else c[l] = bilinear_at(row_taps(pl + g.poff[l], g.h[l], g.w[l], us.sy[l], y), ct[l]);
        }
        float r, gg, bb;
        color_inv_px<SPACE>(c[0], c[1], c[2], r, gg, bb);
        float *out = rgb + (((long long)b * g.H + y) * g.W + x) * 3;
        out[0] = r; out[1] = gg; out[2] = bb;
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
void launch_work_from_tables(hipStream_t st, const Geom &g, const QtGeom &q, const int *leaves, const long long *counts, LeafWork *const *work,
                             int *work_count, int *bad)
{
    WorkPtrsD wp;
    for (int k = 0; k < kMaxSizes; k++) wp.w[k] = work[k];
    long long mx = 1;
    for (int l = 0; l < 3; l++) if (q.leaf_cap[l] > mx) mx = q.leaf_cap[l];
    int bx = (int)((mx + 255) / 256);
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(k_work_from_tables, dim3(bx, 3, g.B), dim3(256), 0, st, g, q, leaves, counts, wp, work_count, ilog2(q.bmin), bad);
}

template <int S>
static void launch_idct_mfma_t(hipStream_t st, const Geom &g, const QtGeom &q, const IdctArgs &a, int blocks)
{
    size_t lds = (size_t)2 * S * S * sizeof(float) + sizeof(LayerTabD) + 8 + (size_t)(a.nplanes + 1) * sizeof(int);
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_idct_mfma<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    hipLaunchKernelGGL(k_idct_mfma<S>, dim3(blocks), dim3(IdctCfg<S>::NTHREADS), lds, st, g, q, a);
}

int launch_idct(hipStream_t st, int size, const Geom &g, const QtGeom &q, const IdctArgs &a, long long max_items)      // 0 = launched (or nothing to do)
{
    if (a.nplanes > kMaxPlanes) return -1;         // the per-plane prefix table would not fit the launch's LDS
    if (max_items <= 0) return 0;
    const size_t pref = (size_t)(a.nplanes + 1) * sizeof(int);
    auto cap = [&](long long per_block, int hi) {
        long long b = (max_items + per_block - 1) / per_block;
        return (int)(b < 1 ? 1 : b > hi ? hi : b);
    };
    switch (size) {
    case 2: hipLaunchKernelGGL(k_idct_small<2>, dim3(cap(128, 2048)), dim3(256), pref, st, g, q, a); break;
    case 4: hipLaunchKernelGGL(k_idct_small<4>, dim3(cap(64, 4096)), dim3(256), pref, st, g, q, a); break;
    case 8: hipLaunchKernelGGL(k_idct_small<8>, dim3(cap(32, 4096)), dim3(256), pref, st, g, q, a); break;
    case 16: hipLaunchKernelGGL(k_idct_small<16>, dim3(cap(16, 4096)), dim3(256), pref, st, g, q, a); break;
    case 32: launch_idct_mfma_t<32>(st, g, q, a, cap(1, 4096)); break;
    case 64: launch_idct_mfma_t<64>(st, g, q, a, cap(1, 768)); break;
    case 128: launch_idct_mfma_t<128>(st, g, q, a, cap(1, 256)); break;
    case 256:
        if (!a.scratch) return -1;
        hipLaunchKernelGGL(k_idct_big<256>, dim3(cap(1, big_blocks(256))), dim3(256), pref, st, g, q, a);
        break;
    case 512:
        if (!a.scratch) return -1;
        hipLaunchKernelGGL(k_idct_big<512>, dim3(cap(1, big_blocks(512))), dim3(256), pref, st, g, q, a);
        break;
    case 1024:
        if (!a.scratch) return -1;
        hipLaunchKernelGGL(k_idct_big<1024>, dim3(cap(1, big_blocks(1024))), dim3(256), pref, st, g, q, a);
        break;
    default: return -1;
    }
    return 0;
}

template <int SPACE>
static void launch_inv_t(hipStream_t st, const float *in, float *out, long long n)
{
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_color_inverse<SPACE>, dim3(blocks), dim3(256), 0, st, in, out, n);
}

int launch_color_inverse(hipStream_t st, int space, const float *in, float *out, long long n)
{
    switch (space) {
    case 0: launch_inv_t<0>(st, in, out, n); break;
    case 1: launch_inv_t<1>(st, in, out, n); break;
    case 2: launch_inv_t<2>(st, in, out, n); break;
    case 3: launch_inv_t<3>(st, in, out, n); break;
    case 4: launch_inv_t<4>(st, in, out, n); break;
    case 5: launch_inv_t<5>(st, in, out, n); break;
    case 6: launch_inv_t<6>(st, in, out, n); break;
    case 7: launch_inv_t<7>(st, in, out, n); break;
    default: return -1;
    }
    return 0;
}

template <int SPACE>
static void launch_up_t(hipStream_t st, const Geom &g, const float *planes, float *rgb)
{
    UpScale us;
    for (int l = 0; l < 3; l++) {
        us.sx[l] = 1.0 / ((double)g.W / (double)g.w[l]);
        us.sy[l] = 1.0 / ((double)g.H / (double)g.h[l]);
    }
    hipLaunchKernelGGL(k_upsample_color<SPACE>, dim3((g.W + 255) / 256, (g.H + kUpRows - 1) / kUpRows, g.B), dim3(256), 0, st, g, us, planes, rgb);
}

int launch_upsample_color(hipStream_t st, int space, const Geom &g, const float *planes, float *rgb)
{
    switch (space) {
    case 0: launch_up_t<0>(st, g, planes, rgb); break;
    case 1: launch_up_t<1>(st, g, planes, rgb); break;
    case 2: launch_up_t<2>(st, g, planes, rgb); break;
    case 3: launch_up_t<3>(st, g, planes, rgb); break;
    case 4: launch_up_t<4>(st, g, planes, rgb); break;
    case 5: launch_up_t<5>(st, g, planes, rgb); break;
    case 6: launch_up_t<6>(st, g, planes, rgb); break;
    default: return -1;
    }
    return 0;
}

}  // namespace aej
