// canny.hip -- the EdgeDetection.canny pipeline (src/jpeg/edge_detection.py:70-86) as LDS-halo-staged
// stencil kernels for gfx950:  CLAHE LUT build -> [CLAHE apply + Gaussian 3x3 + bilateral d=5 + histogram]
// -> percentile thresholds -> [Sobel + magnitude + NMS] -> tiled hysteresis to a fix-point.
//
// All arithmetic is integer, or float32 in a fixed order (no contraction), so every stage is bit-identical
// to the CPU oracle.  OpenCV semantics restated per stage are documented in DESIGN.md ("Canny chain").
#include "aej_common.h"
#include "aej_launch.h"

namespace aej {

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * (n - 1) - i;
    }
    return i;
}

__device__ __forceinline__ int cdiv(int a, int b) { return (a + b - 1) / b; }

// tile index t (over all layers of one image) -> layer, tile coordinates
__device__ __forceinline__ bool locate_tile(const Geom &g, int TW, int TH, int t, int &layer, int &tx, int &ty, int &ntx, int &nty, int &tbase)
{
    tbase = 0;
    for (int l = 0; l < g.nl; l++) {
        ntx = cdiv(g.w[l], TW); nty = cdiv(g.h[l], TH);
        int n = ntx * nty;
        if (t < n) { layer = l; ty = t / ntx; tx = t - ty * ntx; return true; }
        t -= n; tbase += n;
    }
    return false;
}

static long long tiles_per_image(const Geom &g, int TW, int TH)
{
    long long n = 0;
    for (int l = 0; l < g.nl; l++) n += (long long)((g.w[l] + TW - 1) / TW) * ((g.h[l] + TH - 1) / TH);
    return n;
}
long long hyst_tiles_per_image(const Geom &g) { return tiles_per_image(g, kHystTile, kHystTile); }

// ------------------------------------------------------------------------------------------------
// CLAHE: histogram contribution of the REFLECT_101 padding (only when h%4 or w%4 != 0), clahe.cpp
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_clahe_pad_hist(Geom g, const unsigned char *__restrict__ u8, int *__restrict__ tile_hist)
{
    const int l = blockIdx.y, b = blockIdx.z;
    const int w = g.w[l], h = g.h[l];
    if ((w % 4) == 0 && (h % 4) == 0) return;
    const int wp = g.ctw[l] * 4, hp = g.cth[l] * 4;
    const int nright = (wp - w) * hp;          // x in [w,wp), y in [0,hp)
    const int nbottom = w * (hp - h);          // x in [0,w),  y in [h,hp)
    const unsigned char *src = u8 + (long long)b * g.pstride + g.poff[l];
    int *hist = tile_hist + ((long long)b * 3 + l) * 16 * 256;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nright + nbottom; i += gridDim.x * 256) {
        int x, y;
        if (i < nright) { y = i / (wp - w); x = w + (i - y * (wp - w)); }
        else { int k = i - nright; y = h + k / w; x = k % w; }
        int v = src[(long long)reflect101(y, h) * w + reflect101(x, w)];
        atomicAdd(&hist[((y / g.cth[l]) * 4 + (x / g.ctw[l])) * 256 + v], 1);
    }
}

// ------------------------------------------------------------------------------------------------
// CLAHE LUT per tile: clip, redistribute, cumulative sum, scale (CLAHE_CalcLut_Body)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_clahe_lut(Geom g, const int *__restrict__ tile_hist, unsigned char *__restrict__ lut)
{
    __shared__ int s[256];
    __shared__ int s_red[4];
    const int tile = blockIdx.x, l = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const long long base = (((long long)b * 3 + l) * 16 + tile) * 256;
    const int area = g.ctw[l] * g.cth[l];
    const float lutScale = 255.0f / (float)area;
    int clip = (int)(0.75 * (double)area / 256.0);
    if (clip < 1) clip = 1;
    int hv = tile_hist[base + tid];
    int excess = hv > clip ? hv - clip : 0;
    if (hv > clip) hv = clip;
    // block sum of excess
    int v = excess;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    __syncthreads();
    const int clipped = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    const int batch = clipped / 256;
    int resid = clipped - batch * 256;
    hv += batch;
    if (resid != 0) {
        int step = 256 / resid;
        if (step < 1) step = 1;
        if (tid % step == 0 && tid / step < resid) hv++;
    }
    // inclusive scan over 256 bins
    s[tid] = hv;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        int t = tid >= o ? s[tid - o] : 0;
        __syncthreads();
        s[tid] += t;
        __syncthreads();
    }
    float f = (float)s[tid] * lutScale;
    int r = __float2int_rn(f);
    lut[base + tid] = (unsigned char)(r < 0 ? 0 : r > 255 ? 255 : r);
}

// ------------------------------------------------------------------------------------------------
// fused CLAHE apply (a-4) + Gaussian 3x3 (a-5) + bilateral d=5 (a-6) + histogram of the result (a-7)
// Output tile 64 x 32; LDS holds the CLAHE image with a 3-pixel halo and the Gaussian image with a
// 2-pixel halo, indexed by image coordinate so that REFLECT_101 is a coordinate remap.
// ------------------------------------------------------------------------------------------------
constexpr int kAW = kBlurTW + 8;   // LDS row stride (bytes)
constexpr int kAH = kBlurTH + 6;
constexpr int kBH = kBlurTH + 4;

__constant__ int c_bil_dy[13] = { -2, -1, -1, -1, 0, 0, 0, 0, 0, 1, 1, 1, 2 };
__constant__ int c_bil_dx[13] = { 0, -1, 0, 1, -2, -1, 0, 1, 2, -1, 0, 1, 0 };

__global__ __launch_bounds__(256) void k_clahe_blur(Geom g, CannyBuffers cb)
{
    __shared__ __attribute__((aligned(16))) unsigned char sLut[16 * 256];
    __shared__ unsigned char sA[kAH * kAW];
    __shared__ unsigned char sB[kBH * kAW];
    __shared__ float sCw[256];
    __shared__ float sSw[16];
    __shared__ int sHist[256];

    const int tid = threadIdx.x, b = blockIdx.y;
    int l, tx, ty, ntx, nty, tbase;
    if (!locate_tile(g, kBlurTW, kBlurTH, blockIdx.x, l, tx, ty, ntx, nty, tbase)) return;
    const int w = g.w[l], h = g.h[l];
    const int x0 = tx * kBlurTW, y0 = ty * kBlurTH;
    const long long pbase = (long long)b * g.pstride + g.poff[l];
    const unsigned char *src = cb.u8a + pbase;

    reinterpret_cast<uint4 *>(sLut)[tid] = reinterpret_cast<const uint4 *>(cb.lut + ((long long)b * 3 + l) * 4096)[tid];
    sCw[tid] = cb.color_w[tid];
    if (tid < 13) sSw[tid] = cb.space_w[tid];
    sHist[tid] = 0;
    __syncthreads();

    // ---- stage A: CLAHE interpolation (CLAHE_Interpolation_Body) on [x0-3, x0+TW+3) x [y0-3, y0+TH+3)
    const float inv_tw = 1.0f / (float)g.ctw[l], inv_th = 1.0f / (float)g.cth[l];
    for (int idx = tid; idx < kAH * (kBlurTW + 6); idx += 256) {
        int j = idx / (kBlurTW + 6), i = idx - j * (kBlurTW + 6);
        int gx = x0 - 3 + i, gy = y0 - 3 + j;
        if (gx >= 0 && gx < w && gy >= 0 && gy < h) {
            int v = src[(long long)gy * w + gx];
            float tyf = (float)gy * inv_th - 0.5f;
            int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
            float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
            if (ty1 < 0) ty1 = 0;
            if (ty2 > 3) ty2 = 3;
            float txf = (float)gx * inv_tw - 0.5f;
            int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
            float xa = txf - (float)tx1, xa1 = 1.0f - xa;
            if (tx1 < 0) tx1 = 0;
            if (tx2 > 3) tx2 = 3;
            float pa = (float)sLut[(ty1 * 4 + tx1) * 256 + v] * xa1;
            float pb = (float)sLut[(ty1 * 4 + tx2) * 256 + v] * xa;
            float pc = (float)sLut[(ty2 * 4 + tx1) * 256 + v] * xa1;
            float pd = (float)sLut[(ty2 * 4 + tx2) * 256 + v] * xa;
            float top = pa + pb, bot = pc + pd;
            float t1 = top * ya1, t2 = bot * ya;
            float res = t1 + t2;
            int r = __float2int_rn(res);
            unsigned char o = (unsigned char)(r < 0 ? 0 : r > 255 ? 255 : r);
            sA[j * kAW + i] = o;
            if (cb.dump_clahe && i >= 3 && i < kBlurTW + 3 && j >= 3 && j < kBlurTH + 3) cb.dump_clahe[pbase + (long long)gy * w + gx] = o;
        }
    }
    __syncthreads();

    // ---- stage B: Gaussian [1 2 1]^2, (sum + 8) >> 4, REFLECT_101, on [x0-2, x0+TW+2) x [y0-2, y0+TH+2)
    for (int idx = tid; idx < kBH * (kBlurTW + 4); idx += 256) {
        int j = idx / (kBlurTW + 4), i = idx - j * (kBlurTW + 4);
        int gx = x0 - 2 + i, gy = y0 - 2 + j;
        if (gx >= 0 && gx < w && gy >= 0 && gy < h) {
            int xm = reflect101(gx - 1, w) - (x0 - 3), xc = gx - (x0 - 3), xp = reflect101(gx + 1, w) - (x0 - 3);
            int ym = reflect101(gy - 1, h) - (y0 - 3), yc = gy - (y0 - 3), yp = reflect101(gy + 1, h) - (y0 - 3);
            const unsigned char *r0 = sA + ym * kAW, *r1 = sA + yc * kAW, *r2 = sA + yp * kAW;
            int s = (r0[xm] + 2 * r0[xc] + r0[xp]) + 2 * (r1[xm] + 2 * r1[xc] + r1[xp]) + (r2[xm] + 2 * r2[xc] + r2[xp]);
            unsigned char o = (unsigned char)((s + 8) >> 4);
            sB[j * kAW + i] = o;
            if (cb.dump_gauss && i >= 2 && i < kBlurTW + 2 && j >= 2 && j < kBlurTH + 2) cb.dump_gauss[pbase + (long long)gy * w + gx] = o;
        }
    }
    __syncthreads();

    // ---- stage C: bilateral, 13 taps in row-major order, w = sw*cw, wsum += w, sum = fma(v, w, sum)
    unsigned char *dst = cb.u8b + pbase;
    for (int idx = tid; idx < kBlurTW * kBlurTH; idx += 256) {
        int j = idx / kBlurTW, i = idx - j * kBlurTW;
        int gx = x0 + i, gy = y0 + j;
        if (gx < w && gy < h) {
            int v0 = sB[(j + 2) * kAW + (i + 2)];
            float sum = 0.f, wsum = 0.f;
#pragma unroll
            for (int k = 0; k < 13; k++) {
                int yy = reflect101(gy + c_bil_dy[k], h) - (y0 - 2);
                int xx = reflect101(gx + c_bil_dx[k], w) - (x0 - 2);
                int v = sB[yy * kAW + xx];
                int d = v - v0;
                d = d < 0 ? -d : d;
                float wgt = sSw[k] * sCw[d];
                wsum = wsum + wgt;
                sum = __builtin_fmaf((float)v, wgt, sum);
            }
            int r = __float2int_rn(sum / wsum);
            unsigned char o = (unsigned char)(r < 0 ? 0 : r > 255 ? 255 : r);
            dst[(long long)gy * w + gx] = o;
            atomicAdd(&sHist[o], 1);
        }
    }
    __syncthreads();
    int c = sHist[tid];
    if (c) atomicAdd(&cb.blur_hist[((long long)b * 3 + l) * 256 + tid], c);
}

// ------------------------------------------------------------------------------------------------
// a-7: np.percentile(blur, 10), (blur, 30) from the 256-bin histogram, then the Canny integer thresholds
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double pct_from(const int *cum /* inclusive */, const int *val_at /*scratch*/, long long n, double q, int tid,
                                           int *s_a, int *s_b)
{
    (void)val_at;
    double virt = (double)(n - 1) * (q / 100.0);
    long long lo = (long long)floor(virt);
    long long hi = lo + 1;
    if (hi > n - 1) hi = n - 1;
    int prev = tid == 0 ? 0 : cum[tid - 1];
    int cur = cum[tid];
    if ((long long)cur > lo && (long long)prev <= lo) *s_a = tid;
    if ((long long)cur > hi && (long long)prev <= hi) *s_b = tid;
    __syncthreads();
    double t = virt - (double)lo;
    double a = (double)*s_a, bb = (double)*s_b;
    double diff = bb - a;
    double r = a + diff * t;
    if (t >= 0.5) r = bb - diff * (1.0 - t);
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void k_thresholds(Geom g, const int *__restrict__ blur_hist, int *__restrict__ thr)
{
    __shared__ int cum[256];
    __shared__ int s_a, s_b;
    const int l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    cum[tid] = blur_hist[((long long)b * 3 + l) * 256 + tid];
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        int t = tid >= o ? cum[tid - o] : 0;
        __syncthreads();
        cum[tid] += t;
        __syncthreads();
    }
    long long n = (long long)g.w[l] * g.h[l];
    double lo = pct_from(cum, nullptr, n, 0.10 * 100, tid, &s_a, &s_b);
    double hi = pct_from(cum, nullptr, n, 0.30 * 100, tid, &s_a, &s_b);
    if (tid == 0) {
        // cv::Canny, L2gradient=true (canny.cpp)
        if (lo > hi) { double t = lo; lo = hi; hi = t; }
        if (lo > 32767.0) lo = 32767.0;
        if (hi > 32767.0) hi = 32767.0;
        if (lo > 0) lo *= lo;
        if (hi > 0) hi *= hi;
        thr[((long long)b * 3 + l) * 2 + 0] = (int)floor(lo);
        thr[((long long)b * 3 + l) * 2 + 1] = (int)floor(hi);
    }
}

// ------------------------------------------------------------------------------------------------
// a-8 part 1: Sobel 3x3 (BORDER_REPLICATE), magnitude dx^2+dy^2, non-maximum suppression.
// map: 1 = suppressed, 0 = weak candidate, 2 = strong (OpenCV's encoding).
// ------------------------------------------------------------------------------------------------
constexpr int kSW = kBlurTW + 8;        // u8 LDS stride
constexpr int kMW = kBlurTW + 2 + 1;    // magnitude LDS stride (ints), +1 to skew banks

__global__ __launch_bounds__(256) void k_sobel_nms(Geom g, CannyBuffers cb)
{
    __shared__ unsigned char sU[(kBlurTH + 4) * kSW];
    __shared__ int sM[(kBlurTH + 2) * kMW];
    const int tid = threadIdx.x, b = blockIdx.y;
    int l, tx, ty, ntx, nty, tbase;
    if (!locate_tile(g, kBlurTW, kBlurTH, blockIdx.x, l, tx, ty, ntx, nty, tbase)) return;
    const int w = g.w[l], h = g.h[l];
    const int x0 = tx * kBlurTW, y0 = ty * kBlurTH;
    const long long pbase = (long long)b * g.pstride + g.poff[l];
    const unsigned char *src = cb.u8b + pbase;
    const int low = cb.thr[((long long)b * 3 + l) * 2], high = cb.thr[((long long)b * 3 + l) * 2 + 1];

    for (int idx = tid; idx < (kBlurTH + 4) * (kBlurTW + 4); idx += 256) {
        int j = idx / (kBlurTW + 4), i = idx - j * (kBlurTW + 4);
        int gx = x0 - 2 + i, gy = y0 - 2 + j;
        gx = gx < 0 ? 0 : gx >= w ? w - 1 : gx;
        gy = gy < 0 ? 0 : gy >= h ? h - 1 : gy;
        sU[j * kSW + i] = src[(long long)gy * w + gx];
    }
    __syncthreads();
    for (int idx = tid; idx < (kBlurTH + 2) * (kBlurTW + 2); idx += 256) {
        int j = idx / (kBlurTW + 2), i = idx - j * (kBlurTW + 2);
        int gx = x0 - 1 + i, gy = y0 - 1 + j;
        int m = 0;
        if (gx >= 0 && gx < w && gy >= 0 && gy < h) {
            const unsigned char *r0 = sU + j * kSW + i, *r1 = r0 + kSW, *r2 = r1 + kSW;   // centre at (j+1, i+1)
            int dx = (r0[2] + 2 * r1[2] + r2[2]) - (r0[0] + 2 * r1[0] + r2[0]);
            int dy = (r2[0] + 2 * r2[1] + r2[2]) - (r0[0] + 2 * r0[1] + r0[2]);
            m = dx * dx + dy * dy;
        }
        sM[j * kMW + i] = m;
    }
    __syncthreads();
    unsigned char *map = cb.u8a + pbase;
    for (int idx = tid; idx < kBlurTW * kBlurTH; idx += 256) {
        int j = idx / kBlurTW, i = idx - j * kBlurTW;
        int gx = x0 + i, gy = y0 + j;
        if (gx < w && gy < h) {
            const int *ma = sM + (j + 1) * kMW + (i + 1), *mp = ma - kMW, *mn = ma + kMW;
            int m = *ma;
            unsigned char res = 1;
            if (m > low) {
                const unsigned char *r0 = sU + (j + 1) * kSW + (i + 1), *r1 = r0 + kSW, *r2 = r1 + kSW;
                int xs = (r0[2] + 2 * r1[2] + r2[2]) - (r0[0] + 2 * r1[0] + r2[0]);
                int ys = (r2[0] + 2 * r2[1] + r2[2]) - (r0[0] + 2 * r0[1] + r0[2]);
                int ax = xs < 0 ? -xs : xs, ay = (ys < 0 ? -ys : ys) << 15;
                int tg22x = ax * 13573;
                bool keep;
                if (ay < tg22x) keep = (m > ma[-1] && m >= ma[1]);
                else {
                    int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x) keep = (m > mp[0] && m >= mn[0]);
                    else {
                        int s = ((xs ^ ys) < 0) ? -1 : 1;
                        keep = (m > mp[-s] && m > mn[s]);
                    }
                }
                if (keep) res = (m > high) ? 2 : 0;
            }
            map[(long long)gy * w + gx] = res;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// a-8 part 2: hysteresis.  Each pass brings every dirty 64x64 tile to its local fix-point in LDS (given
// the current 1-pixel halo) and marks the 8 neighbours dirty when its border ring changed.  The map only
// ever moves 0 -> 2, so the global fix-point is unique and equals OpenCV's stack-based flood fill.
// ------------------------------------------------------------------------------------------------
constexpr int kHS = kHystTile + 2 + 2;   // LDS stride 68

__global__ __launch_bounds__(256) void k_hyst_pass(Geom g, unsigned char *__restrict__ mapbuf, unsigned char *__restrict__ dirty_cur,
                                                   unsigned char *__restrict__ dirty_nxt, int *__restrict__ pass_changed,
                                                   long long tiles_per_img)
{
    __shared__ unsigned char s[(kHystTile + 2) * kHS];
    __shared__ int s_flag;
    const int tid = threadIdx.x, b = blockIdx.y, t = blockIdx.x;
    unsigned char *dc = dirty_cur + (long long)b * tiles_per_img + t;
    if (*dc == 0) return;          // block-uniform
    int l, tx, ty, ntx, nty, tbase;
    if (!locate_tile(g, kHystTile, kHystTile, t, l, tx, ty, ntx, nty, tbase)) return;
    const int w = g.w[l], h = g.h[l];
    const int x0 = tx * kHystTile, y0 = ty * kHystTile;
    unsigned char *map = mapbuf + (long long)b * g.pstride + g.poff[l];
    if (tid == 0) { *dc = 0; s_flag = 0; }

    for (int idx = tid; idx < (kHystTile + 2) * (kHystTile + 2); idx += 256) {
        int j = idx / (kHystTile + 2), i = idx - j * (kHystTile + 2);
        int gx = x0 - 1 + i, gy = y0 - 1 + j;
        unsigned char v = 1;
        if (gx >= 0 && gx < w && gy >= 0 && gy < h) v = map[(long long)gy * w + gx];
        s[j * kHS + i] = v;
    }
    __syncthreads();
    // each thread owns a 4x4 patch of the 64x64 interior
    const int pxo = (tid & 15) * 4 + 1, pyo = (tid >> 4) * 4 + 1;
    bool any_change = false, border_change = false;
    for (;;) {
        bool changed = false;
#pragma unroll
        for (int dy = 0; dy < 4; dy++)
#pragma unroll
            for (int dx = 0; dx < 4; dx++) {
                unsigned char *p = s + (pyo + dy) * kHS + (pxo + dx);
                if (*p == 0) {
                    bool n2 = p[-kHS - 1] == 2 || p[-kHS] == 2 || p[-kHS + 1] == 2 || p[-1] == 2 || p[1] == 2 ||
                              p[kHS - 1] == 2 || p[kHS] == 2 || p[kHS + 1] == 2;
                    if (n2) {
                        *p = 2;
                        changed = true;
                        int yy = pyo + dy, xx = pxo + dx;
                        if (yy == 1 || yy == kHystTile || xx == 1 || xx == kHystTile) border_change = true;
                    }
                }
            }
        any_change |= changed;
        if (!__syncthreads_or(changed ? 1 : 0)) break;
    }
    if (any_change) {
#pragma unroll
        for (int dy = 0; dy < 4; dy++) {
            int gy = y0 + pyo - 1 + dy;
#pragma unroll
            for (int dx = 0; dx < 4; dx++) {
                int gx = x0 + pxo - 1 + dx;
                if (gx < w && gy < h) {
                    unsigned char v = s[(pyo + dy) * kHS + (pxo + dx)];
                    if (v == 2) map[(long long)gy * w + gx] = 2;
                }
            }
        }
    }
    if (border_change) atomicOr(&s_flag, 1);
    __syncthreads();
    if (s_flag && tid < 8) {
        const int ox[8] = { -1, 0, 1, -1, 1, -1, 0, 1 }, oy[8] = { -1, -1, -1, 0, 0, 1, 1, 1 };
        int nx = tx + ox[tid], ny = ty + oy[tid];
        if (nx >= 0 && nx < ntx && ny >= 0 && ny < nty) dirty_nxt[(long long)b * tiles_per_img + tbase + ny * ntx + nx] = 1;
        if (tid == 0) atomicAdd(pass_changed, 1);
    }
}

__global__ __launch_bounds__(256) void k_edge_final(const unsigned char *__restrict__ map, unsigned char *__restrict__ edge, long long n)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) edge[i] = map[i] == 2 ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
void launch_clahe_pad_hist(hipStream_t st, const Geom &g, const CannyBuffers &cb)
{
    bool need = false;
    int maxpad = 0;
    for (int l = 0; l < g.nl; l++)
        if ((g.w[l] % 4) || (g.h[l] % 4)) {
            need = true;
            int wp = g.ctw[l] * 4, hp = g.cth[l] * 4;
            int n = (wp - g.w[l]) * hp + g.w[l] * (hp - g.h[l]);
            if (n > maxpad) maxpad = n;
        }
    if (!need) return;
    int bx = (maxpad + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(k_clahe_pad_hist, dim3(bx, g.nl, g.B), dim3(256), 0, st, g, cb.u8a, cb.tile_hist);
}

void launch_clahe_lut(hipStream_t st, const Geom &g, const CannyBuffers &cb)
{
    hipLaunchKernelGGL(k_clahe_lut, dim3(16, g.nl, g.B), dim3(256), 0, st, g, cb.tile_hist, cb.lut);
}

void launch_clahe_blur(hipStream_t st, const Geom &g, const CannyBuffers &cb)
{
    long long t = tiles_per_image(g, kBlurTW, kBlurTH);
    hipLaunchKernelGGL(k_clahe_blur, dim3((unsigned)t, g.B), dim3(256), 0, st, g, cb);
}

void launch_thresholds(hipStream_t st, const Geom &g, const CannyBuffers &cb)
{
    hipLaunchKernelGGL(k_thresholds, dim3(g.nl, g.B), dim3(256), 0, st, g, cb.blur_hist, cb.thr);
}

void launch_sobel_nms(hipStream_t st, const Geom &g, const CannyBuffers &cb)
{
    long long t = tiles_per_image(g, kBlurTW, kBlurTH);
    hipLaunchKernelGGL(k_sobel_nms, dim3((unsigned)t, g.B), dim3(256), 0, st, g, cb);
}

void launch_hyst_pass(hipStream_t st, const Geom &g, const CannyBuffers &cb, int pass)
{
    long long t = hyst_tiles_per_image(g);
    unsigned char *cur = cb.dirty + (long long)(pass & 1) * g.B * t;
    unsigned char *nxt = cb.dirty + (long long)((pass + 1) & 1) * g.B * t;
    hipLaunchKernelGGL(k_hyst_pass, dim3((unsigned)t, g.B), dim3(256), 0, st, g, cb.u8a, cur, nxt, cb.pass_changed + pass, t);
}

void launch_edge_final(hipStream_t st, const Geom &g, const unsigned char *map, unsigned char *edge01)
{
    long long n = (long long)g.B * g.pstride;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_edge_final, dim3(blocks), dim3(256), 0, st, map, edge01, n);
}

}  // namespace aej
