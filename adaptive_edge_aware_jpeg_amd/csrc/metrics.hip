// metrics.hip -- EvaluationMetrics (src/image/evaluation_metrics.py:50-89) for batches of image pairs on the GPU:
//   psnr()    = piq.psnr(x, y, data_range=1.0)                       -10 log10(mean((x - y)^2) + 1e-8), all channels
//   ssim()    = piq.ssim(grey(x), grey(y), data_range=255.0)         8-bit grey (Image.get_uint8 + cv.cvtColor RGB2GRAY),
//                                                                    average-pooled by round(min(H, W) / 256), 11x11 Gaussian
//   ms_ssim() = piq.multi_scale_ssim(x, y, data_range=1.0)           5 scales, 2x2 average pooling between them
// piq 0.8.0 is not under /root/reference (requirements.txt:22); this file restates its published algorithm (piq/psnr.py,
// piq/ssim.py, piq/ms_ssim.py, piq/functional/filters.py); tests/test_metrics.py compares it with a numpy restatement.  These are float32 reductions whose summation order torch does not define, so parity here is to a stated
// tolerance, not bitwise.  The window is the normalised Gaussian exp(-(i^2 + j^2) / (2 sigma^2)) / sum, applied as two
// separable passes through LDS ('valid' convolution: no padding).
#include "aej_common.h"
#include "aej_launch.h"
#include <stdlib.h>
#include <type_traits>

namespace aej {

constexpr int kSsimK = 11;               // window size

__device__ __forceinline__ double block_sum(double v, double *s_red)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    const int tid = threadIdx.x;
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

__device__ __forceinline__ unsigned char to_u8(float v)      // Image.get_uint8 (image.py:120-127): (data * 255).astype(np.uint8)
{
    return (unsigned char)((int)(v * 255.0f) & 0xFF);
}

__device__ __forceinline__ unsigned char grey_u8(float r, float g, float b)
{
    // cv.cvtColor(COLOR_RGB2GRAY) on uint8: fixed point, 14 fractional bits (OpenCV color_rgb.simd.hpp RGB2Gray<uchar>)
    const int R = to_u8(r), G = to_u8(g), B = to_u8(b);
    return (unsigned char)((R * 4899 + G * 9617 + B * 1868 + (1 << 13)) >> 14);
}

// ---- pass over both images: squared error (psnr) and the two 8-bit grey planes (ssim) ----
__global__ __launch_bounds__(256) void k_metric_prep(const float *__restrict__ a, const float *__restrict__ b, long long npx, double *__restrict__ acc,
                                                     unsigned char *__restrict__ ga, unsigned char *__restrict__ gb)
{
    __shared__ double s_red[4];
    const int img = blockIdx.y;
    const float *pa = a + (long long)img * npx * 3, *pb = b + (long long)img * npx * 3;
    double sum = 0.0;
    const long long stride = (long long)gridDim.x * 256;
    if ((npx & 3) == 0) {           // 4 pixels = 3 float4 per image
        for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < npx / 4; q += stride) {
            float va[12], vb[12];
            const float4 *qa = reinterpret_cast<const float4 *>(pa) + q * 3, *qb = reinterpret_cast<const float4 *>(pb) + q * 3;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float4 t = qa[k], u = qb[k];
                va[4 * k] = t.x; va[4 * k + 1] = t.y; va[4 * k + 2] = t.z; va[4 * k + 3] = t.w;
                vb[4 * k] = u.x; vb[4 * k + 1] = u.y; vb[4 * k + 2] = u.z; vb[4 * k + 3] = u.w;
            }
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 12; k++) { float d = va[k] - vb[k]; s += d * d; }
            sum += (double)s;
            if (ga) {
                uchar4 oa, ob;
                oa.x = grey_u8(va[0], va[1], va[2]); oa.y = grey_u8(va[3], va[4], va[5]); oa.z = grey_u8(va[6], va[7], va[8]); oa.w = grey_u8(va[9], va[10], va[11]);
                ob.x = grey_u8(vb[0], vb[1], vb[2]); ob.y = grey_u8(vb[3], vb[4], vb[5]); ob.z = grey_u8(vb[6], vb[7], vb[8]); ob.w = grey_u8(vb[9], vb[10], vb[11]);
                reinterpret_cast<uchar4 *>(ga + (long long)img * npx)[q] = oa;
                reinterpret_cast<uchar4 *>(gb + (long long)img * npx)[q] = ob;
            }
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npx; i += stride) {
            float s = 0.f, va[3], vb[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { va[k] = pa[3 * i + k]; vb[k] = pb[3 * i + k]; float d = va[k] - vb[k]; s += d * d; }
            sum += (double)s;
            if (ga) {
                ga[(long long)img * npx + i] = grey_u8(va[0], va[1], va[2]);
                gb[(long long)img * npx + i] = grey_u8(vb[0], vb[1], vb[2]);
            }
        }
    }
    sum = block_sum(sum, s_red);
    if (threadIdx.x == 0) atomicAdd(&acc[(long long)img * kMetricSlots + 0], sum);
}

// ---- piq.ssim: x / data_range, then F.avg_pool2d(kernel_size = f) when f > 1 ----
__global__ __launch_bounds__(256) void k_metric_pool_grey(const unsigned char *__restrict__ ga, const unsigned char *__restrict__ gb, int H, int W, int f,
                                                          int hp, int wp, float *__restrict__ xa, float *__restrict__ xb)
{
    const int img = blockIdx.y;
    const long long n = (long long)hp * wp;
    const float inv = 1.0f / (float)(f * f);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int y = (int)(i / wp), x = (int)(i - (long long)y * wp);
        float sa = 0.f, sb = 0.f;
        for (int dy = 0; dy < f; dy++) {
            const long long row = (long long)img * H * W + (long long)(y * f + dy) * W + (long long)x * f;
            for (int dx = 0; dx < f; dx++) { sa += (float)ga[row + dx] / 255.0f; sb += (float)gb[row + dx] / 255.0f; }
        }
        xa[(long long)img * n + i] = sa * inv;
        xb[(long long)img * n + i] = sb * inv;
    }
}

// ---- one SSIM scale: 'valid' Gaussian means of x, y, x^2 + y^2, xy; the cs and ss maps; their sums ----
// (the contrast term needs sigma_x^2 + sigma_y^2 only as a sum, so x^2 and y^2 share one filtered map: four maps, not piq's five)
// INTERLEAVED: the images are [B][h][w][3] (the caller's float32 RGB); otherwise planar [B][C][h][w].
struct SsimArgs {
    const float *xa, *xb;
    int h, w, C;
    int strip_rows;           // output rows per strip
    int want_ss;              // the luminance term too (the grey SSIM and the last MS-SSIM scale; the others use cs alone)
    float g[kSsimK];          // normalised 1-D Gaussian: g[i] * g[j] is piq's 2-D window
    float c1, c2;
    double *acc;              // [B][kMetricSlots]
    int slot;                 // first slot: channel c adds ss to slot + 2c, cs to slot + 2c + 1
    float *pool_a, *pool_b;   // INTERLEAVED with even h, w: the next scale (2 x 2 means, planar [B][3][h / 2][w / 2]) written on the way, or null
};

// ---- one scale as a sliding window ----
// One WAVE owns a strip of 128 output columns x strip_rows output rows of ONE channel and walks down the input rows.  A lane owns two
// adjacent columns.  Per input row it loads its pixels of x and y (one row ahead: the latency hides under a row of arithmetic), forms
// x, y, x^2 + y^2, xy once per pixel and puts the four maps' row through a wave-private LDS row (no workgroup barrier anywhere); the
// 12-value window of its two columns comes back as six aligned 8-byte reads per map; the four horizontal sums of both columns go into a
// ring of the last 11 rows held in registers (addressed at compile time: the row loop is unrolled 11-fold), and the vertical sums
// complete one output row per input row.  Everything is explicit fmaf: this file is compiled with the library's -ffp-contract=off,
// and a Gaussian mean needs no particular rounding (until round 5 every g * v + s was two instructions).
// Planar scales: the four waves of a workgroup take four strips.  The interleaved RGB of scale 0: a workgroup is THREE waves on the same
// strip, one per channel, so that the row segments the three read (each uses a third of every line) are fetched from HBM once and
// served from the L1 / L2 the other two times -- as a channel loop inside one wave (round 4) every pass over the strip came from HBM
// again: 3 x 6.4 GB for 32 x 4K.  The same kernel writes the 2 x 2 means that are the next scale's input from the rows it holds anyway
// (each strip those of its own output columns / rows, the last strip of a row / column also its halo's): the separate pooling pass read both
// images a second time, 1.46 ms of 7.8 for 32 x 4K.
constexpr int kSsimCols = 128;                         // output columns per wave
constexpr int kSsimIn = kSsimCols + kSsimK - 1;        // input columns per wave: 138
constexpr int kSsimRow = kSsimIn + 2;                  // LDS row stride in floats (even: 8-byte aligned rows)

template <bool INTERLEAVED>
__global__ __launch_bounds__(256) void k_ssim_strip(SsimArgs A)
{
    __shared__ __attribute__((aligned(8))) float rows[4][4][kSsimRow];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int img = blockIdx.z;
    const int oh = A.h - (kSsimK - 1), ow = A.w - (kSsimK - 1);
    const int nsx = (ow + kSsimCols - 1) / kSsimCols, nsy = (oh + A.strip_rows - 1) / A.strip_rows;
    const int sidx = INTERLEAVED ? blockIdx.x : blockIdx.x * 4 + wave;
    if (sidx >= nsx * nsy) return;                       // no workgroup-level synchronisation below
    const int c = INTERLEAVED ? wave : blockIdx.y;
    const int sy_ = sidx / nsx, sx_ = sidx - sy_ * nsx;
    const int x0 = sx_ * kSsimCols, y0 = sy_ * A.strip_rows;
    const int rows_out = min(A.strip_rows, oh - y0), R = rows_out + kSsimK - 1;
    float (*wrow)[kSsimRow] = rows[wave];
    // the pixels this lane loads per row: columns x0 + 2 lane, + 1 and (lanes 0..4) x0 + 128 + 2 lane, + 1
    const int col[4] = { x0 + 2 * lane, x0 + 2 * lane + 1, x0 + kSsimCols + 2 * lane, x0 + kSsimCols + 2 * lane + 1 };
    bool ok[4];
    int off[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        ok[k] = col[k] < A.w && (k < 2 || lane < (kSsimK - 1) / 2);
        off[k] = ok[k] ? (INTERLEAVED ? 3 * col[k] + c : col[k]) : 0;
    }
    const long long row_stride = INTERLEAVED ? (long long)A.w * 3 : A.w;
    const long long first = INTERLEAVED ? ((long long)img * A.h + y0) * row_stride : (((long long)img * A.C + c) * A.h + y0) * row_stride;
    const float *qx = A.xa + first, *qy = A.xb + first;
    float px[4] = { 0.f, 0.f, 0.f, 0.f }, py[4] = { 0.f, 0.f, 0.f, 0.f };      // the NEXT input row
    auto fetch = [&](int r) {
        if (r >= R) return;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (ok[k]) { px[k] = qx[off[k]]; py[k] = qy[off[k]]; }
        qx += row_stride; qy += row_stride;
    };
    fetch(0);
    float h[kSsimK][4][2];
    float ss_acc = 0.f, cs_acc = 0.f;
    const bool out_ok[2] = { x0 + 2 * lane < ow, x0 + 2 * lane + 1 < ow };
    // 2 x 2 means for the next scale: row pairs (y0 + r even, odd), column pairs = this lane's; the halo only where no other strip owns it
    const bool pool = INTERLEAVED && A.pool_a != nullptr;
    const bool pool_halo_x = sx_ == nsx - 1, pool_halo_y = sy_ == nsy - 1;
    const int w2 = A.w >> 1;
    float keep[4] = { 0.f, 0.f, 0.f, 0.f };
    auto step = [&](auto ph, int r) {
        constexpr int PH = decltype(ph)::value;
        if (r >= R) return;
        if (INTERLEAVED && pool && (r < rows_out || pool_halo_y)) {
            const float hs[4] = { px[0] + px[1], py[0] + py[1], px[2] + px[3], py[2] + py[3] };
            if (r & 1) {
                const long long o = (((long long)img * 3 + c) * (A.h >> 1) + ((y0 + r) >> 1)) * w2;
                if (ok[0]) { A.pool_a[o + (col[0] >> 1)] = (keep[0] + hs[0]) * 0.25f; A.pool_b[o + (col[0] >> 1)] = (keep[1] + hs[1]) * 0.25f; }
                if (ok[2] && pool_halo_x) { A.pool_a[o + (col[2] >> 1)] = (keep[2] + hs[2]) * 0.25f; A.pool_b[o + (col[2] >> 1)] = (keep[3] + hs[3]) * 0.25f; }
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) keep[k] = hs[k];
            }
        }
        {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const float s0 = fmaf(px[0], px[0], py[0] * py[0]), s1 = fmaf(px[1], px[1], py[1] * py[1]);
            *reinterpret_cast<f32x2 *>(&wrow[0][2 * lane]) = f32x2{ px[0], px[1] };
            *reinterpret_cast<f32x2 *>(&wrow[1][2 * lane]) = f32x2{ py[0], py[1] };
            *reinterpret_cast<f32x2 *>(&wrow[2][2 * lane]) = f32x2{ s0, s1 };
            *reinterpret_cast<f32x2 *>(&wrow[3][2 * lane]) = f32x2{ px[0] * py[0], px[1] * py[1] };
            if (lane < (kSsimK - 1) / 2) {
                const float s2 = fmaf(px[2], px[2], py[2] * py[2]), s3 = fmaf(px[3], px[3], py[3] * py[3]);
                *reinterpret_cast<f32x2 *>(&wrow[0][kSsimCols + 2 * lane]) = f32x2{ px[2], px[3] };
                *reinterpret_cast<f32x2 *>(&wrow[1][kSsimCols + 2 * lane]) = f32x2{ py[2], py[3] };
                *reinterpret_cast<f32x2 *>(&wrow[2][kSsimCols + 2 * lane]) = f32x2{ s2, s3 };
                *reinterpret_cast<f32x2 *>(&wrow[3][kSsimCols + 2 * lane]) = f32x2{ px[2] * py[2], px[3] * py[3] };
            }
        }
        fetch(r + 1);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < 4; m++) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            float v[kSsimK + 1];
#pragma unroll
            for (int t = 0; t < (kSsimK + 1) / 2; t++) {
                const f32x2 q = *reinterpret_cast<const f32x2 *>(&wrow[m][2 * lane + 2 * t]);
                v[2 * t] = q.x; v[2 * t + 1] = q.y;
            }
            float a0 = A.g[0] * v[0], a1 = A.g[0] * v[1];
#pragma unroll
            for (int t = 1; t < kSsimK; t++) { a0 = fmaf(A.g[t], v[t], a0); a1 = fmaf(A.g[t], v[t + 1], a1); }
            h[PH][m][0] = a0; h[PH][m][1] = a1;
        }
        __builtin_amdgcn_wave_barrier();
        if (r >= kSsimK - 1) {                  // rows r - 10 .. r are in the ring: the oldest sits right after PH
#pragma unroll
            for (int k = 0; k < 2; k++) {
                float v[4];
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    float sacc = A.g[0] * h[(PH + 1) % kSsimK][m][k];
#pragma unroll
                    for (int t = 1; t < kSsimK; t++) sacc = fmaf(A.g[t], h[(PH + 1 + t) % kSsimK][m][k], sacc);
                    v[m] = sacc;
                }
                if (out_ok[k]) {
                    // 1 / d by the hardware reciprocal (1 ulp): these are float32 means compared at 1e-4, and the IEEE division is ten instructions
                    const float mu_xx = v[0] * v[0], mu_yy = v[1] * v[1], mu_xy = v[0] * v[1];
                    const float s_sum = (v[2] - mu_xx) - mu_yy, s_xy = v[3] - mu_xy;
                    const float cs = fmaf(2.f, s_xy, A.c2) * __builtin_amdgcn_rcpf(s_sum + A.c2);
                    cs_acc += cs;
                    if (A.want_ss) ss_acc += fmaf(2.f, mu_xy, A.c1) * __builtin_amdgcn_rcpf(mu_xx + mu_yy + A.c1) * cs;
                }
            }
        }
    };
    for (int r0 = 0; r0 < R; r0 += kSsimK) {
        step(std::integral_constant<int, 0>{}, r0);      step(std::integral_constant<int, 1>{}, r0 + 1);
        step(std::integral_constant<int, 2>{}, r0 + 2);  step(std::integral_constant<int, 3>{}, r0 + 3);
        step(std::integral_constant<int, 4>{}, r0 + 4);  step(std::integral_constant<int, 5>{}, r0 + 5);
        step(std::integral_constant<int, 6>{}, r0 + 6);  step(std::integral_constant<int, 7>{}, r0 + 7);
        step(std::integral_constant<int, 8>{}, r0 + 8);  step(std::integral_constant<int, 9>{}, r0 + 9);
        step(std::integral_constant<int, 10>{}, r0 + 10);
    }
    double ss_d = (double)ss_acc, cs_d = (double)cs_acc;
    for (int o = 32; o > 0; o >>= 1) { ss_d += __shfl_down(ss_d, o); cs_d += __shfl_down(cs_d, o); }
    if (lane == 0) {
        double *acc = A.acc + (long long)img * kMetricSlots + A.slot + 2 * c;
        if (A.want_ss) atomicAdd(&acc[0], ss_d);
        atomicAdd(&acc[1], cs_d);
    }
}

// ---- next MS-SSIM scale: F.pad(replicate, left / top by p = max(h % 2, w % 2)) then F.avg_pool2d(2) -> planar ----
template <bool INTERLEAVED>
__global__ __launch_bounds__(256) void k_pool2(const float *__restrict__ in, int h, int w, int C, int p, int h2, int w2, float *__restrict__ out)
{
    const int c = blockIdx.y, img = blockIdx.z;
    const long long n = (long long)h2 * w2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int y = (int)(i / w2), x = (int)(i - (long long)y * w2);
        float s = 0.f;
#pragma unroll
        for (int dy = 0; dy < 2; dy++)
#pragma unroll
            for (int dx = 0; dx < 2; dx++) {
                int sy = 2 * y + dy - p, sx = 2 * x + dx - p;
                sy = sy < 0 ? 0 : sy; sx = sx < 0 ? 0 : sx;
                const long long o = INTERLEAVED ? (((long long)img * h + sy) * w + sx) * C + c : (((long long)img * C + c) * h + sy) * w + sx;
                s += in[o];
            }
        out[((long long)img * C + c) * n + i] = s * 0.25f;
    }
}

// first MS-SSIM down-scale, from the caller's interleaved RGB: one thread per output pixel handles the three channels of BOTH
// images (2 x 2 x 3 contiguous floats per source row instead of a 12-byte-strided read per channel)
__global__ __launch_bounds__(256) void k_pool2_rgb(const float *__restrict__ ia, const float *__restrict__ ib, int h, int w, int p, int h2, int w2,
                                                   float *__restrict__ oa, float *__restrict__ ob)
{
    const int img = blockIdx.y;
    const long long n = (long long)h2 * w2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int y = (int)(i / w2), x = (int)(i - (long long)y * w2);
        float sa[3] = { 0.f, 0.f, 0.f }, sb[3] = { 0.f, 0.f, 0.f };
#pragma unroll
        for (int dy = 0; dy < 2; dy++)
#pragma unroll
            for (int dx = 0; dx < 2; dx++) {
                int sy = 2 * y + dy - p, sx = 2 * x + dx - p;
                sy = sy < 0 ? 0 : sy; sx = sx < 0 ? 0 : sx;
                const long long o = (((long long)img * h + sy) * w + sx) * 3;
#pragma unroll
                for (int c = 0; c < 3; c++) { sa[c] += ia[o + c]; sb[c] += ib[o + c]; }
            }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            oa[((long long)img * 3 + c) * n + i] = sa[c] * 0.25f;
            ob[((long long)img * 3 + c) * n + i] = sb[c] * 0.25f;
        }
    }
}

// ---- final: sums -> the three scores per image pair ----
struct FinalArgs {
    double npx3;              // H * W * 3
    double n_ssim;            // outputs of the grey SSIM map (0 = not computed)
    double n_level[5];        // outputs per MS-SSIM scale (0 = not computed)
    double weights[5];
};

__global__ void k_metric_final(const double *__restrict__ acc, FinalArgs F, int B, double *__restrict__ out)
{
    const int img = blockIdx.x * blockDim.x + threadIdx.x;
    if (img >= B) return;
    const double *a = acc + (long long)img * kMetricSlots;
    out[img * 3 + 0] = -10.0 * log10(a[0] / F.npx3 + 1e-8);
    out[img * 3 + 1] = F.n_ssim > 0 ? a[kMetricSlotGrey] / F.n_ssim : nan("");
    if (F.n_level[0] > 0) {
        double mean = 0.0;
        for (int c = 0; c < 3; c++) {
            double prod = 1.0;
            for (int l = 0; l < 5; l++) {
                double v = a[kMetricSlotScales + (l * 3 + c) * 2 + (l == 4 ? 0 : 1)] / F.n_level[l];     // cs for scales 0..3, ssim for the last
                v = v > 0.0 ? v : 0.0;                                                   // torch.relu
                prod *= pow(v, F.weights[l]);
            }
            mean += prod;
        }
        out[img * 3 + 2] = mean / 3.0;
    } else {
        out[img * 3 + 2] = nan("");
    }
}

// ---- launchers ----
static int grid_for(long long n)
{
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : b > 4096 ? 4096 : b);
}

void launch_metric_prep(hipStream_t st, const float *a, const float *b, int B, long long npx, double *acc, unsigned char *ga, unsigned char *gb)
{
    hipLaunchKernelGGL(k_metric_prep, dim3(grid_for((npx & 3) == 0 ? npx / 4 : npx), B), dim3(256), 0, st, a, b, npx, acc, ga, gb);
}

void launch_metric_pool_grey(hipStream_t st, const unsigned char *ga, const unsigned char *gb, int B, int H, int W, int f, int hp, int wp, float *xa, float *xb)
{
    hipLaunchKernelGGL(k_metric_pool_grey, dim3(grid_for((long long)hp * wp), B), dim3(256), 0, st, ga, gb, H, W, f, hp, wp, xa, xb);
}

void launch_ssim_level(hipStream_t st, bool interleaved, const float *xa, const float *xb, int B, int C, int h, int w, const float *g11, double *acc, int slot,
                       bool want_ss, float *pool_a, float *pool_b)
{
    SsimArgs A;
    A.xa = xa; A.xb = xb; A.h = h; A.w = w; A.C = C;
    for (int i = 0; i < kSsimK; i++) A.g[i] = g11[i];
    A.c1 = (float)(0.01 * 0.01); A.c2 = (float)(0.03 * 0.03);
    A.acc = acc; A.slot = slot; A.want_ss = want_ss ? 1 : 0;
    A.pool_a = pool_a; A.pool_b = pool_b;
    const int oh = h - (kSsimK - 1), ow = w - (kSsimK - 1);
    // strips of 128 output rows (138 input rows: 8 % of the horizontal sums are formed twice) while that still leaves eight waves per SIMD of the
    // chip; 64 (16 %) below
    const int nsx = (ow + kSsimCols - 1) / kSsimCols;
    const long long waves128 = (long long)nsx * ((oh + 127) / 128) * C * B;
    A.strip_rows = waves128 >= 8LL * 4 * 256 ? 128 : waves128 >= 2LL * 4 * 256 ? 64 : 32;
    const int strips = nsx * ((oh + A.strip_rows - 1) / A.strip_rows);
    if (interleaved) hipLaunchKernelGGL(k_ssim_strip<true>, dim3(strips, 1, B), dim3(64 * 3), 0, st, A);       // C == 3: one wave per channel
    else hipLaunchKernelGGL(k_ssim_strip<false>, dim3((strips + 3) / 4, C, B), dim3(256), 0, st, A);
}

void launch_pool2(hipStream_t st, bool interleaved, const float *in, int B, int C, int h, int w, int p, int h2, int w2, float *out)
{
    dim3 grid(grid_for((long long)h2 * w2), C, B);
    if (interleaved) hipLaunchKernelGGL(k_pool2<true>, grid, dim3(256), 0, st, in, h, w, C, p, h2, w2, out);
    else hipLaunchKernelGGL(k_pool2<false>, grid, dim3(256), 0, st, in, h, w, C, p, h2, w2, out);
}

void launch_pool2_rgb(hipStream_t st, const float *ia, const float *ib, int B, int h, int w, int p, int h2, int w2, float *oa, float *ob)
{
    hipLaunchKernelGGL(k_pool2_rgb, dim3(grid_for((long long)h2 * w2), B), dim3(256), 0, st, ia, ib, h, w, p, h2, w2, oa, ob);
}

void launch_metric_final(hipStream_t st, const double *acc, int B, long long npx, long long n_ssim, const long long *n_level, double *out)
{
    FinalArgs F;
    F.npx3 = (double)npx * 3.0;
    F.n_ssim = (double)n_ssim;
    const double wts[5] = { 0.0448, 0.2856, 0.3001, 0.2363, 0.1333 };      // piq/ms_ssim.py default scale_weights (float32 tensor)
    for (int l = 0; l < 5; l++) { F.n_level[l] = (double)n_level[l]; F.weights[l] = (double)(float)wts[l]; }
    hipLaunchKernelGGL(k_metric_final, dim3((B + 63) / 64), dim3(64), 0, st, acc, F, B, out);
}

}  // namespace aej
