// canny.hip -- the EdgeDetection.canny pipeline (src/jpeg/edge_detection.py:70-86) as LDS-halo-staged
// stencil kernels for gfx950:  CLAHE LUT build -> [CLAHE apply + Gaussian 3x3 + bilateral d=5 + histogram]
// -> percentile thresholds -> [Sobel + magnitude + NMS] -> tiled hysteresis to a fix-point.
//
// All arithmetic is integer, or float32 in a fixed order (no contraction), so every stage is bit-identical
// to the CPU oracle.  OpenCV semantics restated per stage are documented in DESIGN.md ("Canny chain").
#include "aej_common.h"
#include "aej_launch.h"
#include <stdlib.h>

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

// Workgroups are dealt round-robin over the 8 XCDs (id % 8 shares an L2).  Stencil tiles re-read their neighbours'
// halo lines, so give each XCD a contiguous range of tiles: bijective remap of a 1-D grid of n workgroups (speed only).
__device__ __forceinline__ long long xcd_remap(long long id, long long n)
{
    const long long q = n / 8, r = n % 8, xcd = id % 8, k = id / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

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
// slots of the hysteresis work queue: the largest power of two <= 2 x tiles (>= tiles: a tile is queued at most once at a time)
int hyst_ring_slots(const Geom &g)
{
    const long long t2 = 2 * hyst_tiles_per_image(g) * g.B;
    int p = 1;
    while (2LL * p <= t2) p *= 2;
    return p;
}

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
__global__ __launch_bounds__(256) void k_clahe_lut(Geom g, const int *__restrict__ tile_hist, unsigned char *__restrict__ lut, double clip_limit)
{
    __shared__ int s[256];
    __shared__ int s_red[4];
    const int tile = blockIdx.x, l = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const long long base = (((long long)b * 3 + l) * 16 + tile) * 256;
    const int area = g.ctw[l] * g.cth[l];
    const float lutScale = 255.0f / (float)area;
    int clip = 0x7fffffff;                 // clipLimit <= 0: nothing is clipped (clahe.cpp)
    if (clip_limit > 0.0) { clip = (int)(clip_limit * (double)area / 256.0); if (clip < 1) clip = 1; }
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
//
// This kernel is bound by the LDS pipe, not by VALU issue or HBM (PMC: the LDS array is busy > 70 % of the kernel's time,
// profiles/r02_blur_pmc.txt), so its structure minimises LDS cycles per pixel:
//   * 256-thread workgroups on 128 x 32 output tiles (3 workgroups per CU; shape and why below): the staged halo is 27 % / 20 % of the tile for the
//     CLAHE / Gaussian images instead of 34 % / 27 % with 64 x 32 tiles;
//   * stage A (CLAHE): the four tile LUTs a pixel blends are pre-packed, per workgroup, into one float4 per input value
//     ("packed LUT": { L[ty1][tx1][v], L[ty1][tx2][v], L[ty2][tx1][v], L[ty2][tx2][v] } as floats), so a pixel costs ONE
//     16-byte LDS gather and no int -> float conversions instead of four byte gathers + four conversions.  A 136 x 38 window
//     rarely crosses a tile-centre line, and almost never one per axis: two packed LUTs are kept (windows that would need
//     more -- tiny planes, the 0.5 % of tiles on a crossing of both axes -- take the general path that reads LUT bytes from
//     global memory); a thread keeps its column for the whole tile, so the column weights live in registers;
//   * stage B (Gaussian): a thread slides down 6 output rows of one column dword, re-using the horizontal [1 2 1] sums of the
//     previous two rows: 3 LDS reads per row instead of 9.  The result is stored one HALFWORD per pixel: the integer 4 g.  Read as
//     an integer, differences of two halfwords are table byte offsets (one SDWA subtraction picks the halves); read as a float16,
//     the same bits are the DENORMAL 4 g * 2^-24, which `v_fma_mix_f32` widens exactly and for free inside the tap's fma.  The
//     weight tables hold the weights times 2^24, so every product -- and with it every partial sum -- is bit for bit the one the
//     unscaled filter forms (the weight sum comes out times 2^24, an exact scaling that the final quotient undoes): no conversion
//     instructions in the bilateral stage, and a window is 24 registers instead of 48;
//   * stage C (bilateral): a thread owns 4 x 2 pixels.  All window reads are 8-byte reads of whole 256-byte row segments
//     (conflict-free), and the weight of a pixel PAIR, which depends only on |difference| and distance, is fetched
//     once and used for both pixels: 76 instead of 96 table gathers per 8 pixels.
// Every float operation per pixel is the same single IEEE operation in the same order as before (and as the CPU oracle).
// ------------------------------------------------------------------------------------------------
// Workgroup shape.  Alone, 512 threads on 128 x 64 tiles (two workgroups per CU, 128 VGPRs with ~40 dwords of spills, 9 % halo) is the
// fastest (round 2: 2.19 ms against 2.32 ms for 256 threads on 128 x 32 tiles -- three workgroups per CU, 16 % halo).  But this kernel
// never runs alone in the throughput path -- it shares the chip with the HBM-bound stages of other sub-batches -- and
// there the smaller footprint wins: resources change hands in units of a third of the LDS / one wave per SIMD instead of a half / two, and the
// 64 x 4K step goes from 7.50 to 7.33 ms (256 threads on 128 x 64 tiles, 235 VGPRs: 7.44; 384 threads on 128 x 48: 3.8 ms alone).
constexpr int kBT = 256;                  // threads per workgroup
constexpr int kBTW = 128, kBTH = 32;      // output tile
constexpr int kBAW = kBTW + 8;            // staged columns: c <-> gx = x0 - 4 + c (bytes per CLAHE row, dwords per Gaussian row)
constexpr int kBAW4 = kBAW / 4;           // 34 dwords per CLAHE row
constexpr int kBAH = kBTH + 6;            // CLAHE rows  [y0-3, y0+TH+3)
constexpr int kBGH = kBTH + 4;            // Gauss rows  [y0-2, y0+TH+2)
constexpr int kBGW = kBAW + 4;            // Gaussian row stride in halfwords; staged column c is stored at c + 2, so that the 8-halfword window
                                          // of a bilateral thread (columns 4 c4 + 2 .. 4 c4 + 9) is two ALIGNED 8-byte reads (conflict-free)
constexpr int kBGW2 = kBGW / 2;           // ... in dwords
static_assert(kBGW % 4 == 0, "Gaussian rows keep 8-byte alignment");
constexpr int kBStripMax = 15;            // tiles of one tile-row handled by one workgroup (tables / histogram stay in LDS); fewer when
                                          // the batch is small, so that a single image still spreads over the whole chip
constexpr int kASlots = kBT / kBAW4;      // 7 row slots: thread t owns column dword t % 34 and rows t / 34 + 7 k
constexpr int kAIter = (kBAH + kASlots - 1) / kASlots;   // 6
constexpr int kBRows = (kBGH + kASlots - 1) / kASlots;      // Gaussian output rows per thread: rows kBRows * (t / 34) .. + kBRows - 1
static_assert(kASlots * kBRows >= kBGH && kASlots * kAIter >= kBAH, "thread -> row mapping must cover the tile");
constexpr float kTwo24 = 16777216.0f, kTwo22 = 4194304.0f;
constexpr int kHistCopies = 6;                           // (6 copies: 39.5 KiB per workgroup, so that three of them leave the colour kernel its 38 KiB --
                                                        // with the tiled planes it stages four rows per half-wave; 8 and 16 copies measured the same speed)
constexpr int kHistStride = 257;          // dwords per histogram copy: odd, so the same bin of different copies sits in different banks

// Member order matters: DS instructions carry a 16-bit immediate offset, so everything addressed with small compile-time offsets
// (parameter arrays, tables, histogram) sits in the first 64 KB and the big Gaussian image, addressed through computed bases, last.
struct __attribute__((aligned(16))) BlurLds {
    int colq[kBAW];                  // byte offset of the column's packed LUT: 0 or sizeof(P[0])
    float colXa[kBAW], colXa1[kBAW];
    int colT1[kBAW], colT2[kBAW];    // clamped CLAHE tile indices of the columns / rows (general path, class detection)
    int rowq[kBAH];                  // byte offset of the row's packed LUT: 0 or sizeof(P[0])
    float rowYa[kBAH], rowYa1[kBAH];
    int rowT1[kBAH], rowT2[kBAH];
    int cls[8];                      // [0..1] (tx1 << 2 | tx2) of column class 0 / 1, [2..3] likewise for rows, [4] / [5] number of class changes
    int pkey[2];                     // which (tx1, tx2, ty1, ty2) each packed LUT currently holds, -1 = none
    // space weight (radius 1, sqrt 2, 2) x colour weight: the float32 product OpenCV forms per tap, indexed by the SIGNED
    // difference d + 256 so the filter needs no |d|
    float cw[3][512];
    unsigned int hist[kHistCopies * kHistStride];   // lane-striped copies of the 256 counters of this strip
    float4 P[2][256];                // packed LUTs: slot 0 = class 0 of both axes, slot 1 = class 1 of the ONE axis that changes
    unsigned int A[kBAH * kBAW4];    // CLAHE image, one byte per pixel
    unsigned int G[kBGH * kBGW2];    // Gaussian image, one halfword per pixel: the integer 4 * value = the float16 DENORMAL 4 * value * 2^-24
};
static_assert(3 * sizeof(BlurLds) <= 160 * 1024, "three workgroups per CU");

// 13 taps in row-major order (OpenCV bilateral_filter, d = 5 => circular mask of radius 2): (dy, dx) =
// (-2,0) (-1,-1) (-1,0) (-1,1) (0,-2) (0,-1) (0,0) (0,1) (0,2) (1,-1) (1,0) (1,1) (2,0).  The weight of a tap is
// space_w[k] * color_w[|delta|]; space_w takes 3 values (radius 1, sqrt 2, 2), so the float32 products are tabulated once per
// workgroup (same multiplication, same rounding) in cw[0 / 1 / 2].  The centre tap has weight 1.
// pair_w: weight (times 2^24) of the pixel pair (a, b) of radius class T; a / b = window dwords, ah / bh = which halfword (constants
// once the loops are unrolled, so one of the four instruction forms survives).
__device__ __forceinline__ int half_diff(unsigned a, int ah, unsigned b, int bh)       // 4 * (difference of the two Gaussian values): |d| <= 1020
{
    int d;
    if (ah == 0 && bh == 0) asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0" : "=v"(d) : "v"(b), "v"(a));
    else if (ah == 0)       asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "=v"(d) : "v"(b), "v"(a));
    else if (bh == 0)       asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(d) : "v"(b), "v"(a));
    else                    asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(d) : "v"(b), "v"(a));
    return d;
}

template <int T>
__device__ __forceinline__ float pair_w(const BlurLds &L, unsigned a, int ah, unsigned b, int bh)
{
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(L.cw[T]) + 1024 + half_diff(a, ah, b, bh));
}

// one tap: sum = fma(v, wgt, sum) with v = the halfword `h` of p widened inside the instruction
__device__ __forceinline__ float tap_fma(unsigned p, int h, float wgt, float sum)
{
    float r;
    if (h) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(wgt), "v"(sum));
    else   asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(wgt), "v"(sum));
    return r;
}
// the first tap of a pixel: fma(v, wgt, +0) = the rounded product (v, wgt >= 0)
__device__ __forceinline__ float tap_first(unsigned p, int h, float wgt)
{
    float r;
    if (h) asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(wgt));
    else   asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(wgt));
    return r;
}

// CLAHE_Interpolation_Body for one pixel from the four LUT values (general path: LUT bytes straight from global memory)
__device__ __forceinline__ float clahe_blend(float l11, float l12, float l21, float l22, float xa, float xa1, float ya, float ya1)
{
    float pa = l11 * xa1, pb = l12 * xa, pc = l21 * xa1, pd = l22 * xa;
    float top = pa + pb, bot = pc + pd;
    float t1 = top * ya1, t2 = bot * ya;
    return t1 + t2;
}

// raw bytes of a tile for stage A: thread t fetches the dwords it will map itself (column dword t % 34, rows t / 34 + 15 k),
// one tile ahead.  `aligned` tiles (staged window inside the image, w % 4 == 0) use dword loads; tiles on the image border
// gather bytes at REFLECT_101-mapped coordinates, which yields the padded CLAHE image the Gaussian needs (Gaussian of the
// padded image = padded Gaussian, the kernel being symmetric).
__device__ __forceinline__ void blur_prefetch(const unsigned char *src, int w, int h, int x0, int y0, bool aligned, unsigned int (&raw)[kAIter])
{
    const int t = threadIdx.x;
    const int col = t % kBAW4, slot = t / kBAW4;
    if (slot >= kASlots) return;
#pragma unroll
    for (int k = 0; k < kAIter; k++) {
        const int j = slot + kASlots * k;
        if (j < kBAH) {
            if (aligned) {
                raw[k] = *reinterpret_cast<const unsigned int *>(src + (long long)(y0 - 3 + j) * w + (x0 - 4) + 4 * col);
            } else {
                const unsigned char *row = src + (long long)reflect101(y0 - 3 + j, h) * w;
                unsigned int v = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) v |= (unsigned int)row[reflect101(x0 - 4 + 4 * col + q, w)] << (8 * q);
                raw[k] = v;
            }
        }
    }
}

__device__ __forceinline__ bool locate_blur_strip(const Geom &g, int strip, int t, int &layer, int &sx, int &ty)
{
    for (int l = 0; l < g.nl; l++) {
        int ntx = cdiv(g.w[l], kBTW), nty = cdiv(g.h[l], kBTH), nsx = cdiv(ntx, strip);
        int n = nsx * nty;
        if (t < n) { layer = l; ty = t / nsx; sx = t - ty * nsx; return true; }
        t -= n;
    }
    return false;
}


// DUMP: the test hooks that copy the CLAHE / Gaussian intermediates out (aej_canny's stage outputs); the production instantiation does not
// carry their pointers -- four scalar registers that, live across the tile loop, tipped the kernel into scalar-register spills, and a
// spilled scalar lives in a lane of a VECTOR register (v144 held nothing else).  Without them the kernel needs 136 vector registers
// instead of 145: a 136- instead of a 152-register allocation, 104 instead of 56 free registers per SIMD lane beside three workgroups
template <bool DUMP>
__global__ __launch_bounds__(kBT) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_clahe_blur(Geom g, CannyBuffers cb, int strip)      // 3 waves per SIMD (136 VGPRs), 39.5 KiB LDS
{
    __shared__ BlurLds L;
    const int tid = threadIdx.x;
    const long long wg = xcd_remap(blockIdx.x, gridDim.x);
    const int strips = gridDim.x / g.B;
    const int b = (int)(wg / strips);
    int l, sx, ty;
    if (!locate_blur_strip(g, strip, (int)(wg - (long long)b * strips), l, sx, ty)) return;
    const int w = g.w[l], h = g.h[l];
    const int ntx = cdiv(w, kBTW);
    const int y0 = ty * kBTH;
    const int tx_begin = sx * strip, tx_end = min(ntx, tx_begin + strip);
    const long long pbase = (long long)b * g.pstride + g.poff[l];
    const unsigned char *src = cb.u8a + pbase;
    const unsigned char *glut = cb.lut + ((long long)b * 3 + l) * 4096;     // [ty][tx][256]

    // a tile is "aligned" when its staged window [x0-4, x0+TW+4) x [y0-3, y0+TH+3) lies inside the image: dword prefetch
    auto is_aligned = [&](int tx) {
        int x0 = tx * kBTW;
        return x0 >= 4 && y0 >= 3 && x0 + kBTW + 4 <= w && y0 + kBTH + 3 <= h && (w % 4) == 0;
    };
    unsigned int raw[kAIter];
    blur_prefetch(src, w, h, tx_begin * kBTW, y0, is_aligned(tx_begin), raw);      // flies while the per-strip tables are built

    // ---- per-strip set-up (the two rounds of the table loop unrolled: their global loads overlap instead of following each other)
    static_assert(512 % kBT == 0, "weight table set-up assumes whole rounds");
#pragma unroll
    for (int i = tid; i < 512; i += kBT) {
        const int d = i - 256;
        const float cwv = i == 0 ? 0.f : cb.color_w[d < 0 ? -d : d];     // slot 0 (d = -256) is never addressed
        L.cw[0][i] = (cb.space_w[5] * cwv) * kTwo24;       // radius 1: OpenCV's float32 product, then the exact scaling
        L.cw[1][i] = (cb.space_w[1] * cwv) * kTwo24;       // radius sqrt(2)
        L.cw[2][i] = (cb.space_w[0] * cwv) * kTwo24;       // radius 2
    }
    for (int i = tid; i < kHistCopies * kHistStride; i += kBT) L.hist[i] = 0;
    if (tid < 2) L.pkey[tid] = -1;
    if (tid < kBAH) {                          // row parameters of CLAHE_Interpolation_Body: fixed for the strip
        const int j = tid;
        const int gy = reflect101(y0 - 3 + j, h);
        const float inv_th = 1.0f / (float)g.cth[l];
        float tyf = (float)gy * inv_th - 0.5f;
        int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
        float ya = tyf - (float)ty1;
        L.rowYa[j] = ya; L.rowYa1[j] = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > 3) ty2 = 3;
        L.rowT1[j] = ty1; L.rowT2[j] = ty2;
    }
    __syncthreads();
    {                                          // row classes: class 0 = the tile pair of row 0, class 1 = the pair after the (single) change
        bool rchg = false;
        if (tid < kBAH) {
            const int j = tid;
            const int key = (L.rowT1[j] << 2) | L.rowT2[j], key0 = (L.rowT1[0] << 2) | L.rowT2[0];
            rchg = j > 0 && key != ((L.rowT1[j - 1] << 2) | L.rowT2[j - 1]);
            if (rchg) L.cls[3] = key;
            if (j == 0) L.cls[2] = key;
            L.rowq[j] = key != key0 ? (int)sizeof(L.P[0]) : 0;
        }
        const int n = __syncthreads_count(rchg);
        if (tid == 0) L.cls[5] = n;            // read by every tile's prologue after its own barrier
    }
    unsigned char *dst = cb.u8b + pbase;

    for (int tx = tx_begin; tx < tx_end; tx++) {
        const int x0 = tx * kBTW;
        const bool full = x0 + kBTW <= w && y0 + kBTH <= h && (w % 4) == 0;
        // ---- column parameters for this tile (reflected at the image border) and their classes.  A thread evaluates its own
        // column, its left neighbour and column 0, so the class test needs no exchange; the barrier that publishes the parameters
        // also counts the class changes (and separates the previous tile's stage C from this tile's writes to A and G).
        if (tid == 0) L.cls[4] = 0;
        if (tid < kBAW) {
            const int gx = reflect101(x0 - 4 + tid, w);
            const float inv_tw = 1.0f / (float)g.ctw[l];
            float txf = (float)gx * inv_tw - 0.5f;
            int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
            float xa = txf - (float)tx1;
            L.colXa[tid] = xa; L.colXa1[tid] = 1.0f - xa;
            if (tx1 < 0) tx1 = 0;
            if (tx2 > 3) tx2 = 3;
            L.colT1[tid] = tx1; L.colT2[tid] = tx2;
        }
        __syncthreads();                       // (also: the previous tile's stage C is finished with G, stage B with A)
        if (tid < kBAW) {
            const int key = (L.colT1[tid] << 2) | L.colT2[tid], key0 = (L.colT1[0] << 2) | L.colT2[0];
            const bool chg = tid > 0 && key != ((L.colT1[tid - 1] << 2) | L.colT2[tid - 1]);
            if (chg) { atomicAdd(&L.cls[4], 1); L.cls[1] = key; }
            if (tid == 0) { L.cls[0] = key; }
            L.colq[tid] = key != key0 ? (int)sizeof(L.P[0]) : 0;
        }
        __syncthreads();
        const int nchg_c = L.cls[4], nchg_r = L.cls[5];
        // the packed-LUT path serves windows with at most one class change in total (two packed LUTs)
        const bool fast = nchg_c + nchg_r <= 1;
        if (fast) {
            // (re)build the packed LUTs this tile needs and does not hold yet: threads 0..255 slot 0, threads 256..511 slot 1
            const int key_s0 = (L.cls[0] << 4) | L.cls[2];
            const int key_s1 = (L.cls[nchg_c] << 4) | L.cls[2 + nchg_r];
            const bool need1 = nchg_c + nchg_r == 1;
            const bool build0 = L.pkey[0] != key_s0, build1 = need1 && L.pkey[1] != key_s1;
            if (build0 || build1) {              // uniform: everything above comes from LDS words every thread reads alike
                for (int e = tid; e < 512; e += kBT) {       // entries 0..255 slot 0, 256..511 slot 1
                    const int v = e & 255, slot = e >> 8;
                    const int key = slot ? key_s1 : key_s0;
                    if (slot ? build1 : build0) {
                        const int ckey = key >> 4, rkey = key & 15;
                        const int tx1 = ckey >> 2, tx2 = ckey & 3, ty1 = rkey >> 2, ty2 = rkey & 3;
                        L.P[slot][v] = make_float4((float)glut[(ty1 * 4 + tx1) * 256 + v], (float)glut[(ty1 * 4 + tx2) * 256 + v],
                                                   (float)glut[(ty2 * 4 + tx1) * 256 + v], (float)glut[(ty2 * 4 + tx2) * 256 + v]);
                    }
                }
                __syncthreads();                   // every thread has compared the old keys; the new entries are visible
                if (tid < 2 && (tid ? build1 : build0)) L.pkey[tid] = tid ? key_s1 : key_s0;
            }
        }
        // Thread -> work mappings are re-derived per tile from a copy of the thread id the compiler cannot see through: otherwise
        // it hoists a few dozen per-thread LDS addresses out of the tile loop and, with stage C needing the whole register
        // budget, parks them in scratch memory.
        int tq = tid;
        asm volatile("" : "+v"(tq));
        const int acol = tq % kBAW4, aslot = tq / kBAW4;      // stage A / B: this thread's column dword and row slot / row group
        // ---- stage A: CLAHE interpolation, 4 pixels (one dword) per item; this thread's column is fixed
        if (aslot < kASlots) {
            const float4 xa = reinterpret_cast<const float4 *>(L.colXa)[acol], xa1 = reinterpret_cast<const float4 *>(L.colXa1)[acol];
            if (fast) {
                const int4 cq = reinterpret_cast<const int4 *>(L.colq)[acol];
                const char *Pb = reinterpret_cast<const char *>(&L.P[0][0]);
#pragma unroll
                for (int k = 0; k < kAIter; k++) {
                    const int j = aslot + kASlots * k;
                    if (j < kBAH) {
                        const unsigned int rv = raw[k];
                        const int rq = L.rowq[j];
                        const float ya = L.rowYa[j], ya1 = L.rowYa1[j];
                        const float4 f0 = *reinterpret_cast<const float4 *>(Pb + ((rv & 0xffu) * 16u + (unsigned)(cq.x + rq)));
                        const float4 f1 = *reinterpret_cast<const float4 *>(Pb + (((rv >> 8) & 0xffu) * 16u + (unsigned)(cq.y + rq)));
                        const float4 f2 = *reinterpret_cast<const float4 *>(Pb + (((rv >> 16) & 0xffu) * 16u + (unsigned)(cq.z + rq)));
                        const float4 f3 = *reinterpret_cast<const float4 *>(Pb + ((rv >> 24) * 16u + (unsigned)(cq.w + rq)));
                        // v_cvt_pk_u8_f32 = cvRound + saturate_cast<uchar> + pack (round-half-even, checked on hardware)
                        unsigned int o = __builtin_amdgcn_cvt_pk_u8_f32(clahe_blend(f0.x, f0.y, f0.z, f0.w, xa.x, xa1.x, ya, ya1), 0, 0u);
                        o = __builtin_amdgcn_cvt_pk_u8_f32(clahe_blend(f1.x, f1.y, f1.z, f1.w, xa.y, xa1.y, ya, ya1), 1, o);
                        o = __builtin_amdgcn_cvt_pk_u8_f32(clahe_blend(f2.x, f2.y, f2.z, f2.w, xa.z, xa1.z, ya, ya1), 2, o);
                        o = __builtin_amdgcn_cvt_pk_u8_f32(clahe_blend(f3.x, f3.y, f3.z, f3.w, xa.w, xa1.w, ya, ya1), 3, o);
                        L.A[j * kBAW4 + acol] = o;
                    }
                }
            } else {
                const int4 t1 = reinterpret_cast<const int4 *>(L.colT1)[acol], t2 = reinterpret_cast<const int4 *>(L.colT2)[acol];
                const int ct1[4] = { t1.x, t1.y, t1.z, t1.w }, ct2[4] = { t2.x, t2.y, t2.z, t2.w };
                const float cxa[4] = { xa.x, xa.y, xa.z, xa.w }, cxa1[4] = { xa1.x, xa1.y, xa1.z, xa1.w };
                for (int k = 0; k < kAIter; k++) {
                    const int j = aslot + kASlots * k;
                    if (j < kBAH) {
                        const unsigned int rv = raw[k];
                        const int r1 = L.rowT1[j] * 1024, r2 = L.rowT2[j] * 1024;
                        const float ya = L.rowYa[j], ya1 = L.rowYa1[j];
                        unsigned int o = 0;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int v = (rv >> (8 * q)) & 0xff;
                            const float res = clahe_blend((float)glut[r1 + ct1[q] * 256 + v], (float)glut[r1 + ct2[q] * 256 + v],
                                                          (float)glut[r2 + ct1[q] * 256 + v], (float)glut[r2 + ct2[q] * 256 + v], cxa[q], cxa1[q], ya, ya1);
                            o = __builtin_amdgcn_cvt_pk_u8_f32(res, q, o);
                        }
                        L.A[j * kBAW4 + acol] = o;
                    }
                }
            }
        }
        // raw[] is free again: the next tile's bytes fly while this tile runs stages B and C
        if (tx + 1 < tx_end) blur_prefetch(src, w, h, (tx + 1) * kBTW, y0, is_aligned(tx + 1), raw);
        __syncthreads();
        if (DUMP && cb.dump_clahe)
            for (int idx = tid; idx < kBTH * kBTW; idx += kBT) {
                int j = idx / kBTW, i = idx - j * kBTW;
                if (x0 + i < w && y0 + j < h)
                    cb.dump_clahe[pbase + (long long)(y0 + j) * w + x0 + i] = reinterpret_cast<const unsigned char *>(L.A)[(j + 3) * kBAW + i + 4];
            }
        // ---- stage B: Gaussian [1 2 1]^2, (sum + 8) >> 4, SWAR on even/odd bytes (16-bit fields hold <= 4088).  The thread walks
        // down its column dword: horizontal sums of a CLAHE row are formed once and combined with those of the two rows above.
        if (aslot < kASlots) {
            const int i4 = acol;
            const int il = i4 > 0 ? i4 - 1 : 0, ir = i4 < kBAW4 - 1 ? i4 + 1 : kBAW4 - 1;   // edge dwords feed unused columns only
            const int jg0 = aslot * kBRows;
            unsigned int he1 = 0, he2 = 0, ho1 = 0, ho2 = 0;     // horizontal sums of the previous row (1) and the one before (2)
#pragma unroll
            for (int rr = 0; rr < kBRows + 2; rr++) {
                const int ja = jg0 + rr;           // CLAHE row; Gaussian row jg = ja - 2 uses CLAHE rows jg .. jg + 2
                if (ja < kBAH) {
                    // two 16-bit fields per word by byte permutes (selector 0x0C = zero byte) of the 6 bytes at columns -1 .. 4:
                    // LE = (-1, 1), A = (0, 2), B = (1, 3), RO = (2, 4); outputs 0, 2 = LE + 2A + B, outputs 1, 3 = A + 2B + RO
                    const unsigned int *row = L.A + ja * kBAW4;
                    const unsigned int m = row[i4], lf = row[il], rt = row[ir];
                    const unsigned int LE = __builtin_amdgcn_perm(m, lf, 0x0C050C03u);
                    const unsigned int Am = m & 0x00FF00FFu;
                    const unsigned int Bm = __builtin_amdgcn_perm(m, m, 0x0C030C01u);
                    const unsigned int RO = __builtin_amdgcn_perm(rt, m, 0x0C040C02u);
                    const unsigned int he = LE + 2u * Am + Bm, ho = Am + 2u * Bm + RO;
                    if (rr >= 2 && ja - 2 < kBGH) {
                        const unsigned int ve = he2 + 2u * he1 + he + 0x00080008u;
                        const unsigned int vo = ho2 + 2u * ho1 + ho + 0x00080008u;
                        const unsigned int e4 = (ve >> 2) & 0x03FC03FCu;     // 4 * pixel 0 | 4 * pixel 2 << 16
                        const unsigned int o4 = (vo >> 2) & 0x03FC03FCu;     // 4 * pixel 1 | 4 * pixel 3 << 16
                        // halfwords (4 p0, 4 p1), (4 p2, 4 p3) at staged columns 4 i4 + 2 .. + 5
                        unsigned int *gp = &L.G[(ja - 2) * kBGW2 + 2 * i4 + 1];
                        gp[0] = __builtin_amdgcn_perm(o4, e4, 0x05040100u);
                        gp[1] = __builtin_amdgcn_perm(o4, e4, 0x07060302u);
                    }
                    he2 = he1; he1 = he; ho2 = ho1; ho1 = ho;
                }
            }
        }
        __syncthreads();
        if (DUMP && cb.dump_gauss)
            for (int idx = tid; idx < kBTH * kBTW; idx += kBT) {
                int j = idx / kBTW, i = idx - j * kBTW;
                if (x0 + i < w && y0 + j < h)
                    cb.dump_gauss[pbase + (long long)(y0 + j) * w + x0 + i] = (unsigned char)(reinterpret_cast<const unsigned short *>(L.G)[(j + 2) * kBGW + i + 6] >> 2);
            }
        // ---- stage C: bilateral; the thread owns output pixels (x0 + 4 c4 + c, y0 + 4 rg + r), c = 0..3, r = 0..3, as two row pairs
        // handled one after the other ("it" = 0, 1; yy = 4 rg + 2 it is the pair's first row).
        // Window element W[wr][wc] = Gaussian row 4 rg + wr (wr = 0..7), column 4 c4 + 2 + wc (wc = 0..7 <-> dx = -2..5 from output
        // column 0); in terms of the pair's own rows ro = 2 it: W[ro + wr'] is dy = wr' - 2 from the pair's first output row, and output
        // pixel (r, c) of the pair is W[ro + r + 2][c + 2].  The second pair re-uses six of the eight window rows and the 18 pair
        // weights that straddle the two pairs (its "above" pairs are the first pair's "below" pairs): 58 table gathers instead of 76.
        int tc = tid;
        asm volatile("" : "+v"(tc));
        const int c4 = tc & 31;
        unsigned int *hcopy = L.hist + (tc % kHistCopies) * kHistStride;
        static_assert(kBTH == 4 * (kBT / 32), "one group of four rows per 32 threads");
        unsigned int W[8][4];          // halfword wc of row wr = half (wc & 1) of W[wr][wc >> 1]
        {
            // sixteen aligned 8-byte reads (256 contiguous bytes per 32 lanes: conflict-free), issued as written: left to itself
            // the compiler narrows them to the halfwords that are used and emits reads at a lane stride that collide in the banks
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            u32x2 m[16];
            const unsigned int base = (unsigned int)(size_t)(L.G + 4 * (tc >> 5) * kBGW2 + 2 * c4 + 2);       // W[0][0]: 8-byte aligned LDS address
            constexpr int RS = kBGW2 * 4;
            static_assert(7 * RS + 8 < 65536 && (RS % 8) == 0, "DS offset field, alignment");
            asm volatile("ds_read_b64 %0, %16\n\tds_read_b64 %1, %16 offset:8\n\t"
                         "ds_read_b64 %2, %16 offset:%c17\n\tds_read_b64 %3, %16 offset:%c18\n\t"
                         "ds_read_b64 %4, %16 offset:%c19\n\tds_read_b64 %5, %16 offset:%c20\n\t"
                         "ds_read_b64 %6, %16 offset:%c21\n\tds_read_b64 %7, %16 offset:%c22\n\t"
                         "ds_read_b64 %8, %16 offset:%c23\n\tds_read_b64 %9, %16 offset:%c24\n\t"
                         "ds_read_b64 %10, %16 offset:%c25\n\tds_read_b64 %11, %16 offset:%c26\n\t"
                         "ds_read_b64 %12, %16 offset:%c27\n\tds_read_b64 %13, %16 offset:%c28\n\t"
                         "ds_read_b64 %14, %16 offset:%c29\n\tds_read_b64 %15, %16 offset:%c30\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(m[0]), "=&v"(m[1]), "=&v"(m[2]), "=&v"(m[3]), "=&v"(m[4]), "=&v"(m[5]), "=&v"(m[6]), "=&v"(m[7]), "=&v"(m[8]),
                           "=&v"(m[9]), "=&v"(m[10]), "=&v"(m[11]), "=&v"(m[12]), "=&v"(m[13]), "=&v"(m[14]), "=&v"(m[15])
                         : "v"(base), "i"(RS), "i"(RS + 8), "i"(2 * RS), "i"(2 * RS + 8), "i"(3 * RS), "i"(3 * RS + 8), "i"(4 * RS), "i"(4 * RS + 8),
                           "i"(5 * RS), "i"(5 * RS + 8), "i"(6 * RS), "i"(6 * RS + 8), "i"(7 * RS), "i"(7 * RS + 8)
                         : "memory");
#pragma unroll
            for (int wr = 0; wr < 8; wr++) {
                W[wr][0] = m[2 * wr].x; W[wr][1] = m[2 * wr].y; W[wr][2] = m[2 * wr + 1].x; W[wr][3] = m[2 * wr + 1].y;
            }
        }
        float two24 = kTwo24;
        asm volatile("" : "+s"(two24));          // the centre weight lives in a scalar register (VOP3P takes no literal)
        auto PW0 = [&](int ar, int ac, int br, int bc) { return pair_w<0>(L, W[ar][ac >> 1], ac & 1, W[br][bc >> 1], bc & 1); };
        auto PW1 = [&](int ar, int ac, int br, int bc) { return pair_w<1>(L, W[ar][ac >> 1], ac & 1, W[br][bc >> 1], bc & 1); };
        auto PW2 = [&](int ar, int ac, int br, int bc) { return pair_w<2>(L, W[ar][ac >> 1], ac & 1, W[br][bc >> 1], bc & 1); };
        // weights of the pairs between the first pair's rows and the second's, as the first pair fetched them
        float cV1[4], cV2a[4], cV2b[4], cD1[4], cD2[4];
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int ro = 2 * it;
            const int yy = 4 * (tc >> 5) + ro;
            // Pair weights: each is fetched once and used by both pixels of the pair when both belong to this thread.  Indexing
            // (window rows relative to ro): H1[r][i] = pair (r+2, 1+i)-(r+2, 2+i); H2[r][i] = (r+2, i)-(r+2, i+2); V1[rr][c] = (1+rr, c+2)-(2+rr, c+2);
            // V2[rr][c] = (rr, c+2)-(rr+2, c+2); D1[rr][i] = (1+rr, 1+i)-(2+rr, 2+i); D2[rr][i] = (1+rr, 2+i)-(2+rr, 1+i).
            // The two output rows are done one after the other (a scheduling barrier keeps the second row's gathers from being
            // hoisted over the first row's sums); the pairs between them (V1[1], D1[1], D2[1]) carry over.
            float sums[8], wsums[8];
            float V1m[4], D1m[5], D2m[5];
#pragma unroll
            for (int c = 0; c < 4; c++) V1m[c] = PW0(ro + 2, c + 2, ro + 3, c + 2);
#pragma unroll
            for (int i = 0; i < 5; i++) {
                D1m[i] = PW1(ro + 2, 1 + i, ro + 3, 2 + i);
                D2m[i] = PW1(ro + 2, 2 + i, ro + 3, 1 + i);
            }
#pragma unroll
            for (int r = 0; r < 2; r++) {
                float H1[5], H2[6], V1o[4], V2a[4], V2b[4], D1o[4], D2o[4];
#pragma unroll
                for (int i = 0; i < 5; i++) H1[i] = PW0(ro + r + 2, 1 + i, ro + r + 2, 2 + i);
#pragma unroll
                for (int i = 0; i < 6; i++) H2[i] = PW2(ro + r + 2, i, ro + r + 2, i + 2);
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    // (-2, 0): the second pair's are the first pair's (+2, 0)
                    V2a[c] = it == 1 ? (r == 0 ? cV2a[c] : cV2b[c]) : PW2(ro + r, c + 2, ro + r + 2, c + 2);
                    V2b[c] = PW2(ro + r + 2, c + 2, ro + r + 4, c + 2);              // (+2, 0)
                    // the vertical / diagonal pairs towards the row that is NOT this pair's other output row
                    if (r == 0) {
                        V1o[c] = it == 1 ? cV1[c] : PW0(ro + 1, c + 2, ro + 2, c + 2);
                        D1o[c] = it == 1 && c >= 1 ? cD1[c - 1] : PW1(ro + 1, 1 + c, ro + 2, 2 + c);      // (-1,-1)
                        D2o[c] = it == 1 && c <= 2 ? cD2[c + 1] : PW1(ro + 1, 3 + c, ro + 2, 2 + c);      // (-1,+1)
                    } else {
                        V1o[c] = PW0(ro + 3, c + 2, ro + 4, c + 2);
                        D1o[c] = PW1(ro + 3, 2 + c, ro + 4, 3 + c);      // (+1,+1)
                        D2o[c] = PW1(ro + 3, 2 + c, ro + 4, 1 + c);      // (+1,-1)
                    }
                }
                if (it == 0) {
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        if (r == 0) cV2a[c] = V2b[c];
                        else { cV2b[c] = V2b[c]; cV1[c] = V1o[c]; cD1[c] = D1o[c]; cD2[c] = D2o[c]; }
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    auto tap = [&](float &sum, float &wsum, int wr, int wc, float wgt) {
                        wsum = wsum + wgt;
                        sum = tap_fma(W[ro + wr][wc >> 1], wc & 1, wgt, sum);
                    };
                    float wsum = V2a[c];                                                 // (-2,  0): 0 + w = w, fma(v, w, 0) = v * w
                    float sum = tap_first(W[ro + r][(c + 2) >> 1], (c + 2) & 1, V2a[c]);
                    tap(sum, wsum, r + 1, c + 1, r == 0 ? D1o[c] : D1m[c]);              // (-1, -1)
                    tap(sum, wsum, r + 1, c + 2, r == 0 ? V1o[c] : V1m[c]);              // (-1,  0)
                    tap(sum, wsum, r + 1, c + 3, r == 0 ? D2o[c] : D2m[c + 1]);          // (-1, +1)
                    tap(sum, wsum, r + 2, c, H2[c]);                                     // ( 0, -2)
                    tap(sum, wsum, r + 2, c + 1, H1[c]);                                 // ( 0, -1)
                    tap(sum, wsum, r + 2, c + 2, two24);                                 // centre: weight 1
                    tap(sum, wsum, r + 2, c + 3, H1[c + 1]);                             // ( 0, +1)
                    tap(sum, wsum, r + 2, c + 4, H2[c + 2]);                             // ( 0, +2)
                    tap(sum, wsum, r + 3, c + 1, r == 0 ? D2m[c] : D2o[c]);              // (+1, -1)
                    tap(sum, wsum, r + 3, c + 2, r == 0 ? V1m[c] : V1o[c]);              // (+1,  0)
                    tap(sum, wsum, r + 3, c + 3, r == 0 ? D1m[c + 1] : D1o[c]);          // (+1, +1)
                    tap(sum, wsum, r + 4, c + 2, V2b[c]);                                // (+2,  0)
                    sums[r * 4 + c] = sum; wsums[r * 4 + c] = wsum;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            float z[8];
            // cvRound(sum / wsum): only the nearest integer is needed, so the quotient is first formed with the hardware reciprocal
            // (1 ulp; with the product's rounding |z - sum / wsum| < 3 ulp(255) = 4.6e-5) and the exact IEEE division is redone, per
            // output row, only when some lane's z lies within 2^-13 = 1.2e-4 of a rounding boundary k + 0.5 (one pixel in ~4000),
            // where the two could round apart.
#pragma unroll
            for (int r = 0; r < 2; r++) {
                bool amb = false;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int q = r * 4 + c;
                    z[q] = (sums[q] * __builtin_amdgcn_rcpf(wsums[q])) * kTwo22;
                    amb = amb || (__builtin_fabsf(__builtin_amdgcn_fractf(z[q]) - 0.5f) < 1.220703125e-4f);
                }
                if (__any(amb)) {
#pragma unroll
                    for (int c = 0; c < 4; c++) z[r * 4 + c] = (sums[r * 4 + c] / wsums[r * 4 + c]) * kTwo22;
                }
            }
#pragma unroll
            for (int r = 0; r < 2; r++) {
                unsigned int packed = 0;
#pragma unroll
                for (int c = 0; c < 4; c++) packed = __builtin_amdgcn_cvt_pk_u8_f32(z[r * 4 + c], c, packed);
                const int y = y0 + yy + r;
                if (full) {
                    *reinterpret_cast<unsigned int *>(dst + (long long)y * w + x0 + 4 * c4) = packed;
                    // The filtered image is locally flat: lanes of one LDS atomic instruction often hit the same counter and
                    // serialise.  Four equal pixels take one add of 4 instead of four adds (the common case in flat regions).
                    if (packed == (packed & 0xffu) * 0x01010101u) {
                        atomicAdd(&hcopy[packed & 0xffu], 4u);
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; c++) atomicAdd(&hcopy[(packed >> (8 * c)) & 0xffu], 1u);
                    }
                } else if (y < h) {
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        if (x0 + 4 * c4 + c < w) {
                            dst[(long long)y * w + x0 + 4 * c4 + c] = (unsigned char)(packed >> (8 * c));
                            atomicAdd(&hcopy[(packed >> (8 * c)) & 0xffu], 1u);
                        }
                }
            }
        }
    }
    __syncthreads();
    if (tid < 256) {
        unsigned int c = 0;
#pragma unroll
        for (int k = 0; k < kHistCopies; k++) c += L.hist[k * kHistStride + tid];
        if (c) atomicAdd(&cb.blur_hist[((long long)b * 3 + l) * 256 + tid], (int)c);
    }
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

__global__ __launch_bounds__(256) void k_thresholds(Geom g, const int *__restrict__ blur_hist, int *__restrict__ thr, double low_q, double high_q, int l2)
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
    double lo = pct_from(cum, nullptr, n, low_q, tid, &s_a, &s_b);
    double hi = pct_from(cum, nullptr, n, high_q, tid, &s_a, &s_b);
    if (tid == 0) {
        // cv::Canny (canny.cpp): with L2gradient the squared magnitudes are compared
        if (lo > hi) { double t = lo; lo = hi; hi = t; }
        if (l2) {
            if (lo > 32767.0) lo = 32767.0;
            if (hi > 32767.0) hi = 32767.0;
            if (lo > 0) lo *= lo;
            if (hi > 0) hi *= hi;
        }
        thr[((long long)b * 3 + l) * 2 + 0] = (int)floor(lo);
        thr[((long long)b * 3 + l) * 2 + 1] = (int)floor(hi);
    }
}

// ------------------------------------------------------------------------------------------------
// a-8 part 1: Sobel 3x3 (BORDER_REPLICATE), magnitude dx^2+dy^2, non-maximum suppression.
// Output: two bit-planes (one 64-bit word per 64 pixels of a row): `weak` = NMS survivors with
// low < mag <= high (OpenCV map value 0), `strong` = survivors with mag > high (map value 2).
// Stage 0 stages the blurred bytes (replicate padding = clamped source coordinates) through registers with all
// loads in flight; stage 1 computes gradient + magnitude for 4 pixels per item with SWAR [1 2 1] sums;
// stage 2 does the NMS test with one wave per 64-pixel row segment so that the output words are wave ballots.
// ------------------------------------------------------------------------------------------------
constexpr int kStripMax = 8;    // tiles of one tile-row handled by one workgroup (fewer for small batches, see launch_sobel_nms)

__device__ __forceinline__ bool locate_strip(const Geom &g, int strip, int t, int &layer, int &sx, int &ty)
{
    for (int l = 0; l < g.nl; l++) {
        int ntx = cdiv(g.w[l], kBlurTW), nty = cdiv(g.h[l], kBlurTH), nsx = cdiv(ntx, strip);
        int n = nsx * nty;
        if (t < n) { layer = l; ty = t / nsx; sx = t - ty * nsx; return true; }
        t -= n;
    }
    return false;
}

constexpr int kNW4 = (kBlurTW + 8) / 4;      // 18 dwords per staged row: columns c <-> gx = x0 - 4 + c
constexpr int kNUH = kBlurTH + 4;            // u8 rows  [y0-2, y0+TH+2)
constexpr int kNMH = kBlurTH + 2;            // mag rows [y0-1, y0+TH+1)
constexpr int kNMW = kBlurTW + 8;            // mag row stride (ints): a multiple of 4, so that the four values of an item are ONE aligned 16-byte
                                             // LDS write (as 4-byte writes at a 16-byte lane stride they ran into 4-way bank conflicts); every reader walks rows
constexpr int kNRaw = (kNUH * kNW4 + 255) / 256;
constexpr int kS1Rows = 3;                   // output rows per stage-1 thread: 256 / 18 = 14 row groups x 3 rows >= 34
static_assert((256 / kNW4) * kS1Rows >= kNMH, "stage-1 thread -> row mapping must cover the magnitude rows");

struct __attribute__((aligned(16))) NmsLds {
    int M[kNMH * kNMW];
    int G[kNMH * kNMW];        // dx (low 16 bits) | dy << 16
    unsigned int U[kNUH * kNW4];
};
static_assert((kNMW % 4) == 0 && (kNMH * kNMW) % 4 == 0, "16-byte aligned item writes");

__device__ __forceinline__ void nms_prefetch(const unsigned char *src, int w, int x0, int y0, unsigned int (&raw)[kNRaw])
{
#pragma unroll
    for (int k = 0; k < kNRaw; k++) {
        int idx = threadIdx.x + k * 256;
        if (idx < kNUH * kNW4) {
            int j = idx / kNW4, i4 = idx - j * kNW4;
            raw[k] = *reinterpret_cast<const unsigned int *>(src + (long long)(y0 - 2 + j) * w + (x0 - 4) + 4 * i4);
        }
    }
}

// workgroup = strip of kStrip tiles of one tile-row; the blurred bytes of the next tile are prefetched into registers while
// the current tile is processed
template <bool L2>
__global__ __launch_bounds__(256) void k_sobel_nms(Geom g, CannyBuffers cb, int strip)
{
    __shared__ NmsLds L;
    const int tid = threadIdx.x;
    const long long wg = xcd_remap(blockIdx.x, gridDim.x);
    const int strips = gridDim.x / g.B;
    const int b = (int)(wg / strips);
    int l, sx, ty;
    if (!locate_strip(g, strip, (int)(wg - (long long)b * strips), l, sx, ty)) return;
    const int w = g.w[l], h = g.h[l];
    const int ntx = cdiv(w, kBlurTW);
    const int y0 = ty * kBlurTH;
    const int tx_begin = sx * strip, tx_end = min(ntx, tx_begin + strip);
    const long long pbase = (long long)b * g.pstride + g.poff[l];
    const unsigned char *src = cb.u8b + pbase;
    const int low = cb.thr[((long long)b * 3 + l) * 2], high = cb.thr[((long long)b * 3 + l) * 2 + 1];
    const int wpr = g.wpr[l];            // (once: indexing the kernel-argument array with `l` inside the row loop was a kernarg load + wait per row)
    unsigned long long *wk = cb.weak + (long long)b * g.bpstride + g.bpoff[l];
    unsigned long long *sg = cb.strong + (long long)b * g.bpstride + g.bpoff[l];
    auto is_aligned = [&](int tx) {
        int x0 = tx * kBlurTW;
        return x0 >= 4 && y0 >= 2 && x0 + kBlurTW + 4 <= w && y0 + kBlurTH + 2 <= h && (w % 4) == 0;
    };
    unsigned int raw[kNRaw];
    if (is_aligned(tx_begin)) nms_prefetch(src, w, tx_begin * kBlurTW, y0, raw);

    for (int tx = tx_begin; tx < tx_end; tx++) {
        const int x0 = tx * kBlurTW;
        const bool aligned = is_aligned(tx);
        // ---- stage 0 (the previous tile's stage 2 reads only M / G, its stage 1 finished before the barrier below it)
        if (aligned) {
#pragma unroll
            for (int k = 0; k < kNRaw; k++) {
                int idx = tid + k * 256;
                if (idx < kNUH * kNW4) L.U[idx] = raw[k];
            }
        } else {
            unsigned char *U8 = reinterpret_cast<unsigned char *>(L.U);
            for (int idx = tid; idx < kNUH * kNW4 * 4; idx += 256) {
                int j = idx / (kNW4 * 4), c = idx - j * (kNW4 * 4);
                int gx = x0 - 4 + c, gy = y0 - 2 + j;
                gx = gx < 0 ? 0 : gx >= w ? w - 1 : gx;
                gy = gy < 0 ? 0 : gy >= h ? h - 1 : gy;
                U8[idx] = src[(long long)gy * w + gx];
            }
        }
        __syncthreads();
        if (tx + 1 < tx_end && is_aligned(tx + 1)) nms_prefetch(src, w, (tx + 1) * kBlurTW, y0, raw);

        // ---- stage 1: gradients and magnitudes, 4 pixels (one column dword) x kS1Rows output rows per thread; rows gy = y0-1+jm,
        // columns gx = x0-4+4*i4 .. +3.  The thread slides down its column: the horizontal parts of a source row are formed once
        // and used by the three output rows that see it (the first version recomputed them per output row: 15.5 -> 11 instructions
        // per pixel in this stage, 9 -> 5 LDS reads per item).
        {
            int tq = tid;
            asm volatile("" : "+v"(tq));           // (keeps the per-thread LDS addresses from being hoisted out of the tile loop into scratch)
            const int i4 = tq % kNW4, grp = tq / kNW4;
            const int jm0 = grp * kS1Rows;
            if (jm0 < kNMH) {
                const int il = i4 > 0 ? i4 - 1 : 0, ir = i4 < kNW4 - 1 ? i4 + 1 : kNW4 - 1;   // edge dwords feed unused columns only
                // Per source row, four words of two 16-bit fields each, made by byte permutes (selector 0x0C = zero byte) from the
                // 6 bytes at columns -1 .. 4:  LE = (-1, 1), A = (0, 2), B = (1, 3), RO = (2, 4).  Outputs 0 and 2 have
                // (left, mid, right) = (LE, A, B), outputs 1 and 3 have (A, B, RO).
                unsigned int LEv[3], Av[3], Bv[3], ROv[3], he[3], ho[3];      // the last three source rows (index = source row % 3)
                typedef short s16x2 __attribute__((ext_vector_type(2)));
                auto sub2 = [](unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, (s16x2)(__builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b))); };
#pragma unroll
                for (int r = 0; r < kS1Rows + 2; r++) {
                    const int ju = jm0 + r;              // source (U) row; output row jm = ju - 2 uses U rows jm .. jm + 2
                    if (ju < kNUH) {
                        const unsigned int *row = L.U + ju * kNW4;
                        const unsigned int m = row[i4], lf = row[il], rt = row[ir];
                        const int c = r % 3;
                        LEv[c] = __builtin_amdgcn_perm(m, lf, 0x0C050C03u);
                        Av[c] = m & 0x00FF00FFu;
                        Bv[c] = __builtin_amdgcn_perm(m, m, 0x0C030C01u);
                        ROv[c] = __builtin_amdgcn_perm(rt, m, 0x0C040C02u);
                        he[c] = LEv[c] + 2u * Av[c] + Bv[c];          // horizontal [1 2 1] at columns 0, 2
                        ho[c] = Av[c] + 2u * Bv[c] + ROv[c];          // columns 1, 3
                    }
                    if (r >= 2) {
                        const int jm = ju - 2;
                        if (jm < kNMH) {
                            const int c0 = (r - 2) % 3, c1 = (r - 1) % 3, c2 = r % 3;     // top, middle, bottom source row
                            // vertical [1 2 1] of the left / right columns of outputs (0, 2) and (1, 3)
                            const unsigned vle = LEv[c0] + 2u * LEv[c1] + LEv[c2], vre = Bv[c0] + 2u * Bv[c1] + Bv[c2];
                            const unsigned vlo = Av[c0] + 2u * Av[c1] + Av[c2], vro = ROv[c0] + 2u * ROv[c1] + ROv[c2];
                            // the 16-bit fields are pixel pairs (0, 2) and (1, 3): gradients as packed int16 subtractions, then one
                            // (dx | dy << 16) word per pixel -- the format stage 2 reads -- whose dot product with itself is the magnitude
                            const unsigned dxe = sub2(vre, vle), dxo = sub2(vro, vlo);             // dx of pixels (0, 2), (1, 3)
                            const unsigned dye = sub2(he[c2], he[c0]), dyo = sub2(ho[c2], ho[c0]); // dy
                            unsigned gv[4];
                            gv[0] = __builtin_amdgcn_perm(dye, dxe, 0x05040100u);                  // low halves:  dx0 | dy0 << 16
                            gv[2] = __builtin_amdgcn_perm(dye, dxe, 0x07060302u);                  // high halves: dx2 | dy2 << 16
                            gv[1] = __builtin_amdgcn_perm(dyo, dxo, 0x05040100u);
                            gv[3] = __builtin_amdgcn_perm(dyo, dxo, 0x07060302u);
                            const int gy = y0 - 1 + jm;
                            int mv[4];
#pragma unroll
                            for (int p = 0; p < 4; p++) {
                                const int gx = x0 - 4 + 4 * i4 + p;
                                int m;
                                if (L2) {
                                    m = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, gv[p]), __builtin_bit_cast(s16x2, gv[p]), 0, false);   // dx^2 + dy^2
                                } else {                                                                                                  // |dx| + |dy|
                                    const int dxv = (int)(short)(gv[p] & 0xffffu), dyv = (int)gv[p] >> 16;
                                    m = (dxv < 0 ? -dxv : dxv) + (dyv < 0 ? -dyv : dyv);
                                }
                                if (!aligned && (gx < 0 || gx >= w || gy < 0 || gy >= h)) m = 0;      // magnitude outside the image is 0
                                mv[p] = m;
                            }
                            *reinterpret_cast<int4 *>(&L.M[jm * kNMW + 4 * i4]) = make_int4(mv[0], mv[1], mv[2], mv[3]);
                            *reinterpret_cast<int4 *>(&L.G[jm * kNMW + 4 * i4]) = make_int4((int)gv[0], (int)gv[1], (int)gv[2], (int)gv[3]);
                        }
                    }
                }
            }
        }
        __syncthreads();

        // ---- stage 2: NMS, one wave per 64-pixel row segment
        const int i = tid & 63;
        static_assert(64 % kBlurTH == 0, "a tile's rows must stay inside one 64-row band of the bit-plane");
        const long long bp_tile = bp_index(y0, tx, wpr);
#pragma unroll 2
        for (int j = tid >> 6; j < kBlurTH; j += 4) {
            const int gx = x0 + i, gy = y0 + j;
            int res = 1;
            if (gx < w && gy < h) {
                const int *ma = L.M + (j + 1) * kNMW + (i + 4), *mp = ma - kNMW, *mn = ma + kNMW;
                int m = *ma;
                if (m > low) {
                    const int gd = L.G[(j + 1) * kNMW + (i + 4)];
                    int xs = (int)(short)(gd & 0xffff), ys = gd >> 16;
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
            }
            unsigned long long wmask = __ballot(res == 0), smask = __ballot(res == 2);
            if (i == 0 && gy < h) {
                wk[bp_tile + j] = wmask;           // the 32 rows of a tile lie in one 64-row band of the tile-major bit-plane
                sg[bp_tile + j] = smask;
            }
        }
        __syncthreads();       // M / G are rewritten by the next tile's stage 1
    }
}

// ------------------------------------------------------------------------------------------------
// a-8 part 1, register version (round 3): the same Sobel / magnitude / NMS, without LDS and without barriers.
// One WAVE owns a 64 x 64 tile (= the 64 contiguous words of the tile-major bit-planes): lane (q, j) = (lane >> 4, lane & 15)
// owns the four pixels x0 + 4 j .. + 3 and walks down the band of 16 rows y0 + 16 q ..; so the 16 lanes of a DPP row are the 64
// pixels of one image row, and a lane's left / right neighbour dwords are one DPP row shift away (the lanes 0 / 15 of a row read
// the dword beyond the tile themselves: their `old` operand).  Per source row the horizontal [1 2 1] sums and differences of the
// SIX pixels -1 .. 4 are formed once as three words of two 16-bit fields (pixel pairs (-1, 2), (0, 3), (1, 4)) and slide down
// through the three rows a gradient needs; the magnitudes of the six pixels of three rows stay in registers, so the NMS of the
// four own pixels needs no exchange at all.  The NMS itself is branch-free per pixel column and skipped for a column in which no
// lane of the wave has a candidate (m > low).  Results leave as 16-bit pieces of the bit-plane words: two DPP OR steps gather the
// nibbles of four lanes (weak in the low, strong in the high half-word).
// Every integer operation is the one the LDS kernel above performs (the tests compare both with the CPU oracle).
// ------------------------------------------------------------------------------------------------
typedef short nms_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_sub16(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, (nms_s16x2)(__builtin_bit_cast(nms_s16x2, a) - __builtin_bit_cast(nms_s16x2, b))); }
__device__ __forceinline__ unsigned pk_add16(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, (nms_s16x2)(__builtin_bit_cast(nms_s16x2, a) + __builtin_bit_cast(nms_s16x2, b))); }
template <int CTRL>
__device__ __forceinline__ unsigned dpp_row(unsigned old, unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xf, 0xf, false); }
constexpr int kDppRowShl1 = 0x101, kDppRowShr1 = 0x111, kDppRowShr2 = 0x112;

struct NmsRowH { unsigned s[3], d[3]; };        // per source row: [1 2 1] sums and right-minus-left differences of the pixel pairs (-1, 2), (0, 3), (1, 4)

constexpr int kSobelDepth = 4;         // source rows in flight ahead of the row being worked on (round 4; 20 = all of them up front, as in round 3)
// one 64 x 64 tile; EDGE = the tile touches the plane's border (clamped source coordinates, magnitudes outside the image are 0)
template <bool L2, bool EDGE>
__device__ __forceinline__ void sobel_nms_tile(const unsigned char *__restrict__ src, int w, int h, int tx, int ty, int low, int high,
                                               unsigned short *__restrict__ wk16, unsigned short *__restrict__ sg16, long long bp_tile, int lane)
{
    const int j = lane & 15, q = lane >> 4;
    const int x0 = tx * 64, px = x0 + 4 * j;
    const int yb0 = ty * 64 + 16 * q;                 // first output row of this lane's band
    constexpr int kRows = 20;                          // source rows yb0 - 2 .. yb0 + 17
    unsigned own[kRows], halo[kRows];
    // the dword beyond the tile's edge is needed by lanes 0 and 15 of a row only (the DPP shifts' `old` operand); the other lanes
    // re-read their own dword instead of being masked off (same cache lines, no exec juggling)
    const int halo_col = j == 0 ? x0 - 4 : j == 15 ? x0 + 64 : px;
    // source dword of column c (a multiple of 4, possibly outside [0, w)) of clamped row y: BORDER_REPLICATE (EDGE tiles)
    auto load_col = [&](int y, int c) -> unsigned {
        const int yc = y < 0 ? 0 : y >= h ? h - 1 : y;
        const int cc = c < 0 ? 0 : c > w - 4 ? w - 4 : c;
        unsigned d = *reinterpret_cast<const unsigned int *>(src + (long long)yc * w + cc);
        if (c < 0) d = (d & 0xffu) * 0x01010101u;
        else if (c > w - 4) d = (d >> 24) * 0x01010101u;
        return d;
    };
    // interior tile: one wave-uniform row base per source row (scalar arithmetic) plus a per-lane 32-bit offset
    // (row0 starts four pixels left of the tile, so that every per-lane offset is non-negative)
    const unsigned own_off = (unsigned)(16 * q * w + 4 * j + 4), halo_off = (unsigned)(16 * q * w + (halo_col - x0) + 4);
    const unsigned char *row0 = src + (long long)(ty * 64 - 2) * w + (x0 - 4);
    auto fetch = [&](int u) {
        if constexpr (EDGE) {
            own[u] = load_col(yb0 - 2 + u, px);
            halo[u] = load_col(yb0 - 2 + u, halo_col);
        } else {
            const unsigned char *row = row0 + (long long)u * w;
            own[u] = *reinterpret_cast<const unsigned int *>(row + own_off);
            halo[u] = *reinterpret_cast<const unsigned int *>(row + halo_off);
        }
    };
    // kDepth source rows are in flight ahead of the row being worked on: all twenty up front cost forty registers for the whole tile
    constexpr int kDepth = kSobelDepth < kRows ? kSobelDepth : kRows;
#pragma unroll
    for (int u = 0; u < kDepth; u++) fetch(u);

    NmsRowH H[3];                                      // the last three source rows (index = u % 3)
    int M[3][6];                                       // magnitudes of pixels -1 .. 4 of the last three gradient rows (index = row % 3)
    unsigned G[3][4];                                  // dx | dy << 16 of the four own pixels, same rows
#pragma unroll
    for (int u = 0; u < kRows; u++) {
        if constexpr (kDepth < kRows) {
            __builtin_amdgcn_sched_barrier(0);         // (the loads of row u + kDepth are issued here, not hoisted to the top)
            if (u + kDepth < kRows) fetch(u + kDepth);
        }
        // ---- horizontal pass of source row u: p[-2], p[-1] = left bytes 2, 3; p[0..3] = own; p[4], p[5] = right bytes 0, 1
        {
            const unsigned m = own[u];
            const unsigned lf = dpp_row<kDppRowShr1>(halo[u], m);      // lane j - 1's dword; lane 0 of the row keeps its halo dword
            const unsigned rt = dpp_row<kDppRowShl1>(halo[u], m);      // lane j + 1's dword; lane 15 keeps its halo dword
            // E(k) = (p[k], p[k + 3]) as two 16-bit fields, k = -2 .. 2 (selector 0x0C = zero byte)
            const unsigned Em2 = __builtin_amdgcn_perm(m, lf, 0x0C050C02u);      // (lf.b2, m.b1)
            const unsigned Em1 = __builtin_amdgcn_perm(m, lf, 0x0C060C03u);      // (lf.b3, m.b2)
            const unsigned E0 = __builtin_amdgcn_perm(m, m, 0x0C030C00u);        // (m.b0, m.b3)
            const unsigned E1 = __builtin_amdgcn_perm(rt, m, 0x0C040C01u);       // (m.b1, rt.b0)
            const unsigned E2 = __builtin_amdgcn_perm(rt, m, 0x0C050C02u);       // (m.b2, rt.b1)
            NmsRowH &r = H[u % 3];
            r.s[0] = Em2 + 2u * Em1 + E0;  r.d[0] = pk_sub16(E0, Em2);           // pixels (-1, 2)
            r.s[1] = Em1 + 2u * E0 + E1;   r.d[1] = pk_sub16(E1, Em1);           // pixels ( 0, 3)
            r.s[2] = E0 + 2u * E1 + E2;    r.d[2] = pk_sub16(E2, E0);            // pixels ( 1, 4)
        }
        if (u < 2) continue;
        // ---- gradient row: image row gy = yb0 - 3 + u from source rows u - 2 (top), u - 1, u (bottom)
        const int gr = (u - 1) % 3;                    // slot of this gradient row
        {
            const NmsRowH &top = H[(u - 2) % 3], &mid = H[(u - 1) % 3], &bot = H[u % 3];
#pragma unroll
            for (int p = 0; p < 3; p++) {
                const unsigned dx = pk_add16(pk_add16(top.d[p], bot.d[p]), pk_add16(mid.d[p], mid.d[p]));
                const unsigned dy = pk_sub16(bot.s[p], top.s[p]);
                const unsigned glo = __builtin_amdgcn_perm(dy, dx, 0x05040100u);      // pixel p - 1: dx | dy << 16
                const unsigned ghi = __builtin_amdgcn_perm(dy, dx, 0x07060302u);      // pixel p + 2
                int mlo, mhi;
                if (L2) {
                    mlo = __builtin_amdgcn_sdot2(__builtin_bit_cast(nms_s16x2, glo), __builtin_bit_cast(nms_s16x2, glo), 0, false);
                    mhi = __builtin_amdgcn_sdot2(__builtin_bit_cast(nms_s16x2, ghi), __builtin_bit_cast(nms_s16x2, ghi), 0, false);
                } else {
                    const int xl = (int)(short)(glo & 0xffffu), yl = (int)glo >> 16, xh = (int)(short)(ghi & 0xffffu), yh = (int)ghi >> 16;
                    mlo = (xl < 0 ? -xl : xl) + (yl < 0 ? -yl : yl);
                    mhi = (xh < 0 ? -xh : xh) + (yh < 0 ? -yh : yh);
                }
                M[gr][p] = mlo;                        // pixels -1, 0, 1
                M[gr][p + 3] = mhi;                    // pixels  2, 3, 4
                if (p >= 1) G[gr][p - 1] = glo;        // own pixels 0, 1
                if (p <= 1) G[gr][p + 2] = ghi;        // own pixels 2, 3
            }
            if constexpr (EDGE) {
                const int gy = yb0 - 3 + u;
                const bool row_out = gy < 0 || gy >= h;
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    const int x = px - 1 + i;
                    if (row_out || x < 0 || x >= w) M[gr][i] = 0;
                }
            }
        }
        if (u < 4) continue;
        // ---- NMS of image row y = yb0 + u - 4: previous / current / next gradient rows are the slots of u - 3, u - 2, u - 1.
        // Written as plain boolean algebra on compare results: the compiler keeps them as lane masks and combines them on the scalar
        // unit, so a pixel costs its compares plus two selects -- no per-lane neighbour selection, no divergent branch.
        const int *Mp = M[(u - 3) % 3], *Mc = M[(u - 2) % 3], *Mn = M[(u - 1) % 3];
        const unsigned *Gc = G[(u - 2) % 3];
        unsigned v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int m = Mc[k + 1];
            const bool cand = m > low;
            if (!__any(cand)) continue;                // no candidate in this pixel column of the wave's four rows
            const unsigned gd = Gc[k];
            const unsigned absg = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(nms_s16x2, gd), (nms_s16x2)(-__builtin_bit_cast(nms_s16x2, gd))));   // |dx| | |dy| << 16
            const int ax = (int)(absg & 0xffffu);
            const int ay = (int)((absg >> 1) & 0x7fff8000u);          // |dy| << 15
            const int t22 = ax * 13573;
            const int t67 = t22 + (int)(absg << 16);                  // + (|dx| << 16)
            const bool c_h = ay < t22, c_v = ay > t67;
            const bool neg = (int)(gd ^ (gd << 16)) < 0;              // sign(dx) != sign(dy)
            const bool k_h = (m > Mc[k]) & (m >= Mc[k + 2]);
            const bool k_v = (m > Mp[k + 1]) & (m >= Mn[k + 1]);
            const bool k_dp = (m > Mp[k]) & (m > Mn[k + 2]);          // s = +1
            const bool k_dm = (m > Mp[k + 2]) & (m > Mn[k]);          // s = -1
            const bool k_d = (neg & k_dm) | (!neg & k_dp);
            const bool keep = cand & ((c_h & k_h) | (!c_h & ((c_v & k_v) | (!c_v & k_d))));
            const bool strong = keep & (m > high);
            v |= strong ? (0x10000u << k) : 0u;
            v |= (keep & !strong) ? (1u << k) : 0u;
        }
        // this lane's nibbles (weak in bits 0..3, strong in bits 16..19), shifted to its place among four lanes; two DPP OR steps
        // gather the 16-bit pieces in lanes 3, 7, 11, 15 of every row
        v <<= 4 * (j & 3);
        v |= dpp_row<kDppRowShr1>(0u, v);
        v |= dpp_row<kDppRowShr2>(0u, v);
        const int y = yb0 + u - 4;
        if ((j & 3) == 3 && (!EDGE || y < h)) {
            const long long o = (bp_tile + (y & 63)) * 4 + (j >> 2);      // 16-bit piece j / 4 of the row's word
            wk16[o] = (unsigned short)(v & 0xffffu);
            sg16[o] = (unsigned short)(v >> 16);
        }
    }
}

// Waves per SIMD the register budget is cut for.  Round 3 loaded a tile's twenty source rows up front (forty registers for the whole tile):
// 96 registers, five waves.  Round 4 keeps 4 rows in flight instead, which frees the registers for more waves, and more
// waves hide the latency the shorter look-ahead exposes (profiles/r04_ab_sobel_occupancy.txt, 64 x 4K, stage / step in ms):
//   5 waves, all rows up front 0.77 / 5.84-5.93     6 waves, 8 rows 0.724 / 5.72-5.80     7 waves, 4 rows 0.70 / 5.73-5.83
//   8 waves, 4 rows (64 registers, two spilled dwords) 0.69 / 5.75-5.80  <- this       8 waves, 5 rows: slower (36 bytes of spills)
// (natural images: 0.905 -> 0.86 ms.)
template <bool L2>
__global__ __launch_bounds__(256, 8) void k_sobel_nms_reg(Geom g, CannyBuffers cb, long long tiles_per_img, long long total_tiles, int xcd_contiguous)
{
    const int lane = threadIdx.x & 63;
    // the tile index as a 32-bit SCALAR (wave index through readfirstlane): the division by the tiles per image, the walk over the layers and
    // the tile's base addresses are then scalar work, not a 64-bit vector division per wave.
    // Workgroups are dealt round-robin over the XCDs: with plain indices a workgroup's left / right halo dwords and its four overlap rows
    // sit in lines that only OTHER XCDs' L2s hold, and are fetched from memory again (round 5, profiles/r05_counter_calibration.txt);
    // xcd_contiguous gives each XCD a contiguous range of workgroups (speed only).
    int wg = (int)blockIdx.x;
    if (xcd_contiguous) {
        const int n = (int)gridDim.x, q = n >> 3, r = n & 7, xcd = wg & 7, k = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int T = wg * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if ((long long)T >= total_tiles) return;
    const int tpi = (int)tiles_per_img;
    const int b = T / tpi;
    int l, tx, ty, ntx, nty, tbase;
    if (!locate_tile(g, kHystTile, kHystTile, T - b * tpi, l, tx, ty, ntx, nty, tbase)) return;
    const int w = l == 0 ? g.w[0] : l == 1 ? g.w[1] : g.w[2], h = l == 0 ? g.h[0] : l == 1 ? g.h[1] : g.h[2], wpr = l == 0 ? g.wpr[0] : l == 1 ? g.wpr[1] : g.wpr[2];
    const long long poff_l = l == 0 ? g.poff[0] : l == 1 ? g.poff[1] : g.poff[2], bpoff_l = l == 0 ? g.bpoff[0] : l == 1 ? g.bpoff[1] : g.bpoff[2];
    const unsigned char *src = cb.u8b + (long long)b * g.pstride + poff_l;
    const int low = cb.thr[((long long)b * 3 + l) * 2], high = cb.thr[((long long)b * 3 + l) * 2 + 1];
    unsigned short *wk16 = reinterpret_cast<unsigned short *>(cb.weak + (long long)b * g.bpstride + bpoff_l);
    unsigned short *sg16 = reinterpret_cast<unsigned short *>(cb.strong + (long long)b * g.bpstride + bpoff_l);
    const long long bp_tile = bp_index(ty * 64, tx, wpr);
    // tiles whose 68 x 68 source window leaves the plane: clamped loads (BORDER_REPLICATE), magnitudes outside the image are 0
    const bool edge_tile = tx == 0 || tx * 64 + 68 > w || ty == 0 || ty * 64 + 66 > h;
    if (edge_tile) sobel_nms_tile<L2, true>(src, w, h, tx, ty, low, high, wk16, sg16, bp_tile, lane);
    else sobel_nms_tile<L2, false>(src, w, h, tx, ty, low, high, wk16, sg16, bp_tile, lane);
}

// ------------------------------------------------------------------------------------------------
// a-8 part 2: hysteresis on the bit-planes.  One WAVE owns a 64x64 tile: lane r holds row r of the
// strong and weak planes as 64-bit words, the 8-neighbourhood dilation is shifts + two lane shuffles, and
// horizontal runs are filled in one step with a Kogge-Stone occluded fill; the loop runs in registers until
// the tile is at its fix-point for the current halo.  A tile whose border changed dirties exactly the
// neighbours that see a new pixel (device work list for the next pass, de-duplicated with per-parity flags; from
// pass 1 on the wave follows one of them itself, see CHASE below).  Strong bits only ever get set, so the global
// fix-point is unique = OpenCV's stack flood fill.
// ------------------------------------------------------------------------------------------------
// Kogge-Stone occluded fill along the rows: the bits of `seed` spread through runs of `open` bits, to the left and to the right.  The
// masks "the next 1, 2, 4, ... bits are all open" do not depend on the seed: they are formed once per tile (RunMasks), so that a fill is
// 6 x 2 x (shift, and, or).
struct RunMasks { unsigned long long l1, l2, l4, l8, l16, l32, r1, r2, r4, r8, r16, r32; };
__device__ __forceinline__ RunMasks run_masks(unsigned long long open)
{
    RunMasks m;
    m.l1 = open;                 m.r1 = open;
    m.l2 = m.l1 & (m.l1 << 1);   m.r2 = m.r1 & (m.r1 >> 1);
    m.l4 = m.l2 & (m.l2 << 2);   m.r4 = m.r2 & (m.r2 >> 2);
    m.l8 = m.l4 & (m.l4 << 4);   m.r8 = m.r4 & (m.r4 >> 4);
    m.l16 = m.l8 & (m.l8 << 8);  m.r16 = m.r8 & (m.r8 >> 8);
    m.l32 = m.l16 & (m.l16 << 16); m.r32 = m.r16 & (m.r16 >> 16);
    return m;
}
__device__ __forceinline__ unsigned long long fill_runs(unsigned long long seed, const RunMasks &m)
{
    unsigned long long gL = seed, gR = seed;
    gL |= m.l1 & (gL << 1);    gR |= m.r1 & (gR >> 1);
    gL |= m.l2 & (gL << 2);    gR |= m.r2 & (gR >> 2);
    gL |= m.l4 & (gL << 4);    gR |= m.r4 & (gR >> 4);
    gL |= m.l8 & (gL << 8);    gR |= m.r8 & (gR >> 8);
    gL |= m.l16 & (gL << 16);  gR |= m.r16 & (gR >> 16);
    gL |= m.l32 & (gL << 32);  gR |= m.r32 & (gR >> 32);
    return gL | gR;
}

// What a launch of the hysteresis hands to the next one.  Large batches: nothing but FLAGS -- hflags + (k & 1) * tiles, 1 = "look at this
// tile again", set with plain stores and scanned by the next launch (a list would need a counter that every dirtied tile increments:
// same-address atomics retire at about 5 ns each on this chip, and natural images dirty three quarters of their tiles in the first
// launch: 0.7 ms of atomics).  The last launch but one fills a WORK QUEUE instead: CannyBuffers::hlist as a ring of `ring_mask + 1` slots
// (a power of two >= tiles; an entry is tile + 1, 0 = empty slot), de-duplicated by the flags of the other parity (1 = queued), with
// three counters kept 128 bytes apart in pass_count:
//   tail = entries pushed so far, head = tickets handed to consumers, done = entries completely processed.
// A tile is in the queue at most once at a time, so at most `tiles` slots are ever occupied.
constexpr int kQTail = 32, kQHead = 64, kQDone = 96;
// Every wait of the queue is bounded: a wave that waits longer than any correct run can make it (about a second) sets this bit in `tail`,
// which every other wait sees; the launch drains and the host reports AEJ_ERR_STATE (api.hip reads `tail` back as the queue statistic).
constexpr int kQPoison = 0x40000000, kQSpinLimit = 1 << 20, kQIdleLimit = 1 << 19;
constexpr int kChaseDepth = 32;
struct HystOut { int *flags; int *ring; int *tail; int ring_mask; };      // ring == nullptr: flags only
__device__ __forceinline__ HystOut hyst_out(const CannyBuffers &cb, int parity, bool ring, long long total_tiles, int ring_mask)
{
    HystOut q;
    q.flags = cb.hflags + (long long)parity * total_tiles;
    q.ring = ring ? cb.hlist : nullptr;
    q.tail = cb.pass_count + kQTail;
    q.ring_mask = ring_mask;
    return q;
}

// (wave-uniform address: the first lane's value, so that what depends on it is scalar control flow)
__device__ __forceinline__ int ld_agent(const int *p) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ int ld_agent_lane(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Tile nT has a new pixel next to it: flag it for the next launch, or queue it unless it is queued already.  Called by single lanes; the
// caller has drained the stores / atomics that made the tile dirty.
template <bool RING>
__device__ __forceinline__ void hyst_push(const HystOut &q, long long nT)
{
    if constexpr (!RING) {
        q.flags[nT] = 1;
    } else {
        if (atomicExch(&q.flags[nT], 1) != 0) return;
        const int t = atomicAdd(q.tail, 1) & ~kQPoison;
        int *slot = &q.ring[t & q.ring_mask];
        // (second lap of the ring: the slot of ticket t - ring size was handed out at least `tiles` entries ago, its consumer has long
        // emptied it; wait if not -- bounded, so that a logic error, or the lapped-ring interlock ADVICE r4 describes, ends the call with
        // AEJ_ERR_STATE instead of hanging the device)
        if (t > q.ring_mask) {
            int spins = 0;
            while (ld_agent_lane(slot) != 0) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > kQSpinLimit) { atomicOr(q.tail, kQPoison); return; }
                if ((spins & 1023) == 0 && (ld_agent_lane(q.tail) & kQPoison)) return;      // somebody has given up: so does this push
            }
        }
        __hip_atomic_store(slot, (int)nT + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// One 64 x 64 tile to its fix-point for the halo it sees now.  ATOMIC: other waves may work on neighbouring tiles -- or on this one --
// at the same time (queue phase): strong words are read with agent-scope loads and written with atomicOr, and the writes are drained
// before the function returns, so that whoever is told about them afterwards sees them.  Strong bits only ever get set, so the
// fix-point does not depend on the interleaving.  Returns the set of in-bounds neighbours that now see a new pixel next to them
// (bit order: (-1,-1) (0,-1) (1,-1) (-1,0) (1,0) (-1,1) (0,1) (1,1)), 0 when nothing changed or T is not a tile.
struct HystTile { int b, tx, ty, ntx, tbase; };
template <bool ATOMIC>
__device__ __forceinline__ unsigned hyst_tile(const Geom &g, const CannyBuffers &cb, long long T, long long tiles_per_img, int lane, HystTile &ht)
{
    auto ldw = [](const unsigned long long *p) -> unsigned long long {
        if constexpr (ATOMIC) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return *p;
    };
    const int b = (int)(T / tiles_per_img);
    int l, tx, ty, ntx, nty, tbase;
    if (!locate_tile(g, kHystTile, kHystTile, (int)(T - (long long)b * tiles_per_img), l, tx, ty, ntx, nty, tbase)) return 0;
    ht.b = b; ht.tx = tx; ht.ty = ty; ht.ntx = ntx; ht.tbase = tbase;
    const int h = g.h[l], wpr = g.wpr[l];
    unsigned long long *sg = cb.strong + (long long)b * g.bpstride + g.bpoff[l];
    const unsigned long long *wk = cb.weak + (long long)b * g.bpstride + g.bpoff[l];
    const int y = ty * 64 + lane;
    const bool valid = y < h;
    // tile-major layout: this tile's 64 row-words are contiguous; the left / right tiles are +-64 words away.
    // All ten loads are unconditional (addresses clamped into the plane, results masked afterwards) so that they are in
    // flight together: a wave pays one memory latency per tile instead of one per halo piece.
    const int yc = valid ? y : h - 1;
    const int xl = tx > 0 ? tx - 1 : 0, xr = tx + 1 < wpr ? tx + 1 : wpr - 1;
    const int yt = ty > 0 ? ty * 64 - 1 : 0, yb = ty * 64 + 64 < h ? ty * 64 + 64 : h - 1;
    const long long o = bp_index(ty * 64, tx, wpr) + lane;
    unsigned long long S = ldw(&sg[bp_index(yc, tx, wpr)]), W = wk[bp_index(yc, tx, wpr)];
    unsigned long long SLw = ldw(&sg[bp_index(yc, xl, wpr)]), SRw = ldw(&sg[bp_index(yc, xr, wpr)]);
    unsigned long long Tm = ldw(&sg[bp_index(yt, tx, wpr)]), Tlw = ldw(&sg[bp_index(yt, xl, wpr)]), Trw = ldw(&sg[bp_index(yt, xr, wpr)]);
    unsigned long long Bm = ldw(&sg[bp_index(yb, tx, wpr)]), Blw = ldw(&sg[bp_index(yb, xl, wpr)]), Brw = ldw(&sg[bp_index(yb, xr, wpr)]);
    if (!valid) { S = 0; W = 0; SLw = 0; SRw = 0; }
    const bool hasL = tx > 0, hasR = tx + 1 < wpr, hasT = ty > 0, hasB = ty * 64 + 64 < h;
    unsigned long long SL = hasL ? SLw >> 63 : 0ull, SR = hasR ? SRw & 1ull : 0ull;
    unsigned long long Tl = (hasT && hasL) ? Tlw >> 63 : 0ull, Tr = (hasT && hasR) ? Trw & 1ull : 0ull;
    unsigned long long Bl = (hasB && hasL) ? Blw >> 63 : 0ull, Br = (hasB && hasR) ? Brw & 1ull : 0ull;
    if (!hasT) Tm = 0;
    if (!hasB) Bm = 0;
    W &= ~S;                             // candidates still to be decided
    if (!__any(W != 0ull)) return 0;     // no undecided candidate pixel in the tile: nothing can change here
    const unsigned long long S0 = S;
    // the rows above / below come from the neighbouring lanes by DPP whole-wave shifts (one vector instruction per 32 bits; lane 0 /
    // 63 keep the `old` operand = the halo row) instead of `__shfl` (ds_bpermute: an LDS round trip in the loop's dependence chain);
    // the left / right halo columns do not change inside the loop, so their part of the mask is formed once
    auto row_above = [&](unsigned long long v, unsigned long long halo) {
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)halo, (int)(unsigned)v, 0x138, 0xf, 0xf, false);          // wave_shr:1
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(halo >> 32), (int)(unsigned)(v >> 32), 0x138, 0xf, 0xf, false);
        return ((unsigned long long)hi << 32) | lo;
    };
    auto row_below = [&](unsigned long long v, unsigned long long halo) {
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)halo, (int)(unsigned)v, 0x130, 0xf, 0xf, false);          // wave_shl:1
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(halo >> 32), (int)(unsigned)(v >> 32), 0x130, 0xf, 0xf, false);
        return ((unsigned long long)hi << 32) | lo;
    };
    const unsigned long long edge = (SL | row_above(SL, Tl) | row_below(SL, Bl)) | ((SR | row_above(SR, Tr) | row_below(SR, Br)) << 63);
    // W keeps every candidate (decided ones included: a run may be entered through them), S grows
    const unsigned long long Wall = W | S;
    const RunMasks rm = run_masks(Wall);
    for (;;) {
        // one sweep: the 8-neighbourhood of everything strong seeds the rows' runs ...
        unsigned long long m = S | row_above(S, Tm) | row_below(S, Bm);
        unsigned long long nS = S | fill_runs(Wall & (m | (m << 1) | (m >> 1) | edge), rm);
        // ... and two cheap vertical steps follow (a new pixel promotes the three candidates below / above it: 4 DPP moves + 10 logic
        // operations against the ~90 of a sweep), so that contours that run down the rows advance three rows per sweep instead of one
        unsigned long long u = row_above(nS, Tm), d = row_below(nS, Bm);
        nS |= Wall & (u | (u << 1) | (u >> 1) | d | (d << 1) | (d >> 1));
        u = row_above(nS, Tm); d = row_below(nS, Bm);
        nS |= Wall & (u | (u << 1) | (u >> 1) | d | (d << 1) | (d >> 1));
        const bool ch = nS != S;
        S = nS;
        if (!__any(ch)) break;
    }
    const unsigned long long diff = S ^ S0;
    if (diff && valid) {
        if constexpr (ATOMIC) atomicOr(&sg[o], S);
        else sg[o] = S;
    }
    // ATOMIC: the caller may now process a neighbour itself and read, as that tile's halo, the very words just OR-ed, or tell another
    // wave about them through the queue.  Drain the atomics first (they execute at the L2 / memory side, which also serves the
    // agent-scope loads), so that neither can overtake them.
    if constexpr (ATOMIC) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // which of the 8 neighbours see a changed pixel next to them: new bits in the first / last valid row (and their end
    // columns for the diagonal neighbours), in the first / last column
    const int last_row = min(63, h - 1 - ty * 64);
    // (wave-uniform lane indices: v_readlane, not a ds_bpermute round trip)
    const unsigned dlo = (unsigned)diff, dhi = (unsigned)(diff >> 32);
    const unsigned long long dtop = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)dhi, 0) << 32) | (unsigned)__builtin_amdgcn_readlane((int)dlo, 0);
    const int last_row_u = __builtin_amdgcn_readfirstlane(last_row);
    const unsigned long long dbot = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)dhi, last_row_u) << 32) | (unsigned)__builtin_amdgcn_readlane((int)dlo, last_row_u);
    const bool dl = __any((diff & 1ull) != 0), dr = __any((diff >> 63) != 0);
    unsigned dirs = 0;
    if (dtop) dirs |= 2u | ((dtop & 1ull) ? 1u : 0u) | ((dtop >> 63) ? 4u : 0u);
    if (dbot) dirs |= 64u | ((dbot & 1ull) ? 32u : 0u) | ((dbot >> 63) ? 128u : 0u);
    if (dl) dirs |= 8u;
    if (dr) dirs |= 16u;
    unsigned inb = 0;
    if (ty > 0) inb |= 2u | (tx > 0 ? 1u : 0u) | (tx + 1 < ntx ? 4u : 0u);
    if (ty + 1 < nty) inb |= 64u | (tx > 0 ? 32u : 0u) | (tx + 1 < ntx ? 128u : 0u);
    if (tx > 0) inb |= 8u;
    if (tx + 1 < ntx) inb |= 16u;
    return dirs & inb;
}

__device__ __forceinline__ long long hyst_neighbour(const HystTile &ht, long long tiles_per_img, int d)
{
    const int ox = d == 0 || d == 3 || d == 5 ? -1 : (d == 1 || d == 6 ? 0 : 1), oy = d < 3 ? -1 : (d < 5 ? 0 : 1);
    return (long long)ht.b * tiles_per_img + ht.tbase + (long long)(ht.ty + oy) * ht.ntx + (ht.tx + ox);
}

// First launch: every tile once, plain loads and stores (a wave sees its neighbours as they were or as they become: whoever changes a
// border flags / queues the neighbour behind it, so nothing is lost).
template <bool RING>
__global__ __launch_bounds__(256) void k_hyst_pass0(Geom g, CannyBuffers cb, long long tiles_per_img, long long total_tiles, int ring_mask)
{
    const int lane = threadIdx.x & 63;
    const long long nwaves = (long long)gridDim.x * 4;
    const HystOut dst = hyst_out(cb, 0, RING, total_tiles, ring_mask);
    for (long long T = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); T < total_tiles; T += nwaves) {
        HystTile ht;
        const unsigned dirs = hyst_tile<false>(g, cb, T, tiles_per_img, lane, ht);
        if (dirs == 0) continue;
        if constexpr (RING) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tile's stores before the queue entries that announce them
        if (lane < 8 && ((dirs >> lane) & 1u)) hyst_push<RING>(dst, hyst_neighbour(ht, tiles_per_img, lane));
    }
}

// A tile and the contour behind it: of the neighbours the tile has dirtied the wave processes one itself straight away (up to
// kChaseDepth tiles in a row) and flags / queues the others, so a long thin contour costs a chain of tiles on one wave instead of a
// round trip per tile.  Two waves may meet in a tile: hyst_tile<true> writes with atomicOr and reads with agent-scope loads.
template <bool RING>
__device__ __forceinline__ void hyst_chase(const Geom &g, const CannyBuffers &cb, long long T, long long tiles_per_img, int lane, const HystOut &dst)
{
    for (int depth = 0;; depth++) {
        HystTile ht;
        const unsigned dirs = hyst_tile<true>(g, cb, T, tiles_per_img, lane, ht);
        if (dirs == 0) return;
        int chase = -1;
        // prefer a side neighbour over a corner: 4 (right), 3 (left), 6 (down), 1 (up), then the corners
        if (depth < kChaseDepth) chase = (dirs & 16u) ? 4 : (dirs & 8u) ? 3 : (dirs & 64u) ? 6 : (dirs & 2u) ? 1 : __ffs((int)dirs) - 1;
        if (lane < 8 && ((dirs >> lane) & 1u) && lane != chase) hyst_push<RING>(dst, hyst_neighbour(ht, tiles_per_img, lane));
        if (chase < 0) return;
        T = hyst_neighbour(ht, tiles_per_img, chase);
    }
}

// Bulk launches (large batches only): a wave owns kBulkGroup consecutive tiles, reads their flags of parity k - 1 (complete when the launch
// starts: the kernel boundary is the barrier -- no list, no counter, no polling), clears them and chases each flagged tile; what that
// dirties is flagged in the other parity (k < last) or queued for the drain (k == last, de-duplicated by the other parity's flags, all
// zero again by then).
constexpr int kBulkGroup = 4;
template <bool RING>
__global__ __launch_bounds__(256) void k_hyst_bulk(Geom g, CannyBuffers cb, long long tiles_per_img, long long total_tiles, int k, int ring_mask)
{
    const int lane = threadIdx.x & 63;
    int *src = cb.hflags + (long long)((k - 1) & 1) * total_tiles;
    const HystOut dst = hyst_out(cb, k & 1, RING, total_tiles, ring_mask);
    const long long base = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * kBulkGroup;
    if (base >= total_tiles) return;
    const bool mine = lane < kBulkGroup && base + lane < total_tiles;
    const int f = mine ? src[base + lane] : 0;
    unsigned m = (unsigned)__ballot(f != 0);
    if (m == 0) return;
    if (f) src[base + lane] = 0;
    while (m) {
        const int i = __ffs((int)m) - 1;
        m &= m - 1;
        hyst_chase<RING>(g, cb, base + i, tiles_per_img, lane, dst);
    }
}

// Last launch: the work queue is drained to the fix-point by however many waves the launch has: no pass structure, no barrier,
// nothing for the host to guess or to read back.  A wave claims a chunk of tickets (head), waits for the entries of those tickets,
// clears the tiles' "queued" flags BEFORE it reads the tiles (a change that arrives later finds the flag clear and queues the tile
// again), and chases each tile's contour, pushing to the same ring.
// Termination: entries are only ever pushed by a wave that is processing an entry, and `done` counts entries whose processing -- pushes
// included -- is complete; done == tail (done read first) therefore means that nothing is queued, nothing is running and nothing can be
// pushed any more.  Every wave reaches that state or an entry of its own: a wave never waits for another wave to be scheduled, only for
// a running one to finish its tile, so the launch needs no co-residency.  Waiting waves poll two counters: the launch is kept small
// (more waves measured slower: 64 x 4K hysteresis stage 0.48 / 0.53 / 0.66 / 0.83 ms with 256 / 512 / 1024 / 2048 workgroups draining
// everything the first launch dirtied), which is why large batches run two bulk launches first.
constexpr int kHystChunk = 4;          // tickets a wave claims at once: one claim, one round of slot reads and one round of flag clears per chunk
// (Measured and dropped: the first look at every tile folded into this launch as "virtual" tickets, one launch in all for a single image --
// 0.062 against 0.044 ms for the two launches: the first look then pays agent-scope loads and every wave of it polls at the end.)
__global__ __launch_bounds__(256) void k_hyst_drain(Geom g, CannyBuffers cb, long long tiles_per_img, long long total_tiles, int parity, int ring_mask)
{
    // tickets per claim: latency-sized problems (a few images) have a short queue and plenty of waves: one entry per wave at a time
    const int K = total_tiles <= 8192 ? 1 : kHystChunk;
    const int lane = threadIdx.x & 63;
    const HystOut q = hyst_out(cb, parity, true, total_tiles, ring_mask);
    int *head = cb.pass_count + kQHead, *done_p = cb.pass_count + kQDone;
    for (;;) {
        int my = 0;
        if (lane == 0) my = atomicAdd(head, K);
        my = __builtin_amdgcn_readfirstlane(my);
        int next = my;                         // first ticket of the chunk not served yet
        int idle = 0;                          // (every wait of the queue is bounded: kQPoison)
        while (next - (my + K) < 0) {
            int tail = ld_agent(q.tail);
            if (tail & kQPoison) return;
            const int avail = (tail - (my + K) < 0 ? tail : my + K) - next;
            if (avail <= 0) {
                const int done = ld_agent(done_p);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // `done` is read before `tail`
                tail = ld_agent(q.tail);
                if (done == tail || (tail & kQPoison)) return;             // quiescent: the rest of this chunk will never be served (or: given up)
                if (tail - next <= 0) __builtin_amdgcn_s_sleep(100);
                if (++idle > kQIdleLimit) { if (lane == 0) atomicOr(q.tail, kQPoison); return; }
                continue;
            }
            // lanes 0 .. avail-1 fetch one entry each: the slot (written, or about to be: its producer holds the ticket), then the tile's
            // "queued" flag is cleared BEFORE the tile is read
            int e = 0;
            if (lane < avail) {
                int *slot = &q.ring[(next + lane) & ring_mask];
                int spins = 0;
                while ((e = ld_agent_lane(slot)) == 0) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > kQSpinLimit) { atomicOr(q.tail, kQPoison); break; }
                }
                if (e) {
                    __hip_atomic_store(slot, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    (void)atomicExch(&q.flags[e - 1], 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // flags cleared (performed at the L2) before any of the tiles is read
            for (int i = 0; i < avail; i++) {
                const int ei = __shfl(e, i);
                if (ei) hyst_chase<true>(g, cb, (long long)ei - 1, tiles_per_img, lane, q);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // these entries' pushes (tail tickets, slots) before they count as done
            if (lane == 0) atomicAdd(done_p, avail);
            next += avail;
        }
    }
}

// bit-plane -> uint8 expansions (stand-alone EdgeDetection.canny output and the stage dump for the tests)
__global__ __launch_bounds__(256) void k_bits_to_edge(Geom g, const unsigned long long *__restrict__ strong, unsigned char *__restrict__ edge)
{
    const int l = blockIdx.y, b = blockIdx.z;
    const int w = g.w[l], h = g.h[l], wpr = g.wpr[l];
    const unsigned long long *sg = strong + (long long)b * g.bpstride + g.bpoff[l];
    unsigned char *out = edge + (long long)b * g.pstride + g.poff[l];
    const long long n = (long long)w * h;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        int y = (int)(i / w), x = (int)(i - (long long)y * w);
        out[i] = (unsigned char)((sg[bp_index(y, x >> 6, wpr)] >> (x & 63)) & 1ull);
    }
}

__global__ __launch_bounds__(256) void k_bits_to_map(Geom g, const unsigned long long *__restrict__ weak, const unsigned long long *__restrict__ strong,
                                                     unsigned char *__restrict__ map)
{
    const int l = blockIdx.y, b = blockIdx.z;
    const int w = g.w[l], h = g.h[l], wpr = g.wpr[l];
    const unsigned long long *sg = strong + (long long)b * g.bpstride + g.bpoff[l];
    const unsigned long long *wk = weak + (long long)b * g.bpstride + g.bpoff[l];
    unsigned char *out = map + (long long)b * g.pstride + g.poff[l];
    const long long n = (long long)w * h;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        int y = (int)(i / w), x = (int)(i - (long long)y * w);
        long long o = bp_index(y, x >> 6, wpr);
        int s = (int)((sg[o] >> (x & 63)) & 1ull), k = (int)((wk[o] >> (x & 63)) & 1ull);
        out[i] = (unsigned char)(s ? 2 : k ? 0 : 1);
    }
}

// uint8 edge image (non-zero == edge) -> bit-plane (stand-alone QuadTree entry)
__global__ __launch_bounds__(256) void k_pack_edge_bits(Geom g, const unsigned char *__restrict__ edge, unsigned long long *__restrict__ bits)
{
    const int l = blockIdx.y, b = blockIdx.z;
    const int w = g.w[l], h = g.h[l], wpr = g.wpr[l];
    const unsigned char *src = edge + (long long)b * g.pstride + g.poff[l];
    unsigned long long *out = bits + (long long)b * g.bpstride + g.bpoff[l];
    const long long nwords = (long long)h * wpr;
    const int lane = threadIdx.x & 63;
    for (long long wd = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); wd < nwords; wd += (long long)gridDim.x * 4) {
        int y = (int)(wd / wpr), xw = (int)(wd - (long long)y * wpr);
        int x = xw * 64 + lane;
        bool e = x < w && src[(long long)y * w + x] != 0;
        unsigned long long m = __ballot(e);
        if (lane == 0) out[bp_index(y, xw, wpr)] = m;
    }
}

// zero-fill as a kernel: the graph-replay path of aej_encode_batch holds kernel nodes only (the runtime's memset / memcpy
// graph nodes proved unsafe to replay once other copies had run in between, see api.hip encode_graph)
__global__ __launch_bounds__(256) void k_zero16(uint4 *__restrict__ p, long long n16)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

void launch_zero(hipStream_t st, void *p, size_t bytes)      // p 16-byte aligned, bytes a multiple of 16
{
    const long long n16 = (long long)(bytes / 16);
    if (n16 <= 0) return;
    long long blocks = (n16 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_zero16, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<uint4 *>(p), n16);
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
    hipLaunchKernelGGL(k_clahe_lut, dim3(16, g.nl, g.B), dim3(256), 0, st, g, cb.tile_hist, cb.lut, cb.clip_limit);
}

// strips per image for a strip length; the longest strip that still gives every CU a few workgroups is used
static long long strips_per_image(const Geom &g, int TW, int TH, int strip)
{
    long long t = 0;
    for (int l = 0; l < g.nl; l++) {
        int ntx = (g.w[l] + TW - 1) / TW, nty = (g.h[l] + TH - 1) / TH;
        t += (long long)((ntx + strip - 1) / strip) * nty;
    }
    return t;
}
static int pick_strip(const Geom &g, int TW, int TH, int strip_max, long long want_workgroups)
{
    int strip = strip_max;
    while (strip > 1 && strips_per_image(g, TW, TH, strip) * g.B < want_workgroups) strip = (strip + 1) >> 1;
    return strip;
}

void launch_clahe_blur(hipStream_t st, const Geom &g, const CannyBuffers &cb)
{
    // Strip length: the one of 5 .. 15 tiles that cuts the tile rows of ALL layers into the most even pieces (fewest unused tile slots in
    // the rows' last strips; ties go to the longer strip), then halved while the launch would not fill the chip.  Round 3 used 8 throughout;
    // a 4K plane row is 30 (chroma 15) tiles -- 8 + 8 + 8 + 6 and 8 + 7 -- and strips of 15 (or 5) run the stage 4-7 % faster alone (2.02-2.07
    // against 2.14-2.22 ms; 6: 2.08, 10: 2.22, 4: 2.28), blocking 64 x 4K calls 0.1-0.2 ms faster; 1080p (15 / 8 tiles) keeps its 8
    // (15: 0.626 against 0.605 ms).  profiles/r04_ab_blur_strip.txt
    int best = 8, best_waste = 1 << 30;
    for (int s = 5; s <= kBStripMax; s++) {
        int waste = 0;
        for (int l = 0; l < g.nl; l++) {
            const int ntx = (g.w[l] + kBTW - 1) / kBTW;
            waste += (ntx + s - 1) / s * s - ntx;
        }
        if (waste <= best_waste) { best_waste = waste; best = s; }
    }
    const int strip = pick_strip(g, kBTW, kBTH, best, 1024);             // 256 CUs x 2 resident workgroups x 2
    if (cb.dump_clahe || cb.dump_gauss)
        hipLaunchKernelGGL(k_clahe_blur<true>, dim3((unsigned)(strips_per_image(g, kBTW, kBTH, strip) * g.B)), dim3(kBT), 0, st, g, cb, strip);
    else
        hipLaunchKernelGGL(k_clahe_blur<false>, dim3((unsigned)(strips_per_image(g, kBTW, kBTH, strip) * g.B)), dim3(kBT), 0, st, g, cb, strip);
}

void launch_thresholds(hipStream_t st, const Geom &g, const CannyBuffers &cb)
{
    hipLaunchKernelGGL(k_thresholds, dim3(g.nl, g.B), dim3(256), 0, st, g, cb.blur_hist, cb.thr, cb.low_q, cb.high_q, cb.l2);
}

void launch_sobel_nms(hipStream_t st, const Geom &g, const CannyBuffers &cb, const Tuning &tn)
{
    // register kernel: every layer's rows must be whole aligned dwords (w % 4 == 0, w >= 4); other shapes take the LDS kernel
    // (aej_set_option "sobel_lds": the LDS-tiled kernel of rounds 1-2 for every shape)
    bool reg_ok = !tn.sobel_lds;
    for (int l = 0; l < g.nl; l++) reg_ok = reg_ok && (g.w[l] % 4) == 0 && g.w[l] >= 4;
    if (reg_ok) {
        const long long t = hyst_tiles_per_image(g), total = t * g.B;
        const dim3 rgrid((unsigned)((total + 3) / 4));
        const int xcd = tn.sobel_xcd;
        if (cb.l2) hipLaunchKernelGGL(k_sobel_nms_reg<true>, rgrid, dim3(256), 0, st, g, cb, t, total, xcd);
        else hipLaunchKernelGGL(k_sobel_nms_reg<false>, rgrid, dim3(256), 0, st, g, cb, t, total, xcd);
        return;
    }
    const int strip = pick_strip(g, kBlurTW, kBlurTH, kStripMax, 2048);
    const dim3 grid((unsigned)(strips_per_image(g, kBlurTW, kBlurTH, strip) * g.B));
    if (cb.l2) hipLaunchKernelGGL(k_sobel_nms<true>, grid, dim3(256), 0, st, g, cb, strip);
    else hipLaunchKernelGGL(k_sobel_nms<false>, grid, dim3(256), 0, st, g, cb, strip);
}

// The launches of the hysteresis: every tile once; for large batches two bulk launches over the tiles flagged by the launch before
// (synthetic bench batch: a tenth of the tiles, then a fifth of that; natural images: three quarters, then a quarter); then the rest
// drained to the fix-point on the device by one small persistent launch.
constexpr int kHystDrainWgs = 64;

int hyst_bulk_launches(const Geom &g)
{
    const long long total = hyst_tiles_per_image(g) * g.B;
    return total > 32768 ? 2 : total > 8192 ? 1 : 0;
}
void launch_hysteresis(hipStream_t st, const Geom &g, const CannyBuffers &cb)
{
    const long long t = hyst_tiles_per_image(g), total = t * g.B;
    if (total <= 0) return;
    const int nbulk = hyst_bulk_launches(g), ring_mask = hyst_ring_slots(g) - 1;
    long long blocks = (total + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    if (nbulk == 0) hipLaunchKernelGGL(k_hyst_pass0<true>, dim3((unsigned)blocks), dim3(256), 0, st, g, cb, t, total, ring_mask);
    else hipLaunchKernelGGL(k_hyst_pass0<false>, dim3((unsigned)blocks), dim3(256), 0, st, g, cb, t, total, ring_mask);
    const unsigned bulk_blocks = (unsigned)((total + 4 * kBulkGroup - 1) / (4 * kBulkGroup));
    for (int k = 1; k <= nbulk; k++) {
        if (k == nbulk) hipLaunchKernelGGL(k_hyst_bulk<true>, dim3(bulk_blocks), dim3(256), 0, st, g, cb, t, total, k, ring_mask);
        else hipLaunchKernelGGL(k_hyst_bulk<false>, dim3(bulk_blocks), dim3(256), 0, st, g, cb, t, total, k, ring_mask);
    }
    // consumers of the queue: a wave per 16 tiles (per 4 for latency-sized problems), at most 4 x kHystDrainWgs waves
    long long drain = total <= 8192 ? (total + 15) / 16 : (total + 63) / 64;
    if (drain > kHystDrainWgs) drain = kHystDrainWgs;
    hipLaunchKernelGGL(k_hyst_drain, dim3((unsigned)drain), dim3(256), 0, st, g, cb, t, total, nbulk & 1, ring_mask);
}

static int expand_blocks(const Geom &g)
{
    long long n = 0;
    for (int l = 0; l < g.nl; l++) { long long m = (long long)g.w[l] * g.h[l]; if (m > n) n = m; }
    long long b = (n + 255) / 256;
    return (int)(b > 4096 ? 4096 : b < 1 ? 1 : b);
}

void launch_bits_to_edge(hipStream_t st, const Geom &g, const unsigned long long *strong, unsigned char *edge01)
{
    hipLaunchKernelGGL(k_bits_to_edge, dim3(expand_blocks(g), g.nl, g.B), dim3(256), 0, st, g, strong, edge01);
}

void launch_bits_to_map(hipStream_t st, const Geom &g, const unsigned long long *weak, const unsigned long long *strong, unsigned char *map)
{
    hipLaunchKernelGGL(k_bits_to_map, dim3(expand_blocks(g), g.nl, g.B), dim3(256), 0, st, g, weak, strong, map);
}

void launch_pack_edge_bits(hipStream_t st, const Geom &g, const unsigned char *edge, unsigned long long *bits)
{
    hipLaunchKernelGGL(k_pack_edge_bits, dim3(expand_blocks(g), g.nl, g.B), dim3(256), 0, st, g, edge, bits);
}

}  // namespace aej
