// color.hip -- colour conversion (a-1), chroma down-sampling (a-2), uint8 scaling (a-3) and the CLAHE
// tile histograms, fused into one pass over the float32 RGB input.  gfx950 only.
//
// Replaces, for the encode path:  color.convert("sRGB", space, x)  (src/color/conversion.py:95-124 and the
// per-space files), Jpeg._downsample (src/jpeg/jpeg.py:323-338), apply_normalization
// (src/color/common.py:161-174 via jpeg.py:387-390) and (img*255).astype(uint8) (src/jpeg/edge_detection.py:70).
//
// Numerics contract: every float op below is a single IEEE-754 operation in a fixed order
// (compiled with -ffp-contract=off; fma only where written), so results are bit-identical to the CPU oracle.
#include "aej_common.h"
#include "aej_launch.h"
#include "aej_devmath.h"
#include <stdlib.h>

namespace aej {

__device__ __forceinline__ void to_xyz(float lr, float lg, float lb, float &X, float &Y, float &Z)   // xyz.py:27-32, 63-64 after linearisation
{
    X = dot3(F(0.4124564), F(0.3575761), F(0.1804375), lr, lg, lb);
    Y = dot3(F(0.2126729), F(0.7151522), F(0.0721750), lr, lg, lb);
    Z = dot3(F(0.0193339), F(0.1191920), F(0.9503041), lr, lg, lb);
}

// colour transform from LINEAR rgb (spaces >= 3: everything after common.py:34-60's sRGB linearisation) or from sRGB
// itself (the three matrix spaces)
template <int SPACE>
__device__ __forceinline__ void color_px_lin(float r, float g, float b, float &o0, float &o1, float &o2, const PowTabs &pt)
{
    if constexpr (SPACE == 0) {          // YCbCr, ycbcr.py:25-30, 61
        o0 = dot3(F(0.299000), F(0.587000), F(0.114000), r, g, b);
        o1 = dot3(F(-0.168736), F(-0.331264), F(0.500000), r, g, b);
        o2 = dot3(F(0.500000), F(-0.418688), F(-0.081312), r, g, b);
    } else if constexpr (SPACE == 1) {   // YCoCg, ycocg.py:25-30, 82
        o0 = dot3(F(0.25), F(0.50), F(0.25), r, g, b);
        o1 = dot3(F(0.50), F(0.00), F(-0.50), r, g, b);
        o2 = dot3(F(-0.25), F(0.50), F(-0.25), r, g, b);
    } else if constexpr (SPACE == 2) {   // YCoCg-R, ycocg.py:46-51, 121
        o0 = dot3(F(0.25), F(0.50), F(0.25), r, g, b);
        o1 = dot3(F(1.00), F(0.00), F(-1.00), r, g, b);
        o2 = dot3(F(-0.50), F(1.00), F(-0.50), r, g, b);
    } else if constexpr (SPACE == 3) {   // OKLAB, oklab.py:27-44, 71-75
        float X, Y, Z;
        to_xyz(r, g, b, X, Y, Z);
        float l = dot3(F(0.8189330101), F(0.3618667424), F(-0.1288597137), X, Y, Z);
        float m = dot3(F(0.0329845436), F(0.9293118715), F(0.0361456387), X, Y, Z);
        float s = dot3(F(0.0482003018), F(0.2643662691), F(0.6338517070), X, Y, Z);
        float lp = dev_pow_third_f32(l, pt), mp = dev_pow_third_f32(m, pt), sp = dev_pow_third_f32(s, pt);      // np.power(f32, 1 / 3)
        o0 = dot3(F(0.2104542553), F(0.7936177850), F(-0.0040720468), lp, mp, sp);
        o1 = dot3(F(1.9779984951), F(-2.4285922050), F(0.4505937099), lp, mp, sp);
        o2 = dot3(F(0.0259040371), F(0.7827717662), F(-0.8086757660), lp, mp, sp);
    } else if constexpr (SPACE == 4 || SPACE == 5) {   // ICtCp ictcp.py:45-81,142-157 / ICaCb icacb.py:45-81,142-157
        float X, Y, Z;
        to_xyz(r, g, b, X, Y, Z);
        float L, M, S;
        if constexpr (SPACE == 4) {
            L = lin3(F(0.3592), F(0.6976), F(-0.0358), X, Y, Z);
            M = lin3(F(-0.1922), F(1.1004), F(0.0755), X, Y, Z);
            S = lin3(F(0.0070), F(0.0749), F(0.8434), X, Y, Z);
        } else {
            L = lin3(F(0.37613), F(0.70431), F(-0.05675), X, Y, Z);
            M = lin3(F(-0.21649), F(1.14744), F(0.05356), X, Y, Z);
            S = lin3(F(0.02567), F(0.16713), F(0.74235), X, Y, Z);
        }
        const double pm2 = 2523.0 / 32.0;
        double Lp = pq_inverse_eotf((double)L, pm2, pt), Mp = pq_inverse_eotf((double)M, pm2, pt), Sp = pq_inverse_eotf((double)S, pm2, pt);
        if constexpr (SPACE == 4) {
            o0 = (float)lin3d(F(0.5000), F(0.5000), F(0.0000), Lp, Mp, Sp);
            o1 = (float)lin3d(F(1.6137), F(-3.3234), F(1.7097), Lp, Mp, Sp);
            o2 = (float)lin3d(F(4.3781), F(-4.2455), F(-0.1325), Lp, Mp, Sp);
        } else {
            o0 = (float)lin3d(F(0.4949), F(0.5037), F(0.0015), Lp, Mp, Sp);
            o1 = (float)lin3d(F(4.2854), F(-4.5462), F(0.2609), Lp, Mp, Sp);
            o2 = (float)lin3d(F(0.3605), F(1.1499), F(-1.5105), Lp, Mp, Sp);
        }
    } else if constexpr (SPACE == 7) {    // XYZ itself, xyz.py:63-64 (helper space of color.convert; not a codec space)
        to_xyz(r, g, b, o0, o1, o2);
    } else {                              // JzAzBz, jzazbz.py:54-99, 178-206
        float X, Y, Z;
        to_xyz(r, g, b, X, Y, Z);
        const double bb = 1.15, gg = 0.66, d = -0.56, d0 = 1.6295499532821566e-11, p = 1.7 * 2523.0 / 32.0;
        double Xp = bb * (double)X - (bb - 1.0) * (double)Z;
        double Yp = gg * (double)Y - (gg - 1.0) * (double)X;
        // numba typing: M[i,2] * Z_p is float32*float32 -> float32; the other products are float64
        double L = ((double)F(0.41478972) * Xp + (double)F(0.579999) * Yp) + (double)(F(0.0146480) * Z);
        double M = ((double)F(-0.2015100) * Xp + (double)F(1.120649) * Yp) + (double)(F(0.0531008) * Z);
        double S = ((double)F(-0.0166008) * Xp + (double)F(0.264800) * Yp) + (double)(F(0.6684799) * Z);
        double Lp = pq_inverse_eotf(L, p, pt), Mp = pq_inverse_eotf(M, p, pt), Sp = pq_inverse_eotf(S, p, pt);
        double Iz = lin3d(F(0.500000), F(0.500000), F(0.000000), Lp, Mp, Sp);
        double Az = lin3d(F(3.524000), F(-4.066708), F(0.542708), Lp, Mp, Sp);
        double Bz = lin3d(F(0.199076), F(1.096799), F(-1.295875), Lp, Mp, Sp);
        double Jz = ((1.0 + d) * Iz) / (1.0 + d * Iz) - d0;
        o0 = (float)Jz; o1 = (float)Az; o2 = (float)Bz;
    }
}

template <int SPACE>
__device__ __forceinline__ void color_px(float r, float g, float b, float &o0, float &o1, float &o2)
{
    const PowTabs pt = pow_tabs_global();
    if constexpr (SPACE >= 3) color_px_lin<SPACE>(srgb_to_linear(r, pt), srgb_to_linear(g, pt), srgb_to_linear(b, pt), o0, o1, o2, pt);
    else color_px_lin<SPACE>(r, g, b, o0, o1, o2, pt);
}

// a-3: (v*255).astype(uint8): float32 multiply, truncate toward zero, keep the low byte
__device__ __forceinline__ unsigned char scale_u8(float v)
{
    float s = v * 255.0f;
    int t = (int)s;
    return (unsigned char)(t & 0xFF);
}

// ------------------------------------------------------------------------------------------------
// stand-alone colour conversion: [n][3] -> [n][3]
// ------------------------------------------------------------------------------------------------
template <int SPACE>
__global__ __launch_bounds__(256) void k_color_convert(const float *__restrict__ rgb, float *__restrict__ out, long long n)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
        float o0, o1, o2;
        color_px<SPACE>(r, g, b, o0, o1, o2);
        out[3 * i] = o0; out[3 * i + 1] = o1; out[3 * i + 2] = o2;
    }
}

// ------------------------------------------------------------------------------------------------
// fused: RGB -> 3 layers (normalised float32 + uint8 + optional raw float32) + CLAHE tile histograms.
// One thread = a 4 (wide) x 2 (tall) pixel patch; a 256-thread block = 128 x 16 pixels.
// Layout: lanes run along x, so a wave reads 32 x 48 B = 1.5 KiB contiguous bytes per image row.
// ------------------------------------------------------------------------------------------------
struct NormConst { float mid[3]; float scale[3]; };

// The LDS histogram is addressed through an explicit address-space-3 pointer: through a generic `int *` the compiler emits FLAT
// atomics (flat_atomic_add: both memory pipes, both wait counters) instead of ds_add_u32.
typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ void hist_add(lds_int *lds_hist, int *__restrict__ ghist, int layer, int tx0, int ty0, int tx, int ty, int v)
{
    int dx = tx - tx0, dy = ty - ty0;
    if ((unsigned)dx < 2u && (unsigned)dy < 2u) __hip_atomic_fetch_add(&lds_hist[((layer * 4) + dy * 2 + dx) * 256 + v], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else atomicAdd(&ghist[((layer * 16) + ty * 4 + tx) * 256 + v], 1);
}

// IN = float: [B][H][W][3] float32 in [0, 1].  IN = unsigned char: the same image as uint8; the kernel forms
// float32(v) / 255.0f per channel (image.py:80, `imread(path).astype(np.float32) / 255.0`) from a 256-entry LDS table
// of exactly those quotients, so both ingest paths produce identical planes for images that came from 8-bit files.
template <int SPACE, int RH, int RW, typename IN>
__global__ __launch_bounds__(256) void k_color_planes(const IN *__restrict__ rgb, Geom g, NormConst nc,
                                                      float *__restrict__ planes_raw, float *__restrict__ planes_norm,
                                                      unsigned char *__restrict__ planes_u8, int *__restrict__ tile_hist)
{
    __shared__ int s_hist[3 * 4 * 256];
    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int px = (blockIdx.x * 32 + (tid & 31)) * 4;
    const int py = (blockIdx.y * 8 + (tid >> 5)) * 2;
    const bool do_hist = tile_hist != nullptr;
    constexpr bool kU8 = sizeof(IN) == 1;
    // Spaces with an sRGB linearisation (a float64 pow per channel): images that came from 8-bit files hold only the 256
    // values k / 255.0f, so the block tabulates srgb_to_linear of exactly those -- computed by the same device function -- and a
    // pixel row whose 12 inputs all are such values (checked by comparing with the table of inputs) skips its 12 pows.
    constexpr bool kLin = SPACE >= 3;
    __shared__ float s_u8f[(kU8 || kLin) ? 256 : 1];
    __shared__ float s_lin[kLin ? 256 : 1];
    __shared__ double s_pow[kLin ? 192 : 1];       // LDS copy of the pow tables
    PowTabs pt = pow_tabs_global();
    if constexpr (kLin) {
        if (tid < 192) s_pow[tid] = tid < 64 ? POW_INVC[tid] : tid < 128 ? POW_LOGC[tid - 64] : POW_EXP2T[tid - 128];
        __syncthreads();
        pt = PowTabs{ s_pow, s_pow + 64, s_pow + 128 };
    }
    if (kU8 || kLin) s_u8f[tid] = (float)tid / 255.0f;
    if (kLin) s_lin[tid] = srgb_to_linear((float)tid / 255.0f, pt);
    if (do_hist) {
        for (int i = tid; i < 3 * 4 * 256; i += 256) s_hist[i] = 0;
    }
    if (do_hist || kU8 || kLin) __syncthreads();
    // first CLAHE tile touched by this block, per layer
    const int bx0 = blockIdx.x * 128, by0 = blockIdx.y * 16;
    const int tx0_l = bx0 / g.ctw[0], ty0_l = by0 / g.cth[0];
    const int tx0_c = (bx0 / RW) / g.ctw[1], ty0_c = (by0 / RH) / g.cth[1];
    // A CLAHE tile is a quarter of the layer, so for all but tiny images the 128 x 16 block meets at most one tile boundary per
    // axis and the tile of a pixel is a comparison; `wide` is uniform over the launch.  (An integer division per histogram
    // update made this kernel VALU-bound.)
    const bool wide = g.ctw[0] >= 128 && g.cth[0] >= 16 && g.ctw[1] * RW >= 128 && g.cth[1] * RH >= 16;
    const int xb_l = (tx0_l + 1) * g.ctw[0], yb_l = (ty0_l + 1) * g.cth[0];
    const int xb_c = (tx0_c + 1) * g.ctw[1], yb_c = (ty0_c + 1) * g.cth[1];
    auto tile_lx = [&](int x) { return wide ? tx0_l + (x >= xb_l ? 1 : 0) : x / g.ctw[0]; };
    auto tile_ly = [&](int y) { return wide ? ty0_l + (y >= yb_l ? 1 : 0) : y / g.cth[0]; };
    auto tile_cx = [&](int x) { return wide ? tx0_c + (x >= xb_c ? 1 : 0) : x / g.ctw[1]; };
    auto tile_cy = [&](int y) { return wide ? ty0_c + (y >= yb_c ? 1 : 0) : y / g.cth[1]; };
    int *ghist = do_hist ? tile_hist + (long long)b * 3 * 16 * 256 : nullptr;

    if (px < g.W && py < g.H) {
        float c0[2][4], c1[2][4], c2[2][4];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            float in[12];
            if constexpr (kU8) {      // 4 pixels = 12 bytes = 3 aligned dwords (px % 4 == 0, W % 4 == 0)
                const unsigned int *p = reinterpret_cast<const unsigned int *>(rgb + (((long long)b * g.H + (py + r)) * g.W + px) * 3);
                const unsigned int d[3] = { p[0], p[1], p[2] };
#pragma unroll
                for (int k = 0; k < 12; k++) in[k] = (kLin ? s_lin : s_u8f)[(d[k >> 2] >> (8 * (k & 3))) & 0xffu];
            } else {
                const float4 *p = reinterpret_cast<const float4 *>(rgb + (((long long)b * g.H + (py + r)) * g.W + px) * 3);
                float4 a = p[0], bq = p[1], c = p[2];
                in[0] = a.x; in[1] = a.y; in[2] = a.z; in[3] = a.w; in[4] = bq.x; in[5] = bq.y; in[6] = bq.z; in[7] = bq.w;
                in[8] = c.x; in[9] = c.y; in[10] = c.z; in[11] = c.w;
            }
            if constexpr (kLin && !kU8) {
                float lin[12];
                bool hit = true;
#pragma unroll
                for (int k = 0; k < 12; k++) {
                    int idx = __float2int_rn(in[k] * 255.0f);
                    idx = idx < 0 ? 0 : idx > 255 ? 255 : idx;
                    hit = hit && (s_u8f[idx] == in[k]);
                    lin[k] = s_lin[idx];
                }
                if (!__all(hit)) {        // some value of this wave's rows is not k / 255.0f: those lanes take the float64 pow
                    if (!hit) {
#pragma unroll
                        for (int k = 0; k < 12; k++) lin[k] = srgb_to_linear(in[k], pt);
                    }
                }
#pragma unroll
                for (int k = 0; k < 12; k++) in[k] = lin[k];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) color_px_lin<SPACE>(in[3 * k], in[3 * k + 1], in[3 * k + 2], c0[r][k], c1[r][k], c2[r][k], pt);
        }
        const long long ibase = (long long)b * g.pstride;
        // ---- layer 0 (luma): ratio 1x1 => copy
#pragma unroll
        for (int r = 0; r < 2; r++) {
            long long o = ibase + g.poff[0] + (long long)(py + r) * g.w[0] + px;
            float4 nv;
            nv.x = (c0[r][0] - nc.mid[0]) * nc.scale[0];
            nv.y = (c0[r][1] - nc.mid[0]) * nc.scale[0];
            nv.z = (c0[r][2] - nc.mid[0]) * nc.scale[0];
            nv.w = (c0[r][3] - nc.mid[0]) * nc.scale[0];
            if (planes_norm) *reinterpret_cast<float4 *>(planes_norm + o) = nv;
            if (planes_raw) *reinterpret_cast<float4 *>(planes_raw + o) = make_float4(c0[r][0], c0[r][1], c0[r][2], c0[r][3]);
            uchar4 u;
            u.x = scale_u8(c0[r][0]); u.y = scale_u8(c0[r][1]); u.z = scale_u8(c0[r][2]); u.w = scale_u8(c0[r][3]);
            if (planes_u8) *reinterpret_cast<uchar4 *>(planes_u8 + o) = u;
            if (do_hist) {
                int ty = tile_ly(py + r);
                hist_add((lds_int *)s_hist, ghist, 0, tx0_l, ty0_l, tile_lx(px + 0), ty, u.x);
                hist_add((lds_int *)s_hist, ghist, 0, tx0_l, ty0_l, tile_lx(px + 1), ty, u.y);
                hist_add((lds_int *)s_hist, ghist, 0, tx0_l, ty0_l, tile_lx(px + 2), ty, u.z);
                hist_add((lds_int *)s_hist, ghist, 0, tx0_l, ty0_l, tile_lx(px + 3), ty, u.w);
            }
        }
        // ---- layers 1, 2 (chroma): INTER_AREA box mean
#pragma unroll
        for (int ch = 1; ch < 3; ch++) {
            float(*cc)[4] = ch == 1 ? c1 : c2;
            float v[2];
            int cx, cy[2];
            if constexpr (RH == 2 && RW == 2) {      // ((r0e+r0o)+(r1e+r1o))*0.25f
                v[0] = ((cc[0][0] + cc[0][1]) + (cc[1][0] + cc[1][1])) * 0.25f;
                v[1] = ((cc[0][2] + cc[0][3]) + (cc[1][2] + cc[1][3])) * 0.25f;
                cx = px / 2; cy[0] = cy[1] = py / 2;
            } else {                                  // RH == 1, RW == 4: sequential sum * (1/4)
                v[0] = (((cc[0][0] + cc[0][1]) + cc[0][2]) + cc[0][3]) * 0.25f;
                v[1] = (((cc[1][0] + cc[1][1]) + cc[1][2]) + cc[1][3]) * 0.25f;
                cx = px / 4; cy[0] = py; cy[1] = py + 1;
            }
            if constexpr (RH == 2 && RW == 2) {      // the two outputs are x-neighbours: paired stores (cx is even)
                long long o = ibase + g.poff[ch] + (long long)cy[0] * g.w[ch] + cx;
                if (planes_norm)
                    *reinterpret_cast<float2 *>(planes_norm + o) =
                        make_float2((v[0] - nc.mid[ch]) * nc.scale[ch], (v[1] - nc.mid[ch]) * nc.scale[ch]);
                if (planes_raw) *reinterpret_cast<float2 *>(planes_raw + o) = make_float2(v[0], v[1]);
                uchar2 u;
                u.x = scale_u8(v[0]); u.y = scale_u8(v[1]);
                if (planes_u8) *reinterpret_cast<uchar2 *>(planes_u8 + o) = u;
                if (do_hist) {
                    const int tyc = tile_cy(cy[0]);          // layers 1 and 2 share their geometry
                    hist_add((lds_int *)s_hist, ghist, ch, tx0_c, ty0_c, tile_cx(cx), tyc, u.x);
                    hist_add((lds_int *)s_hist, ghist, ch, tx0_c, ty0_c, tile_cx(cx + 1), tyc, u.y);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    int y = cy[q];
                    long long o = ibase + g.poff[ch] + (long long)y * g.w[ch] + cx;
                    if (planes_norm) planes_norm[o] = (v[q] - nc.mid[ch]) * nc.scale[ch];
                    if (planes_raw) planes_raw[o] = v[q];
                    unsigned char u = scale_u8(v[q]);
                    if (planes_u8) planes_u8[o] = u;
                    if (do_hist) hist_add((lds_int *)s_hist, ghist, ch, tx0_c, ty0_c, tile_cx(cx), tile_cy(y), u);
                }
            }
        }
    }
    if (do_hist) {
        __syncthreads();
        for (int i = tid; i < 3 * 4 * 256; i += 256) {
            int c = s_hist[i];
            if (c) {
                int layer = i / 1024, t = (i >> 8) & 3, v = i & 255;
                int tx = (layer == 0 ? tx0_l : tx0_c) + (t & 1), ty = (layer == 0 ? ty0_l : ty0_c) + (t >> 1);
                if (tx < 4 && ty < 4) atomicAdd(&ghist[((layer * 16) + ty * 4 + tx) * 256 + v], c);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Strip version of the fused plane kernel (round 3): the shapes every BASELINE configuration has (W % 16 == 0, H % 8 == 0, so the
// CLAHE tiles are W/4 x H/4 without padding and a 4 x 2 patch never straddles a tile edge in any layer).  A workgroup owns a
// 128-pixel-wide column (128-aligned, so every row segment it writes is whole 128-byte lines) and walks down `rows` rows of ONE
// CLAHE tile row, 16 rows per iteration, so that
//   * the histogram is zeroed and flushed once per strip instead of once per 2048 pixels (the 128 x 16 kernel above spends 24 LDS
//     accesses and up to 12 global atomics per thread on that), and needs two slots of 3 x 256 counters only: a 128-pixel column
//     meets at most one vertical tile edge, a thread's columns lie on one side of it for the whole strip, so its slot is a constant
//     LDS base -- no tile test per update; columns that meet no edge use the two slots as lane-striped copies (odd stride: equal
//     values of neighbouring pixels land in different banks);
//   * the kernel's footprint is small on purpose -- 6 KiB of LDS, at most 56 VGPRs -- so that one of its workgroups fits on a CU
//     beside three resident workgroups of the blur kernel (3 x 152 VGPRs per SIMD, 3 x 39.5 KiB; with the tiled planes' staging this kernel takes 38 KiB of what is left): the HBM-bound stage of one chain
//     then really shares SIMDs with the issue-bound stage of another.
// Per-pixel arithmetic is the same sequence of single IEEE operations as k_color_planes.
// ------------------------------------------------------------------------------------------------
constexpr int kStripSlots = 2;
constexpr int kStripHistStride = 257;

// PROD: the whole-path call (normalised + uint8 planes and the histograms wanted, no raw planes): the per-output null tests, which
// otherwise are a scalar compare + branch per store, fold away.
//
// Thread -> pixels: lanes 0..31 of a 32-lane half-wave own the 4-pixel columns of the 128-pixel strip, the 8 half-waves own 8 bands
// of rows / 8 consecutive rows; a thread walks down its band TWO rows per step and keeps the next row's 48 bytes in flight while it
// converts the current one (row B is requested before row A is touched, the next row A as soon as row A's registers are free), so
// a wave always has loads outstanding -- the kernel is meant to stream at few waves per CU.
template <typename IN> struct RowRaw;
template <> struct RowRaw<float> { float4 a, b, c; };
template <> struct RowRaw<unsigned char> { unsigned int d[3]; };

template <typename IN>
__device__ __forceinline__ RowRaw<IN> strip_load_row(const IN *img, unsigned byte_off)
{
    RowRaw<IN> r;
    if constexpr (sizeof(IN) == 1) {
        const unsigned int *p = reinterpret_cast<const unsigned int *>(reinterpret_cast<const char *>(img) + byte_off);
        r.d[0] = p[0]; r.d[1] = p[1]; r.d[2] = p[2];
    } else {
        const float4 *p = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(img) + byte_off);
        r.a = p[0]; r.b = p[1]; r.c = p[2];
    }
    return r;
}

template <int SPACE, int RH, int RW, typename IN, bool PROD, bool TILED>
__global__ __launch_bounds__(256) void k_color_planes_strip(const IN *__restrict__ rgb, Geom g, NormConst nc, float *__restrict__ planes_raw,
                                                            float *__restrict__ planes_norm, unsigned char *__restrict__ planes_u8,
                                                            int *__restrict__ tile_hist, int nxb, int nys, int rows, int nstrips)
{
    __shared__ int s_hist[3 * kStripSlots * kStripHistStride];
    extern __shared__ __attribute__((aligned(16))) float s_stage[];      // Geom::tiled only: 8 half-waves x 1024 floats (luma 4 x 128, chroma 2 x 4 x 64)
    const int tid = threadIdx.x;
    constexpr bool kU8 = sizeof(IN) == 1;
    constexpr bool kLin = SPACE >= 3;
    __shared__ float s_u8f[(kU8 || kLin) ? 256 : 1];
    __shared__ float s_lin[kLin ? 256 : 1];
    __shared__ double s_pow[kLin ? 192 : 1];
    PowTabs pt = pow_tabs_global();
    if constexpr (kLin) {
        if (tid < 192) s_pow[tid] = tid < 64 ? POW_INVC[tid] : tid < 128 ? POW_LOGC[tid - 64] : POW_EXP2T[tid - 128];
        __syncthreads();
        pt = PowTabs{ s_pow, s_pow + 64, s_pow + 128 };
    }
    if (kU8 || kLin) s_u8f[tid] = (float)tid / 255.0f;
    if (kLin) s_lin[tid] = srgb_to_linear((float)tid / 255.0f, pt);
    const bool do_hist = PROD || tile_hist != nullptr;
    const bool has_norm = PROD || planes_norm != nullptr, has_raw = !PROD && planes_raw != nullptr, has_u8 = PROD || planes_u8 != nullptr;
    if (kU8 || kLin) __syncthreads();
    constexpr int kLayerInts = kStripSlots * kStripHistStride;
    const long long poff1 = g.poff[1], poff2 = g.poff[2];
    const unsigned in_row_bytes = (unsigned)g.W * 3u * (unsigned)sizeof(IN);

    // one row of four pixels: colour transform, luma outputs, histogram; the chroma values come back to the caller
    // (RH = RW = 2: the horizontal pair sums (c[0] + c[1], c[2] + c[3]) of both chroma channels; RH = 1, RW = 4: the row's finished
    // means in h1[0] / h2[0])
    auto do_row = [&](const RowRaw<IN> &raw, float (&h1)[2], float (&h2)[2], float *norm0, float *raw0, unsigned char *u80, unsigned o, unsigned on, lds_int *hcopy) {
        float in[12];
        if constexpr (kU8) {
#pragma unroll
            for (int k = 0; k < 12; k++) in[k] = (kLin ? s_lin : s_u8f)[(raw.d[k >> 2] >> (8 * (k & 3))) & 0xffu];
        } else {
            in[0] = raw.a.x; in[1] = raw.a.y; in[2] = raw.a.z; in[3] = raw.a.w; in[4] = raw.b.x; in[5] = raw.b.y; in[6] = raw.b.z; in[7] = raw.b.w;
            in[8] = raw.c.x; in[9] = raw.c.y; in[10] = raw.c.z; in[11] = raw.c.w;
        }
        if constexpr (kLin && !kU8) {
            float lin[12];
            bool hit = true;
#pragma unroll
            for (int k = 0; k < 12; k++) {
                int idx = __float2int_rn(in[k] * 255.0f);
                idx = idx < 0 ? 0 : idx > 255 ? 255 : idx;
                hit = hit && (s_u8f[idx] == in[k]);
                lin[k] = s_lin[idx];
            }
            if (!__all(hit)) {        // some value of this wave's rows is not k / 255.0f: those lanes take the float64 pow
                if (!hit) {
#pragma unroll
                    for (int k = 0; k < 12; k++) lin[k] = srgb_to_linear(in[k], pt);
                }
            }
#pragma unroll
            for (int k = 0; k < 12; k++) in[k] = lin[k];
        }
        float c0[4], c1[4], c2[4];
#pragma unroll
        for (int k = 0; k < 4; k++) color_px_lin<SPACE>(in[3 * k], in[3 * k + 1], in[3 * k + 2], c0[k], c1[k], c2[k], pt);
        if constexpr (RH == 2 && RW == 2) {
            h1[0] = c1[0] + c1[1]; h1[1] = c1[2] + c1[3];
            h2[0] = c2[0] + c2[1]; h2[1] = c2[2] + c2[3];
        } else {
            h1[0] = (((c1[0] + c1[1]) + c1[2]) + c1[3]) * 0.25f; h1[1] = 0.f;
            h2[0] = (((c2[0] + c2[1]) + c2[2]) + c2[3]) * 0.25f; h2[1] = 0.f;
        }
        if (has_norm) {
            const float4 nv = make_float4((c0[0] - nc.mid[0]) * nc.scale[0], (c0[1] - nc.mid[0]) * nc.scale[0], (c0[2] - nc.mid[0]) * nc.scale[0], (c0[3] - nc.mid[0]) * nc.scale[0]);
            if (TILED) *reinterpret_cast<float4 *>(s_stage + on) = nv;          // (staged: whole 4 x 4 blocks leave together, see the row loop)
            else *reinterpret_cast<float4 *>(reinterpret_cast<char *>(norm0) + 4u * on) = nv;
        }
        if (has_raw) *reinterpret_cast<float4 *>(reinterpret_cast<char *>(raw0) + 4u * o) = make_float4(c0[0], c0[1], c0[2], c0[3]);
        uchar4 u;
        u.x = scale_u8(c0[0]); u.y = scale_u8(c0[1]); u.z = scale_u8(c0[2]); u.w = scale_u8(c0[3]);
        if (has_u8) *reinterpret_cast<uchar4 *>(u80 + o) = u;
        if (do_hist) {
            __hip_atomic_fetch_add(&hcopy[u.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&hcopy[u.y], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&hcopy[u.z], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&hcopy[u.w], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };

    // Persistent over strips: the launch holds as many workgroups as the caller wants resident (one or two per CU when the kernel is
    // to run in the gaps beside another chain's issue-bound kernels), each takes strips blockIdx.x, + gridDim.x, ... .  Strips are
    // numbered x fastest, then tile row / strip row, then image, so that the workgroups in flight sweep the input linearly.
    const int per_img = nxb * nys * 4;
    for (int strip = blockIdx.x; strip < nstrips; strip += gridDim.x) {
        const int b = strip / per_img;
        const int rem = strip - b * per_img;
        const int yb_i = rem / nxb, xb_i = rem - yb_i * nxb;
        if (do_hist) {
            if (strip != (int)blockIdx.x) __syncthreads();          // the previous strip's flush has read the counters
            // (fixed trip count from one address register: the kernel has to stay within 56 VGPRs)
            constexpr int kHistInts = 3 * kStripSlots * kStripHistStride;
#pragma unroll
            for (int k = 0; k < kHistInts / 256; k++) s_hist[k * 256 + tid] = 0;
            if (tid < kHistInts % 256) s_hist[(kHistInts / 256) * 256 + tid] = 0;
            __syncthreads();
        }
        // the strip's rectangle: columns [x0, x0 + 128), rows [y0, yend) inside CLAHE tile row ty (the same tile indices hold in
        // the chroma layers: their tiles are the luma tiles divided by the down-sampling ratios)
        const int ty = yb_i / nys, ys = yb_i - ty * nys;
        const int x0 = xb_i * 128;
        const int y0 = ty * g.cth[0] + ys * rows, yend = min(y0 + rows, (ty + 1) * g.cth[0]);
        const int tx0 = x0 / g.ctw[0];                 // CLAHE tile column of the strip's first pixel
        const int xedge = (tx0 + 1) * g.ctw[0];        // the one vertical tile edge the strip can meet
        const bool straddle = xedge < x0 + 128 && xedge < g.W;
        const IN *img = rgb + (long long)b * g.H * g.W * 3;
        const long long ibase = (long long)b * g.pstride;
        {
            // (per-thread values are re-derived per strip from a copy of the thread id the compiler cannot see through: hoisted out of
            // the strip loop they cost ten registers, and the kernel has to stay within 56)
            int tq = tid;
            asm volatile("" : "+v"(tq));
            const int px = x0 + 4 * (tq & 31);
            const int band = rows >> 3;                 // rows per half-wave: even (rows is a multiple of 16)
            const int ya = y0 + (tq >> 5) * band, yb = min(ya + band, yend);
            const int slot = straddle ? (px >= xedge ? 1 : 0) : (tq & 1);
            lds_int *hcopy = (lds_int *)s_hist + slot * kStripHistStride;
            if (px < g.W && ya < yb) {
                float *norm0 = planes_norm + ibase + g.poff[0], *raw0 = planes_raw + ibase + g.poff[0];
                unsigned char *u80 = planes_u8 + ibase + g.poff[0];
                float *normc[2] = { planes_norm + ibase + poff1, planes_norm + ibase + poff2 }, *rawc[2] = { planes_raw + ibase + poff1, planes_raw + ibase + poff2 };
                unsigned char *u8c[2] = { planes_u8 + ibase + poff1, planes_u8 + ibase + poff2 };
                unsigned ioff = (unsigned)(ya * g.W + px) * (3u * (unsigned)sizeof(IN));      // byte offset inside the image: < 2^32
                unsigned o_l = (unsigned)(ya * g.w[0] + px);
                unsigned o_c = RH == 2 ? (unsigned)((ya >> 1) * g.w[1] + (px >> 1)) : (unsigned)(ya * g.w[1] + (px >> 2));
                // element offsets of the NORMALISED planes (what the DCT kernels read): the same as o_l / o_c when those planes are row-major;
                // Geom::tiled: 4 x 4 blocks (plane_elem) -- ya is a multiple of 4, a lane's four luma pixels are one row of a block, and the
                // four rows of a block are written by consecutive iterations of the same lane
                // With Geom::tiled the rows are STAGED in LDS, four at a time per half-wave, and leave as whole blocks: a half-wave's 4 x 128
                // luma pixels are 32 blocks = 2 KiB contiguous in the plane (chroma: 4 x 64 = 1 KiB, or 4 x 32 = 512 B), written as 16-byte
                // pieces by consecutive lanes.  In LDS, 16-byte unit u of row r sits at u ^ 4r (rows of 8 units: u ^ 4 (r >> 1)), which
                // makes the row-wise writes and the block-wise reads conflict-free without padding.  n_l / n_c = this lane's slot in the
                // current block row's run.
                const unsigned st0 = (unsigned)(tq >> 5) * 1024u;          // this half-wave's staging area (floats)
                unsigned n_l = TILED ? (unsigned)plane_elem(1, g.w[0], ya, x0) + 4u * (unsigned)(tq & 31) : o_l;
                unsigned n_c = TILED ? (unsigned)(RH == 2 ? plane_elem(1, g.w[1], (ya >> 1) & ~3, x0 >> 1) : plane_elem(1, g.w[1], ya, x0 >> 2)) + 4u * (unsigned)(tq & 31) : o_c;
                int c_lo = (ya >> 1) & 3;          // (RH == 2) first row of the current chroma block this half-wave produces: 0, or 2 when the band starts
                                                   // in the middle of a block (CLAHE tiles 4 (mod 8) rows high) -- the other rows are another half-wave's
                // Rows in flight.  float32 input: a row is 48 bytes per lane, two rows ahead is what 56 registers allow, and at one workgroup
                // per CU that is 24 KiB in flight per CU -- enough to stream 6.4 GB in 2.1 ms.  8-bit input is 12 bytes per lane: the same
                // two rows are 6 KiB per CU, and the kernel took the same 2.1 ms for a quarter of the bytes (round 4: "the colour stage is no
                // longer HBM-bound and nobody looked at why" -- it is bound by bytes in flight, Little's law).  So the 8-bit instantiation
                // requests its whole band -- up to eight rows, 24 registers -- before it converts the first one.
                RowRaw<IN> rowA = strip_load_row<IN>(img, ioff);
                auto row_pair = [&](const RowRaw<IN> &rowB, int y) {
                    float a1[2], a2[2], b1[2], b2[2];
                    const int rb = y & 3;                 // row of the block (0 or 2: ya is a multiple of 4)
                    // (the lane index once more from a copy the compiler cannot see through: hoisted out of the row loop, the dozen LDS addresses
                    // derived from it cost two dozen registers, and the kernel has to stay within 56)
                    int tl = tq;
                    asm volatile("" : "+v"(tl));
                    const int lane32 = tl & 31;
                    do_row(rowA, a1, a2, norm0, raw0, u80, o_l, TILED ? st0 + (unsigned)(rb * 128 + ((lane32 ^ (4 * rb)) << 2)) : n_l, hcopy);
                    // the next row A (the band's last step re-reads row B instead: an unconditional load keeps the registers of
                    // rowA out of a copy at the loop's back edge)
                    if constexpr (!kU8) rowA = strip_load_row<IN>(img, ioff + (y + 2 < yb ? 2u : 1u) * in_row_bytes);
                    do_row(rowB, b1, b2, norm0, raw0, u80, o_l + (unsigned)g.w[0],
                           TILED ? st0 + (unsigned)((rb + 1) * 128 + ((lane32 ^ (4 * (rb + 1))) << 2)) : n_l + (unsigned)g.w[0], hcopy);
                    // ---- layers 1, 2 (chroma): INTER_AREA box mean
#pragma unroll
                    for (int ch = 1; ch < 3; ch++) {
                        const float *ca = ch == 1 ? a1 : a2, *cb = ch == 1 ? b1 : b2;
                        float v[2];
                        if constexpr (RH == 2 && RW == 2) {      // ((r0e+r0o)+(r1e+r1o))*0.25f
                            v[0] = (ca[0] + cb[0]) * 0.25f;
                            v[1] = (ca[1] + cb[1]) * 0.25f;
                            if (has_norm) {
                                const float2 nv = make_float2((v[0] - nc.mid[ch]) * nc.scale[ch], (v[1] - nc.mid[ch]) * nc.scale[ch]);
                                if (TILED) {
                                    const int rc = (y >> 1) & 3;
                                    *reinterpret_cast<float2 *>(s_stage + st0 + 512u + (unsigned)((ch - 1) * 256 + rc * 64 + (((lane32 >> 1) ^ (4 * rc)) << 2) + (lane32 & 1) * 2)) = nv;
                                } else {
                                    *reinterpret_cast<float2 *>(reinterpret_cast<char *>(normc[ch - 1]) + 4u * n_c) = nv;
                                }
                            }
                            if (has_raw) *reinterpret_cast<float2 *>(reinterpret_cast<char *>(rawc[ch - 1]) + 4u * o_c) = make_float2(v[0], v[1]);
                            uchar2 u;
                            u.x = scale_u8(v[0]); u.y = scale_u8(v[1]);
                            if (has_u8) *reinterpret_cast<uchar2 *>(u8c[ch - 1] + o_c) = u;
                            if (do_hist) {
                                __hip_atomic_fetch_add(&hcopy[ch * kLayerInts + u.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_fetch_add(&hcopy[ch * kLayerInts + u.y], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        } else {                                  // RH == 1, RW == 4: sequential sum * (1/4)
                            v[0] = ca[0];
                            v[1] = cb[0];
#pragma unroll
                            for (int q = 0; q < 2; q++) {
                                const unsigned o = o_c + (unsigned)(q * g.w[ch]);
                                if (has_norm) {
                                    const float nv = (v[q] - nc.mid[ch]) * nc.scale[ch];
                                    if (TILED) {
                                        const int rr = (y & 3) + q;
                                        s_stage[st0 + 512u + (unsigned)((ch - 1) * 256 + rr * 32 + (((lane32 >> 2) ^ ((rr >> 1) << 2)) << 2) + (lane32 & 3))] = nv;
                                    } else {
                                        *reinterpret_cast<float *>(reinterpret_cast<char *>(normc[ch - 1]) + 4u * (n_c + (unsigned)(q * g.w[ch]))) = nv;
                                    }
                                }
                                if (has_raw) *reinterpret_cast<float *>(reinterpret_cast<char *>(rawc[ch - 1]) + 4u * o) = v[q];
                                const unsigned char u = scale_u8(v[q]);
                                if (has_u8) u8c[ch - 1][o] = u;
                                if (do_hist) __hip_atomic_fetch_add(&hcopy[ch * kLayerInts + u], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    }
                    ioff += 2u * in_row_bytes;
                    o_l += 2u * (unsigned)g.w[0];
                    o_c += RH == 2 ? (unsigned)g.w[1] : 2u * (unsigned)g.w[1];
                    if (TILED) {
                        // (LDS operations of one wave execute in order: the reads below see the writes above, the next rows' writes come after)
                        const int fr = lane32 & 3, fb = lane32 >> 2;          // this lane's row / block in a flush
                        if (y & 2) {                              // rows 2, 3 of the luma blocks are in: 32 blocks = 2 KiB leave
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                *reinterpret_cast<float4 *>(reinterpret_cast<char *>(norm0) + 4u * (n_l + 128u * k)) =
                                    *reinterpret_cast<const float4 *>(s_stage + st0 + (unsigned)(fr * 128 + (((fb + 8 * k) ^ (4 * fr)) << 2)));
                                __builtin_amdgcn_sched_barrier(0);       // (one piece in registers at a time: the kernel has to stay within 56 VGPRs)
                            }
                            n_l += 4u * (unsigned)g.w[0];         // the next block row: w / 4 blocks of 16 elements on
                        }
                        if (RH == 2) {
                            const int rc = (y >> 1) & 3;
                            if (rc == 3 || y + 2 >= yb) {         // the block's last row, or the band's: rows c_lo .. rc of 16 blocks per plane leave
                                if (fr >= c_lo && fr <= rc) {
#pragma unroll
                                    for (int ch = 1; ch < 3; ch++)
#pragma unroll
                                        for (int k = 0; k < 2; k++) {
                                            *reinterpret_cast<float4 *>(reinterpret_cast<char *>(normc[ch - 1]) + 4u * (n_c + 128u * k)) =
                                                *reinterpret_cast<const float4 *>(s_stage + st0 + 512u + (unsigned)((ch - 1) * 256 + fr * 64 + (((fb + 8 * k) ^ (4 * fr)) << 2)));
                                            __builtin_amdgcn_sched_barrier(0);
                                        }
                                }
                                n_c += 4u * (unsigned)g.w[1];
                                c_lo = 0;
                            }
                        } else if (y & 2) {                       // RH == 1, RW == 4: 8 blocks = 512 B per plane
#pragma unroll
                            for (int ch = 1; ch < 3; ch++) {
                                *reinterpret_cast<float4 *>(reinterpret_cast<char *>(normc[ch - 1]) + 4u * n_c) =
                                    *reinterpret_cast<const float4 *>(s_stage + st0 + 512u + (unsigned)((ch - 1) * 256 + fr * 32 + ((fb ^ ((fr >> 1) << 2)) << 2)));
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            n_c += 4u * (unsigned)g.w[1];
                        }
                    } else {
                        n_l = o_l;
                        n_c = o_c;
                    }
                };
                if constexpr (kU8) {
                    RowRaw<IN> pre[7];                    // rows ya + 1 .. ya + 7 (the band is at most eight rows: strips are at most 64 rows high)
#pragma unroll
                    for (int k = 0; k < 7; k++) pre[k] = strip_load_row<IN>(img, ioff + (unsigned)(ya + 1 + k < yb ? 1 + k : yb - ya - 1) * in_row_bytes);
                    __builtin_amdgcn_sched_barrier(0);       // (the loads are issued HERE, all of them, not sunk to their first use)
#pragma unroll
                    for (int sp = 0; sp < 4; sp++) {
                        const int y = ya + 2 * sp;
                        if (y < yb) {
                            row_pair(pre[2 * sp], y);
                            if (sp < 3) rowA = pre[2 * sp + 1];
                        }
                    }
                } else {
                    for (int y = ya; y < yb; y += 2) {
                        const RowRaw<IN> rowB = strip_load_row<IN>(img, ioff + in_row_bytes);
                        row_pair(rowB, y);
                    }
                }
            }
        }
        if (do_hist) {
            __syncthreads();
            int *ghist = tile_hist + (long long)b * 3 * 16 * 256 + (ty * 4 + tx0) * 256;
#pragma unroll
            for (int layer = 0; layer < 3; layer++) {
                const int c0 = s_hist[(layer * kStripSlots + 0) * kStripHistStride + tid], c1 = s_hist[(layer * kStripSlots + 1) * kStripHistStride + tid];
                if (straddle) {
                    if (c0) atomicAdd(&ghist[layer * 16 * 256 + tid], c0);
                    if (c1) atomicAdd(&ghist[layer * 16 * 256 + 256 + tid], c1);
                } else if (c0 + c1) {
                    atomicAdd(&ghist[layer * 16 * 256 + tid], c0 + c1);
                }
            }
        }
    }       // strips
}

// ------------------------------------------------------------------------------------------------
// generic (slow-path) version of the fused plane kernel for shapes the 4x2-patch kernel cannot take: widths that are
// not a multiple of 4, odd heights, and sizes that do not divide by the down-sampling ratios, where cv.resize(INTER_AREA)
// is the general area-weighted resize (OpenCV resize.cpp computeResizeAreaTab + ResizeArea_Invoker):
//   per destination pixel: sum = 0; for each vertical tap: buf = 0; for each horizontal tap: buf += S * alpha; sum += beta * buf
// One thread per destination pixel of a layer; source pixels are colour-converted on the fly.
// ------------------------------------------------------------------------------------------------
template <int SPACE, typename IN>
__global__ __launch_bounds__(256) void k_color_planes_generic(const IN *__restrict__ rgb, Geom g, NormConst nc, AreaTabs tabs,
                                                              float *__restrict__ planes_raw, float *__restrict__ planes_norm,
                                                              unsigned char *__restrict__ planes_u8, int *__restrict__ tile_hist)
{
    const int l = blockIdx.y, b = blockIdx.z;
    const int w = g.w[l], h = g.h[l];
    const long long n = (long long)w * h;
    const IN *img = rgb + (long long)b * g.H * g.W * 3;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int y = (int)(i / w), x = (int)(i - (long long)y * w);
        float v;
        auto chan = [&](int sy, int sx) {
            const IN *p = img + ((long long)sy * g.W + sx) * 3;
            float o0, o1, o2;
            if constexpr (sizeof(IN) == 1) color_px<SPACE>((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, o0, o1, o2);
            else color_px<SPACE>(p[0], p[1], p[2], o0, o1, o2);
            return l == 0 ? o0 : l == 1 ? o1 : o2;
        };
        if (h == g.H && w == g.W) {
            v = chan(y, x);                                              // dsize == ssize: copy
        } else if (tabs.mode == 0) {                                     // exact 2x2: ((r0e + r0o) + (r1e + r1o)) * 0.25f
            v = ((chan(2 * y, 2 * x) + chan(2 * y, 2 * x + 1)) + (chan(2 * y + 1, 2 * x) + chan(2 * y + 1, 2 * x + 1))) * 0.25f;
        } else if (tabs.mode == 1) {                                     // other integer areas: sequential sum * (1 / area)
            float sum = 0.f;
            for (int dy = 0; dy < tabs.isy; dy++)
                for (int dx = 0; dx < tabs.isx; dx++) sum += chan(y * tabs.isy + dy, x * tabs.isx + dx);
            v = sum * (1.f / (float)(tabs.isx * tabs.isy));
        } else {                                                         // general area tables
            float sum = 0.f;
            for (int j = tabs.yoff[y]; j < tabs.yoff[y + 1]; j++) {
                float buf = 0.f;
                for (int k = tabs.xoff[x]; k < tabs.xoff[x + 1]; k++) { float t = chan(tabs.ysi[j], tabs.xsi[k]) * tabs.xal[k]; buf = buf + t; }
                float t = tabs.yal[j] * buf;
                sum = sum + t;
            }
            v = sum;
        }
        const long long o = (long long)b * g.pstride + g.poff[l] + i;
        if (planes_raw) planes_raw[o] = v;
        if (planes_norm) planes_norm[o] = (v - nc.mid[l]) * nc.scale[l];
        const unsigned char u = scale_u8(v);
        if (planes_u8) planes_u8[o] = u;
        if (tile_hist) atomicAdd(&tile_hist[(((long long)b * 3 + l) * 16 + (y / g.cth[l]) * 4 + (x / g.ctw[l])) * 256 + u], 1);
    }
}

// stand-alone a-3 for one float32 plane (EdgeDetection.canny entry): uint8 + CLAHE tile histograms
__global__ __launch_bounds__(256) void k_plane_u8(const float *__restrict__ plane, Geom g, unsigned char *__restrict__ u8,
                                                  int *__restrict__ tile_hist)
{
    long long n = (long long)g.h[0] * g.w[0];
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned char v = scale_u8(plane[i]);
        u8[i] = v;
        int y = (int)(i / g.w[0]), x = (int)(i - (long long)y * g.w[0]);
        atomicAdd(&tile_hist[((y / g.cth[0]) * 4 + (x / g.ctw[0])) * 256 + v], 1);
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
template <int SPACE>
static void launch_convert_t(hipStream_t st, const float *rgb, float *out, long long n)
{
    int blocks = (int)((n + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_color_convert<SPACE>, dim3(blocks), dim3(256), 0, st, rgb, out, n);
}

int launch_color_convert(hipStream_t st, int space, const float *rgb, float *out, long long n)
{
    switch (space) {
    case 0: launch_convert_t<0>(st, rgb, out, n); break;
    case 1: launch_convert_t<1>(st, rgb, out, n); break;
    case 2: launch_convert_t<2>(st, rgb, out, n); break;
    case 3: launch_convert_t<3>(st, rgb, out, n); break;
    case 4: launch_convert_t<4>(st, rgb, out, n); break;
    case 5: launch_convert_t<5>(st, rgb, out, n); break;
    case 6: launch_convert_t<6>(st, rgb, out, n); break;
    case 7: launch_convert_t<7>(st, rgb, out, n); break;
    default: return -1;
    }
    return 0;
}

// Shapes the persistent strip kernel takes (CLAHE tiles without padding whose edges no 4 x 2 patch straddles, in every layer), and its strip
// height.  Tuning (aej_set_option): color_strip = 0 -> the 128 x 16 kernel of rounds 1-2, color_strip_rows = strip height.
static bool strip_shape(const Geom &g, const Tuning &t, int &nxb, int &rows)
{
    if (!((g.W % 16) == 0 && (g.H % 8) == 0 && g.ctw[0] * 4 == g.W && g.cth[0] * 4 == g.H && g.ctw[0] >= 128 && t.color_strip)) return false;
    nxb = (g.W + 127) / 128;
    // rows per workgroup: as long as the launch still has a few thousand workgroups (one image spreads over the chip), at most 64
    rows = 64;
    while (rows > 16 && (long long)nxb * ((g.cth[0] + rows - 1) / rows) * 4 * g.B < 4096) rows >>= 1;
    if (t.color_strip_rows > 0) rows = (t.color_strip_rows + 15) / 16 * 16;
    return true;
}

// Whether the encode path may keep the normalised planes of this geometry in 4 x 4 blocks (Geom::tiled, plane_elem): the strip kernel
// writes them (its half-waves start on rows that are multiples of 4 when the strip height is a multiple of 32 and the CLAHE tile
// height a multiple of 4) and every layer's sides are multiples of 4.
// On by default; the option planes_row_major keeps the planes row-major.  64 x 4K: `k_dct4` / `k_dct8_shfl` 0.285 / 0.225 -> 0.23 / 0.163 ms (every
// leaf is whole sectors; 1.35 GB less read per call), the colour stage 2.12 -> 2.15 ms.  (Written piecewise -- 16 bytes per lane at a 64-byte
// stride, four times the memory transactions of a row -- the colour stage took 2.5 ms: hence the staging in LDS.)
bool color_planes_can_tile(const Geom &g, int space, bool in_u8, const Tuning &t)
{
    // (the spaces with float64 pows -- OKLAB, ICtCp, ICaCb, JzAzBz -- are arithmetic-bound in the colour stage: the staging costs them 0.15-0.2 ms per
    // 8 x 8K / 16 x 4K, the small-block DCTs gain 0.03: row-major)
    if (space >= 3) return false;
    // (8-bit input: the colour stage reads a quarter of the bytes and is no longer hidden behind them -- the staging costs it 0.26 ms per 64 x 4K,
    // more than the small-block DCTs gain: row-major)
    if (in_u8) return false;
    int nxb, rows;
    if (t.planes_row_major || g.nl != 3 || !strip_shape(g, t, nxb, rows)) return false;
    // (every strip full: W a multiple of 128; half-waves start on multiples of 4 luma rows: strips of 32 or 64 rows, CLAHE tiles a multiple of 4 high)
    if ((rows % 32) != 0 || (g.cth[0] % 4) != 0 || (g.W % 128) != 0) return false;
    for (int l = 0; l < 3; l++)
        if ((g.w[l] % 4) != 0 || (g.h[l] % 4) != 0) return false;
    return true;
}

template <int SPACE, int RH, int RW>
static void launch_planes_t(hipStream_t st, const void *rgb, bool in_u8, const Geom &g, const NormConst &nc, float *raw, float *norm,
                            unsigned char *u8, int *hist, const Tuning &t)
{
    // strip kernel: CLAHE tiles without padding whose edges no 4 x 2 patch straddles, in every layer
    int nxb, rows;
    if (strip_shape(g, t, nxb, rows)) {
        const int nys = (g.cth[0] + rows - 1) / rows;
        const long long nstrips = (long long)nxb * nys * 4 * g.B;
        // Workgroups in the launch.  The matrix spaces are a pure stream: ONE workgroup per CU (256 in all, each walking ~270 strips)
        // is the fastest shape alone -- 64 x 4K: 1.95 ms against 2.27 ms with 8 per CU and 2.19 ms for the 128 x 16 kernel; 320
        // workgroups: 2.5 ms (a quarter of the CUs get two) -- and, being one small workgroup per CU, it runs in the gaps beside the
        // other chains' kernels (64 x 4K pipelined step 7.4 -> 7.0 ms; with 128 workgroups the kernel takes 3.0 ms alone and the step
        // is still 7.0: the stage is off the critical path).  The spaces with float64 pows are arithmetic-bound: fill the chip.
        long long want = SPACE < 3 ? 256 : 256 * 8;
        if (t.color_workgroups > 0) want = t.color_workgroups;
        dim3 sgrid((unsigned)(nstrips < want ? nstrips : want));
        const bool prod = norm && u8 && hist && !raw;
        const size_t stage_bytes = g.tiled ? 8 * 1024 * sizeof(float) : 0;
        auto go = [&](auto kern, auto *in) { hipLaunchKernelGGL(kern, sgrid, dim3(256), stage_bytes, st, in, g, nc, raw, norm, u8, hist, nxb, nys, rows, (int)nstrips); };
        // (the tiled form of the normalised planes only exists for the encode path's instantiation)
        if (in_u8) {
            if (prod && g.tiled) go(k_color_planes_strip<SPACE, RH, RW, unsigned char, true, true>, static_cast<const unsigned char *>(rgb));
            else if (prod) go(k_color_planes_strip<SPACE, RH, RW, unsigned char, true, false>, static_cast<const unsigned char *>(rgb));
            else go(k_color_planes_strip<SPACE, RH, RW, unsigned char, false, false>, static_cast<const unsigned char *>(rgb));
        } else {
            if (prod && g.tiled) go(k_color_planes_strip<SPACE, RH, RW, float, true, true>, static_cast<const float *>(rgb));
            else if (prod) go(k_color_planes_strip<SPACE, RH, RW, float, true, false>, static_cast<const float *>(rgb));
            else go(k_color_planes_strip<SPACE, RH, RW, float, false, false>, static_cast<const float *>(rgb));
        }
        return;
    }
    dim3 grid((g.W + 127) / 128, (g.H + 15) / 16, g.B);
    if (in_u8)
        hipLaunchKernelGGL((k_color_planes<SPACE, RH, RW, unsigned char>), grid, dim3(256), 0, st, static_cast<const unsigned char *>(rgb), g, nc,
                           raw, norm, u8, hist);
    else
        hipLaunchKernelGGL((k_color_planes<SPACE, RH, RW, float>), grid, dim3(256), 0, st, static_cast<const float *>(rgb), g, nc, raw, norm, u8,
                           hist);
}

int launch_color_planes(hipStream_t st, int space, const void *rgb, bool in_u8, const Geom &g, const float *mid, const float *scale,
                        float *raw, float *norm, unsigned char *u8, int *hist, const Tuning &t)
{
    if (g.tiled && (!color_planes_can_tile(g, space, in_u8, t) || !(norm && u8 && hist && !raw))) return -1;       // only the encode path's strip kernel writes the tiled form
    NormConst nc;
    for (int i = 0; i < 3; i++) { nc.mid[i] = mid[i]; nc.scale[i] = scale[i]; }
    switch (space) {
    case 0: launch_planes_t<0, 2, 2>(st, rgb, in_u8, g, nc, raw, norm, u8, hist, t); break;
    case 1: launch_planes_t<1, 2, 2>(st, rgb, in_u8, g, nc, raw, norm, u8, hist, t); break;
    case 2: launch_planes_t<2, 2, 2>(st, rgb, in_u8, g, nc, raw, norm, u8, hist, t); break;
    case 3: launch_planes_t<3, 2, 2>(st, rgb, in_u8, g, nc, raw, norm, u8, hist, t); break;
    case 4: launch_planes_t<4, 1, 4>(st, rgb, in_u8, g, nc, raw, norm, u8, hist, t); break;
    case 5: launch_planes_t<5, 1, 4>(st, rgb, in_u8, g, nc, raw, norm, u8, hist, t); break;
    case 6: launch_planes_t<6, 2, 2>(st, rgb, in_u8, g, nc, raw, norm, u8, hist, t); break;
    default: return -1;
    }
    return 0;
}

template <int SPACE>
static void launch_generic_t(hipStream_t st, const void *rgb, bool in_u8, const Geom &g, const NormConst &nc, const AreaTabs &tabs, float *raw,
                             float *norm, unsigned char *u8, int *hist)
{
    long long n = (long long)g.W * g.H;
    int bx = (int)((n + 255) / 256);
    if (bx > 2048) bx = 2048;
    if (in_u8)
        hipLaunchKernelGGL((k_color_planes_generic<SPACE, unsigned char>), dim3(bx, 3, g.B), dim3(256), 0, st, static_cast<const unsigned char *>(rgb),
                           g, nc, tabs, raw, norm, u8, hist);
    else
        hipLaunchKernelGGL((k_color_planes_generic<SPACE, float>), dim3(bx, 3, g.B), dim3(256), 0, st, static_cast<const float *>(rgb), g, nc, tabs,
                           raw, norm, u8, hist);
}

int launch_color_planes_generic(hipStream_t st, int space, const void *rgb, bool in_u8, const Geom &g, const float *mid, const float *scale,
                                const AreaTabs &tabs, float *raw, float *norm, unsigned char *u8, int *hist)
{
    NormConst nc;
    for (int i = 0; i < 3; i++) { nc.mid[i] = mid[i]; nc.scale[i] = scale[i]; }
    switch (space) {
    case 0: launch_generic_t<0>(st, rgb, in_u8, g, nc, tabs, raw, norm, u8, hist); break;
    case 1: launch_generic_t<1>(st, rgb, in_u8, g, nc, tabs, raw, norm, u8, hist); break;
    case 2: launch_generic_t<2>(st, rgb, in_u8, g, nc, tabs, raw, norm, u8, hist); break;
    case 3: launch_generic_t<3>(st, rgb, in_u8, g, nc, tabs, raw, norm, u8, hist); break;
    case 4: launch_generic_t<4>(st, rgb, in_u8, g, nc, tabs, raw, norm, u8, hist); break;
    case 5: launch_generic_t<5>(st, rgb, in_u8, g, nc, tabs, raw, norm, u8, hist); break;
    case 6: launch_generic_t<6>(st, rgb, in_u8, g, nc, tabs, raw, norm, u8, hist); break;
    default: return -1;
    }
    return 0;
}

void launch_plane_u8(hipStream_t st, const float *plane, const Geom &g, unsigned char *u8, int *hist)
{
    long long n = (long long)g.h[0] * g.w[0];
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_plane_u8, dim3(blocks), dim3(256), 0, st, plane, g, u8, hist);
}

}  // namespace aej
