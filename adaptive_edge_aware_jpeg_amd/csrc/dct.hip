// dct.hip -- per-leaf gather (+ np.pad reflect), 2-D DCT-II, quantisation and zigzag scatter
// (src/jpeg/jpeg.py:393-404, 471, 499-502, 579-588) for gfx950.
//
// Contract (DESIGN.md "DCT"): with D[k][n] = float32(alpha_k cos(pi (2n+1) k / 2s)),
//     T = D.X      T[i][j] = fma-chain over k = 0..s-1 of D[i][k] * X[k][j], starting from +0
//     Y = T.D^T    Y[i][j] = fma-chain over k = 0..s-1 of T[i][k] * D[j][k], starting from +0
// q = rint(double(Y) / double(Q)) (round-half-even, as np.round of the float64 quotient), written at the
// zigzag position.  s = 4, 8, 16 (and 2): VALU fma chains, one thread per column then per row, transposed
// through LDS.  s = 32, 64, 128 (and 256 via tiles of 128?) : v_mfma_f32_32x32x2_f32, whose accumulation is
// exactly that k-ordered fma chain.
#include "aej_common.h"
#include "aej_launch.h"
#include "aej_bigblock.h"
#include "aej_mfma.h"
#include <stdlib.h>

namespace aej {


// np.pad(mode='reflect') index map for a block clipped to n valid samples (jpeg.py:399-402)
__device__ __forceinline__ int reflect_pad_idx(int i, int n)
{
    if (i < n) return i;
    if (n <= 1) return 0;
    int p = 2 * (n - 1);
    int j = i % p;
    return j >= n ? p - j : j;
}

// position of raster element (r, c) in the zigzag sequence of an S x S block (jpeg.py:726-766): anti-diagonal d = r + c,
// odd diagonals run top -> bottom, even ones bottom -> top
template <int S>
__device__ __forceinline__ int zigzag_pos(int r, int c)
{
    const int d = r + c;
    if (d < S) {
        const int before = d * (d + 1) / 2;
        return before + ((d & 1) ? r : c);
    }
    const int dd = 2 * (S - 1) - d;
    const int before = S * S - (dd + 1) * (dd + 2) / 2;
    return before + ((d & 1) ? (S - 1 - c) : (S - 1 - r));
}

// np.round(block / q).astype(int32): float64 quotient, round half to even (jpeg.py:499-502).
__device__ __forceinline__ int quantise_f64(float y, int q)
{
    double v = (double)y / (double)q;
    return (int)rint(v);
}

// The same integer from float32 operations only (the float64 division is ~15 double-rate instructions per coefficient and was
// the longest phase of the low-frequency wave of every large leaf).  k = rint(y * (1/q)) is at most one off; the remainder
// r = fma(-k, q, y) = y - k q is EXACT in float32 (it is a multiple of ulp(y) no larger than 1.5 q), so comparing |r| with q / 2
// decides between k and its neighbour, and |r| == q / 2 is exactly the case where the float64 quotient is k +- 1/2 and
// np.round goes to the even one.  (A non-zero |2r - q| is at least one ulp(y) >= 2^-24 |y|, far above the 2^-53 relative spacing
// at which the rounded float64 quotient could fake a tie.)  Range guard: quantisers above 2^22 or quotients above 2^18 -- never
// produced by the codec's own tables -- take the float64 division.  tests/native/quantise_check.c verifies the sequence, with the
// reciprocal perturbed by +-4 ulp, against rint((double)y / q) on 5e8 random, tie and near-tie cases.
__device__ __forceinline__ int quantise_f32(float y, float qf, float *quot = nullptr)      // qf = (float)q, q <= 2^22; valid while |y / q| < 2^18
{
    const float t = y * __builtin_amdgcn_rcpf(qf);
    const float k = __builtin_rintf(t);
    const float r = __builtin_fmaf(-k, qf, y);
    const float h = 0.5f * qf, ar = __builtin_fabsf(r);
    int ki = (int)k;
    if (ar > h || (ar == h && (ki & 1))) ki += r > 0.f ? 1 : -1;
    if (quot) *quot = t;
    return ki;
}

__device__ __forceinline__ int quantise(float y, int q)
{
    float t;
    const int ki = quantise_f32(y, (float)q, &t);
    const bool slow = q > (1 << 22) || !(__builtin_fabsf(t) < 262144.0f);
    if (__any(slow)) return quantise_f64(y, q);
    return ki;
}

// Work items of one block size are the concatenation, over planes (b, l), of that plane's Morton-ordered leaf list.
// s_pref[p] = number of items in planes < p (built once per workgroup by wave 0); an item index is mapped back to
// (plane, index in plane) with a binary search.  Per-layer geometry is copied to LDS once so that the per-leaf code
// never indexes kernel arguments dynamically (that would be a dependent kernarg load per leaf).
struct LayerTab {
    int w[3], h[3];
    long long poff[3], coff[3], woff[3];
};

__device__ __forceinline__ int wave_incl_scan_i(int v, int /*lane*/) { return wave_scan_incl(v); }      // (threads 0..63 of the workgroup: a whole wave)

__device__ __forceinline__ void dct_prologue(const Geom &g, const QtGeom &q, const DctArgs &a, int *s_pref, LayerTab &lt)
{
    if (threadIdx.x < 3) {
        const int l = threadIdx.x;
        lt.w[l] = g.w[l]; lt.h[l] = g.h[l];
        lt.poff[l] = g.poff[l]; lt.coff[l] = q.coeff_off[l]; lt.woff[l] = q.work_off[l][a.k];
    }
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        int carry = 0;
        if (lane == 0) s_pref[0] = 0;
        for (int base = 0; base < a.nplanes; base += 64) {
            int p = base + lane;
            int v = p < a.nplanes ? a.work_count[(long long)p * kMaxSizes + a.k] : 0;
            int inc = wave_incl_scan_i(v, lane);
            if (p < a.nplanes) s_pref[p + 1] = carry + inc;
            carry += __builtin_amdgcn_readlane(inc, 63);
        }
    }
    __syncthreads();
}

__device__ __forceinline__ int4 fetch_item(const DctArgs &a, long long work_stride, const LayerTab &lt, const int *s_pref, long long item)
{
    int lo = 0, hi = a.nplanes;          // largest p with s_pref[p] <= item
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if ((long long)s_pref[mid] <= item) lo = mid; else hi = mid;
    }
    const int b = lo / 3, l = lo - 3 * b;
    const long long idx = (long long)b * work_stride + lt.woff[l] + (item - s_pref[lo]);
    return unpack_work(lo, a.work[idx]);
}

// The MFMA kernels (one leaf per workgroup and iteration) keep the descriptors of their next leaves in LDS: chunks of kDescChunk
// descriptors, one per thread, are fetched a whole chunk ahead, so that no global-memory latency sits between two leaves (a
// descriptor fetched when it is needed costs a full HBM round trip per leaf: measured 4 700 of 16 800 cycles per 64 x 64 leaf).
constexpr int kDescChunk = 64;

// ------------------------------------------------------------------------------------------------
// small blocks: S in {2, 4, 8, 16}; 256 threads = 256/S leaves, S threads per leaf.
// The descriptor of the next leaf is fetched while the current one is transformed.
// ------------------------------------------------------------------------------------------------
template <int S, bool WANT_DCT>
__global__ __launch_bounds__(256) void k_dct_small(Geom g, QtGeom q, DctArgs a, long long max_items)
{
    constexpr int LPB = 256 / S;
    constexpr int SS = S * S;
    __shared__ float sT[LPB * S * (S + 1)];
    __shared__ int sQ[LPB * SS];
    __shared__ float sD[SS];
    __shared__ int sZ[SS];
    __shared__ int sQm[3 * SS];
    __shared__ LayerTab lt;
    extern __shared__ int s_pref[];
    const int tid = threadIdx.x;
    for (int i = tid; i < SS; i += 256) { sD[i] = a.D[i]; sZ[i] = a.zzinv[i]; }
    for (int i = tid; i < 3 * SS; i += 256) sQm[i] = a.qm[i / SS] ? a.qm[i / SS][i % SS] : 1;
    dct_prologue(g, q, a, s_pref, lt);
    long long count = s_pref[a.nplanes];
    if (count > max_items) count = max_items;
    const long long wstride = q.work_stride[a.k];
    const int slot = tid / S, j = tid % S;
    const long long step = (long long)gridDim.x * LPB;
    long long base = (long long)blockIdx.x * LPB;
    int4 wk = make_int4(0, 0, 0, 0);
    if (base + slot < count) wk = fetch_item(a, wstride, lt, s_pref, base + slot);
    for (; base < count; base += step) {
        const bool active = base + slot < count;
        const int4 cur = wk;
        int layer = 0, b = 0;
        if (active) {
            b = cur.x / 3;
            layer = cur.x - b * 3;
            const int w = lt.w[layer], h = lt.h[layer];
            const float *src = a.norm + (long long)b * g.pstride + lt.poff[layer];
            const int hc = min(S, h - cur.z), wc = min(S, w - cur.y);
            const int col = cur.y + reflect_pad_idx(j, wc);
            float x[S];
#pragma unroll
            for (int k = 0; k < S; k++) x[k] = src[plane_elem(g.tiled, w, cur.z + reflect_pad_idx(k, hc), col)];
            if (base + step + slot < count) wk = fetch_item(a, wstride, lt, s_pref, base + step + slot);
            // T[i][j] = sum_k D[i][k] X[k][j]
#pragma unroll
            for (int i = 0; i < S; i++) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < S; k++) acc = __builtin_fmaf(sD[i * S + k], x[k], acc);
                sT[(slot * S + i) * (S + 1) + j] = acc;
            }
        }
        __syncthreads();
        long long out_base = 0;
        if (active) {
            out_base = (long long)b * q.coeff_stride + lt.coff[layer] + cur.w;
            // Y[i][jj] = sum_k T[i][k] D[jj][k], this thread owns row i = j
            float t[S];
#pragma unroll
            for (int k = 0; k < S; k++) t[k] = sT[(slot * S + j) * (S + 1) + k];
            float yv[S];
            int qv[S];
            float ymax = 0.f;
            int qmax = 0;
#pragma unroll
            for (int jj = 0; jj < S; jj++) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < S; k++) acc = __builtin_fmaf(t[k], sD[jj * S + k], acc);
                const int ridx = j * S + jj;
                if (WANT_DCT) a.dct_f32[out_base + ridx] = acc;
                yv[jj] = acc;
                qv[jj] = sQm[layer * SS + ridx];
                ymax = __builtin_fmaxf(ymax, __builtin_fabsf(acc));
                qmax = max(qmax, qv[jj]);
            }
            // one range test per row for the float32 quantiser (quantise_f32: q <= 2^22, |y / q| < 2^18), then branch-free
            if (!__any(qmax > (1 << 22) || !(ymax < 131072.0f))) {
#pragma unroll
                for (int jj = 0; jj < S; jj++) sQ[slot * SS + sZ[j * S + jj]] = quantise_f32(yv[jj], (float)qv[jj]);
            } else {
#pragma unroll
                for (int jj = 0; jj < S; jj++) sQ[slot * SS + sZ[j * S + jj]] = quantise_f64(yv[jj], qv[jj]);
            }
        }
        __syncthreads();
        // coalesced copy-out: S threads of a leaf write its S*S coefficients
        if (active) {
#pragma unroll
            for (int r = 0; r < S; r++) a.coeffs[out_base + r * S + j] = sQ[slot * SS + r * S + j];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// 4 x 4 blocks (the most numerous leaves -- half of the plane area of a natural image): FOUR LANES per leaf, registers and DPP only.
// Lane r of a quad loads row r of X (16 bytes; with the planes in 4 x 4 blocks a leaf is one 64-byte run, so a quad makes one 64-byte
// request and sixteen Morton-consecutive leaves a few contiguous lines).  T[r][.] = sum_k D[r][k] X[k][.] takes the other rows from the
// quad's lanes by DPP quad_perm broadcasts (the k-ascending fma chain of the contract), Y[r][.] = T[r][.].D^T is lane-local; four
// quotients per lane with the quantisers AND their reciprocals tabulated once per workgroup (v_rcp_f32 is the slowest instruction of
// the float32 quantiser); the leaf's 64 bytes pass through a wave-private LDS slab at their zigzag positions (LDS operations of one
// wave execute in order) and leave as one 16-byte store per lane: a wave writes 1 KiB contiguously.
// A workgroup walks a CONTIGUOUS range of the item list, so a lane's plane index moves forward by a comparison per leaf instead of a
// binary search, and what it reads and writes stays local.  (Round 3's kernel was one thread per leaf: 86 registers, 400 instructions per
// leaf of which the quantiser and its range test were two thirds; natural images 0.97 ms.)
// ------------------------------------------------------------------------------------------------
template <int SEL>
__device__ __forceinline__ float quad_bcast(float v)      // value of lane SEL of this lane's quad
{
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), SEL * 0x55, 0xf, 0xf, true));      // quad_perm [SEL, SEL, SEL, SEL]
}

__device__ __forceinline__ int quantise_f32_rcp(float y, float qf, float rq)      // quantise_f32 with the reciprocal looked up (same instruction, same bits)
{
    const float k = __builtin_rintf(y * rq);
    const float r = __builtin_fmaf(-k, qf, y);
    const float h = 0.5f * qf, ar = __builtin_fabsf(r);
    int ki = (int)k;
    if (ar > h || (ar == h && (ki & 1))) ki += r > 0.f ? 1 : -1;
    return ki;
}

// (the body takes its block index and block count as arguments: k_dct_multi below runs it on a share of its grid)
template <bool WANT_DCT>
__device__ __forceinline__ void dct4_body(const Geom &g, const QtGeom &q, const DctArgs &a, long long max_items, unsigned bid, unsigned nb)
{
    constexpr int LPB = 64;                   // leaves per workgroup and iteration
    __shared__ float sQf[3 * 16], sQr[3 * 16], sQh[3 * 16];
    __shared__ int sOut[4][16 * 16];          // per wave: 16 leaves x 16 coefficients
    __shared__ int sSlowQ;
    __shared__ LayerTab lt;
    extern __shared__ int s_pref[];
    const int tid = threadIdx.x;
    if (tid == 0) sSlowQ = 0;
    __syncthreads();
    if (tid < 48) {
        const int qi = a.qm[tid / 16] ? a.qm[tid / 16][tid % 16] : 1;
        sQf[tid] = (float)qi;
        sQr[tid] = __builtin_amdgcn_rcpf((float)qi);
        sQh[tid] = 0.5f * (float)qi;
        if (qi > (1 << 22)) sSlowQ = 1;      // (never with the codec's own tables: quantise_f32 needs q <= 2^22)
    }
    dct_prologue(g, q, a, s_pref, lt);       // ends with a barrier
    long long count = s_pref[a.nplanes];
    if (count > max_items) count = max_items;
    const bool slow_q = sSlowQ != 0;
    const long long wstride = q.work_stride[a.k];
    const int lane = tid & 63, wv = tid >> 6, r = tid & 3, slot = tid >> 2;
    float D[4][4], Dr[4];
#pragma unroll
    for (int i = 0; i < 16; i++) D[i >> 2][i & 3] = a.D[i];
#pragma unroll
    for (int k = 0; k < 4; k++) Dr[k] = a.D[r * 4 + k];
    int zz[4];
#pragma unroll
    for (int c = 0; c < 4; c++) zz[c] = zigzag_pos<4>(r, c);
    int *slab = sOut[wv] + (lane >> 2) * 16;
    // this workgroup's contiguous share of the items, a multiple of LPB
    const long long per = ((count + nb - 1) / nb + LPB - 1) / LPB * LPB;
    const long long first = (long long)bid * per, last = first + per < count ? first + per : count;
    int p = 0;                                // plane of the lane's current item: largest p with s_pref[p] <= item
    if (first + slot < last) {
        int lo = 0, hi = a.nplanes;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if ((long long)s_pref[mid] <= first + slot) lo = mid; else hi = mid;
        }
        p = lo;
    }
    for (long long base = first; base < last; base += LPB) {
        const long long item = base + slot;
        const bool active = item < last;
        float x[4] = { 0.f, 0.f, 0.f, 0.f };
        int layer = 0, b = 0;
        int4 cur = make_int4(0, 0, 0, 0);
        if (active) {
            while ((long long)s_pref[p + 1] <= item) p++;      // (item < count = s_pref[nplanes]: p + 1 <= nplanes)
            b = p / 3;
            layer = p - 3 * b;
            cur = unpack_work(p, a.work[(long long)b * wstride + lt.woff[layer] + (item - s_pref[p])]);
            const int w = lt.w[layer], h = lt.h[layer];
            const float *src = a.norm + (long long)b * g.pstride + lt.poff[layer];
            const int hc = min(4, h - cur.z), wc = min(4, w - cur.y);
            if (hc == 4 && wc == 4 && (w & 3) == 0) {
                const float4 v = *reinterpret_cast<const float4 *>(src + plane_elem(g.tiled, w, cur.z + r, cur.y));      // (tiled: the quad reads one 64-byte run)
                x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
            } else {                          // clipped at the plane border (np.pad reflect) or unaligned rows
                const int y = cur.z + reflect_pad_idx(r, hc);
#pragma unroll
                for (int c = 0; c < 4; c++) x[c] = src[plane_elem(g.tiled, w, y, cur.y + reflect_pad_idx(c, wc))];
            }
        }
        // T[r][c] = sum_k D[r][k] X[k][c]: row k of X comes from lane k of the quad (all 64 lanes take part; inactive quads carry zeros)
        float t[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float acc = 0.f;
            acc = __builtin_fmaf(Dr[0], quad_bcast<0>(x[c]), acc);
            acc = __builtin_fmaf(Dr[1], quad_bcast<1>(x[c]), acc);
            acc = __builtin_fmaf(Dr[2], quad_bcast<2>(x[c]), acc);
            acc = __builtin_fmaf(Dr[3], quad_bcast<3>(x[c]), acc);
            t[c] = acc;
        }
        // Y[r][c] = sum_k T[r][k] D[c][k]
        float y[4], ymax = 0.f;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) acc = __builtin_fmaf(t[k], D[c][k], acc);
            y[c] = acc;
            ymax = __builtin_fmaxf(ymax, __builtin_fabsf(acc));
        }
        const long long out_base = active ? (long long)b * q.coeff_stride + lt.coff[layer] + cur.w : 0;
        if (WANT_DCT && active) *reinterpret_cast<float4 *>(a.dct_f32 + out_base + r * 4) = make_float4(y[0], y[1], y[2], y[3]);
        const float4 qf = *reinterpret_cast<const float4 *>(&sQf[layer * 16 + r * 4]), rq = *reinterpret_cast<const float4 *>(&sQr[layer * 16 + r * 4]);
        // one range test per wave for the float32 quantiser (quantise_f32: q <= 2^22, |y / q| < 2^18), then branch-free
        if (!slow_q && !__any(!(ymax < 131072.0f))) {
            // quantise_f32_rcp with q / 2 looked up too: only a remainder at or beyond q / 2 needs the correction -- one test per wave (round 5:
            // the four corrections were four branches)
            const float4 qh = *reinterpret_cast<const float4 *>(&sQh[layer * 16 + r * 4]);
            const float qfa[4] = { qf.x, qf.y, qf.z, qf.w }, rqa[4] = { rq.x, rq.y, rq.z, rq.w }, qha[4] = { qh.x, qh.y, qh.z, qh.w };
            float kf[4], rem[4];
            unsigned long long fix = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                kf[c] = __builtin_rintf(y[c] * rqa[c]);
                rem[c] = __builtin_fmaf(-kf[c], qfa[c], y[c]);
                fix |= __builtin_amdgcn_ballot_w64(__builtin_fabsf(rem[c]) >= qha[c]);
            }
            int ki[4];
#pragma unroll
            for (int c = 0; c < 4; c++) ki[c] = (int)kf[c];
            if (fix != 0) {
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float ar = __builtin_fabsf(rem[c]);
                    if (ar > qha[c] || (ar == qha[c] && (ki[c] & 1))) ki[c] += rem[c] > 0.f ? 1 : -1;
                }
            }
#pragma unroll
            for (int c = 0; c < 4; c++) slab[zz[c]] = ki[c];
        } else {
            slab[zz[0]] = quantise_f64(y[0], (int)qf.x);
            slab[zz[1]] = quantise_f64(y[1], (int)qf.y);
            slab[zz[2]] = quantise_f64(y[2], (int)qf.z);
            slab[zz[3]] = quantise_f64(y[3], (int)qf.w);
        }
        if (active) {
            const int4 o = reinterpret_cast<const int4 *>(slab)[r];
            reinterpret_cast<int4 *>(a.coeffs + out_base)[r] = o;      // coefficient offsets are multiples of 4
        }
    }
}
template <bool WANT_DCT>
__global__ __launch_bounds__(256) void k_dct4(Geom g, QtGeom q, DctArgs a, long long max_items)
{
    dct4_body<WANT_DCT>(g, q, a, max_items, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// 8 x 8 blocks: eight LANES per leaf, registers and wavefront shuffles only -- no LDS transpose, no workgroup barrier.
// Lane j of a leaf's group loads column j of X (eight rows of 32 contiguous bytes per leaf), so the column transform
// T[.][j] = D.X[.][j] is lane-local (64 fma, D in registers).  An 8 x 8 transpose inside the 8-lane group -- three butterfly
// stages: partners lane ^ 1 and lane ^ 2 by DPP quad_perm, lane ^ 4 by DPP row_shl / row_shr under bank masks -- hands lane i
// the row T[i][.], so Y[i][.] = T[i][.].D^T is lane-local again.  Same k-ordered fma chains as every other kernel.  The leaf's
// 256 bytes of coefficients pass through a wave-private LDS slab at their zigzag positions (LDS operations of one wave execute
// in order) and leave as two 16-byte stores per lane: a wave writes 2 KiB contiguously.
// ------------------------------------------------------------------------------------------------
template <int CTRL, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float old, float v)
{
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(old), (int)__float_as_uint(v), CTRL, 0xf, BANK_MASK, false));
}
// value of lane ^ S (S = 1, 2, 4) within the lane's group of 8
template <int S>
__device__ __forceinline__ float lane_xor(float v)
{
    if (S == 1) return dpp_f<0xB1>(v, v);                    // quad_perm [1, 0, 3, 2]
    if (S == 2) return dpp_f<0x4E>(v, v);                    // quad_perm [2, 3, 0, 1]
    float r = dpp_f<0x104, 0x5>(v, v);                       // row_shl:4 -> lanes 0-3, 8-11 of each row take lane + 4
    return dpp_f<0x114, 0xA>(r, v);                          // row_shr:4 -> lanes 4-7, 12-15 take lane - 4
}
// a[r] of lane l  <->  a[l] of lane r  (l, r = 0..7 inside the group)
template <int S>
__device__ __forceinline__ void transpose8_stage(float (&a)[8], bool upper)      // upper = (lane & S) != 0
{
#pragma unroll
    for (int r = 0; r < 8; r++)
        if ((r & S) == 0) {
            const float send = upper ? a[r] : a[r | S];
            const float recv = lane_xor<S>(send);
            if (upper) a[r] = recv; else a[r | S] = recv;
        }
}

// (Round 5 rebuilt this kernel the way the 16 x 16 one was -- compile-time plane layout, contiguous item ranges, tabulated quantiser, the basis
// pinned in 64 vector registers, two leaves of pixel look-ahead: half the instructions, 0.165 -> 0.160 ms alone at 164 registers, and a pipelined
// step that did not move: it runs in the 104-register gap beside three blur workgroups, and bounded to 96 registers the rebuilt form spills.  The
// round-4 form below stays: 58 registers, 9 KiB.)
// (the body takes its block index and block count as arguments: k_dct_multi below runs it on a share of its grid)
template <bool WANT_DCT>
__device__ __forceinline__ void dct8_body(const Geom &g, const QtGeom &q, const DctArgs &a, long long max_items, unsigned bid, unsigned nb)
{
    constexpr int S = 8, SS = 64, LPB = 32;
    __shared__ float sQf[3 * SS];
    __shared__ int sOut[4][8 * SS];          // per wave: 8 leaves x 64 coefficients
    __shared__ LayerTab lt;
    extern __shared__ int s_pref[];
    const int tid = threadIdx.x;
    if (tid < 3 * SS) sQf[tid] = (float)(a.qm[tid / SS] ? a.qm[tid / SS][tid % SS] : 1);
    dct_prologue(g, q, a, s_pref, lt);       // ends with a barrier
    long long count = s_pref[a.nplanes];
    if (count > max_items) count = max_items;
    const long long wstride = q.work_stride[a.k];
    const int lane = tid & 63, wv = tid >> 6, j = tid & 7, slot = tid >> 3;      // slot: leaf of this block iteration (0..31)
    float D[8][8];
#pragma unroll
    for (int i = 0; i < 64; i++) D[i >> 3][i & 7] = a.D[i];
    int zz[8];
#pragma unroll
    for (int c = 0; c < 8; c++) zz[c] = zigzag_pos<S>(j, c);                     // after the transpose this lane owns row j
    int *slab = sOut[wv] + ((lane >> 3) * SS);
    const long long step = (long long)nb * LPB;
    long long base = (long long)bid * LPB;
    int4 wk = make_int4(0, 0, 0, 0);
    if (base + slot < count) wk = fetch_item(a, wstride, lt, s_pref, base + slot);
    for (; base < count; base += step) {
        const bool active = base + slot < count;
        const int4 cur = wk;
        float x[8];
        int layer = 0, b = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = 0.f;
        if (active) {
            b = cur.x / 3;
            layer = cur.x - b * 3;
            const int w = lt.w[layer], h = lt.h[layer];
            const float *src = a.norm + (long long)b * g.pstride + lt.poff[layer];
            const int hc = min(S, h - cur.z), wc = min(S, w - cur.y);
            const int col = cur.y + reflect_pad_idx(j, wc);
            if (hc == S) {
#pragma unroll
                for (int k = 0; k < 8; k++) x[k] = src[plane_elem(g.tiled, w, cur.z + k, col)];      // (tiled: the leaf is two 128-byte lines)
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) x[k] = src[plane_elem(g.tiled, w, cur.z + reflect_pad_idx(k, hc), col)];
            }
            if (base + step + slot < count) wk = fetch_item(a, wstride, lt, s_pref, base + step + slot);
        }
        // T[i][j] = sum_k D[i][k] X[k][j]: this lane's column
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 8; k++) acc = __builtin_fmaf(D[i][k], x[k], acc);
            t[i] = acc;
        }
        // (all 64 lanes take part in the shuffles; the groups of inactive leaves carry zeros)
        transpose8_stage<1>(t, (lane & 1) != 0);
        transpose8_stage<2>(t, (lane & 2) != 0);
        transpose8_stage<4>(t, (lane & 4) != 0);
        // now t[k] = T[j][k]: Y[j][c] = sum_k T[j][k] D[c][k]
        float y[8], ymax = 0.f, qmax = 0.f, qf[8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 8; k++) acc = __builtin_fmaf(t[k], D[c][k], acc);
            y[c] = acc;
            qf[c] = sQf[layer * SS + j * S + c];
            ymax = __builtin_fmaxf(ymax, __builtin_fabsf(acc));
            qmax = __builtin_fmaxf(qmax, qf[c]);
        }
        const long long out_base = active ? (long long)b * q.coeff_stride + lt.coff[layer] + cur.w : 0;
        if (WANT_DCT && active) {
#pragma unroll
            for (int c = 0; c < 8; c++) a.dct_f32[out_base + j * S + c] = y[c];
        }
        // one range test per wave for the float32 quantiser (quantise_f32: q <= 2^22, |y / q| < 2^18), then branch-free
        if (!__any(qmax > 4194304.0f || !(ymax < 131072.0f))) {
#pragma unroll
            for (int c = 0; c < 8; c++) slab[zz[c]] = quantise_f32(y[c], qf[c]);
        } else {
#pragma unroll
            for (int c = 0; c < 8; c++) slab[zz[c]] = quantise_f64(y[c], (int)qf[c]);
        }
        if (active) {
            const int4 o0 = reinterpret_cast<const int4 *>(slab)[2 * j], o1 = reinterpret_cast<const int4 *>(slab)[2 * j + 1];
            int4 *dst = reinterpret_cast<int4 *>(a.coeffs + out_base);      // coefficient offsets are multiples of 4
            dst[2 * j] = o0;
            dst[2 * j + 1] = o1;
        }
    }
}
template <bool WANT_DCT>
__global__ __launch_bounds__(256) void k_dct8_shfl(Geom g, QtGeom q, DctArgs a, long long max_items)
{
    dct8_body<WANT_DCT>(g, q, a, max_items, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// 16 x 16 blocks on the matrix pipe: one WAVE per leaf, v_mfma_f32_16x16x4_f32, no LDS and no barrier between the products.
// The instruction accumulates its four k values as the k-ascending fma chain (checked bit for bit on hardware,
// tools/ubench/mfma16_order.hip), so four of them in a row are exactly the contract's chain over k = 0..15.
//   P = X^T.D^T : A operand of step s = X[4 s + g][i] (lane = 16 g + i), i.e. the lanes of a 16-lane group read 64 contiguous
//                 bytes of one plane row STRAIGHT FROM GLOBAL MEMORY into the operand register; B = D[i][4 s + g], loop-invariant.
//                 P[i][j] = T[j][i]; the C layout leaves P[4 g + r][i] in register r.
//   Y = T.D^T   : A operand of step s = T[i][4 s + g] = P[4 s + g][i]: a 4 x 4 transpose between the four 16-lane groups and
//                 the four registers -- two v_permlane32_swap + two v_permlane16_swap (gfx950), no LDS round trip.
// Y[4 g + r][i] sits in register r: quantise, stage the leaf's 1 KiB through a wave-private LDS slab at the zigzag positions,
// and write it with ONE 16-byte store per lane.  The next leaf's pixels and the descriptor after that are in flight meanwhile.
// ------------------------------------------------------------------------------------------------
typedef float floatx4 __attribute__((ext_vector_type(4)));

// Round 5: what the wave does per leaf besides the eight MFMAs was 224 vector instructions (PMC: the kernel issue-bound at 0.69, not
// HBM-bound).  Now: the plane layout is a template parameter (as a run-time flag every element address carried a branch); a workgroup
// walks a CONTIGUOUS range of the item list, so the plane of an item is tracked by a scalar comparison instead of a binary search through
// LDS with a readfirstlane per probe; descriptors and everything derived from them are scalar (the work list is read by scalar loads);
// the four pixel loads of an unclipped leaf are scalar base + per-lane constant; quantisers and their reciprocals come from per-lane
// tables (two 16-byte LDS reads), and the quantiser's correction is one rarely-taken branch per leaf.
// (the body takes its block index and block count as arguments: k_dct_multi below runs it on a share of its grid)
struct __attribute__((aligned(16))) Dct16Lds {     // (declared once, in dct16_body: as statics of the two layout instantiations it would be allocated twice)
    float qf[3 * 256], qr[3 * 256];                  // [layer][lane][r]: what a lane reads for its four outputs
    int out[4][256];
    int slow;
    LayerTab lt;
};
template <bool WANT_DCT, bool TILED>
__device__ __forceinline__ void dct16_impl(Dct16Lds &L, const Geom &g, const QtGeom &q, const DctArgs &a, long long max_items, unsigned bid, unsigned nb)
{
    constexpr int S = 16, SS = 256;
    float (&sQf)[3 * SS] = L.qf, (&sQr)[3 * SS] = L.qr;
    int (&sOut)[4][SS] = L.out;
    int &sSlowQ = L.slow;
    LayerTab &lt = L.lt;
    extern __shared__ int s_pref[];
    const int tid = threadIdx.x;
    if (tid == 0) sSlowQ = 0;
    __syncthreads();
    for (int e = tid; e < 3 * SS; e += 256) {
        const int layer = e / SS, ln = (e >> 2) & 63, r = e & 3;
        const int ridx = (4 * (ln >> 4) + r) * S + (ln & 15);
        const int qi = a.qm[layer] ? a.qm[layer][ridx] : 1;
        sQf[e] = (float)qi;
        sQr[e] = __builtin_amdgcn_rcpf((float)qi);
        if (qi > (1 << 22)) sSlowQ = 1;      // (never with the codec's own tables: quantise_f32 needs q <= 2^22)
    }
    dct_prologue(g, q, a, s_pref, lt);       // ends with a barrier
    // (LDS values are per-lane to the compiler, and so is everything compared as 64-bit integers -- the scalar unit has no such comparison: item
    // indices are ints, as the prefix sums are)
    int count = __builtin_amdgcn_readfirstlane(s_pref[a.nplanes]);
    if ((long long)count > max_items) count = (int)max_items;
    const bool slow_q = __builtin_amdgcn_readfirstlane(sSlowQ) != 0;
    const long long wstride = q.work_stride[a.k];
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), gq = lane >> 4, i = lane & 15;
    auto rfl = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    float dB[4];
    int zz[4];
#pragma unroll
    for (int s = 0; s < 4; s++) dB[s] = a.D[i * S + 4 * s + gq];
#pragma unroll
    for (int r = 0; r < 4; r++) zz[r] = zigzag_pos<S>(4 * gq + r, i);
    int *out_slab = sOut[wv];
    // per-layer geometry in scalar registers
    int lw[3], lh[3];
    long long lpoff[3], lcoff[3], lwoff[3];
#pragma unroll
    for (int l = 0; l < 3; l++) {
        lw[l] = rfl(lt.w[l]); lh[l] = rfl(lt.h[l]);
        lpoff[l] = ((long long)rfl((int)(lt.poff[l] >> 32)) << 32) | (unsigned)rfl((int)lt.poff[l]);
        lcoff[l] = ((long long)rfl((int)(lt.coff[l] >> 32)) << 32) | (unsigned)rfl((int)lt.coff[l]);
        lwoff[l] = ((long long)rfl((int)(lt.woff[l] >> 32)) << 32) | (unsigned)rfl((int)lt.woff[l]);
    }
    auto sel = [](int l, auto v0, auto v1, auto v2) { return l == 0 ? v0 : l == 1 ? v1 : v2; };
    // this workgroup's contiguous share of the items (a multiple of 4: one leaf per wave and round)
    const int per = (int)((((unsigned)count + nb - 1) / nb + 3) / 4 * 4);
    const long long first64 = (long long)bid * per;
    if (first64 >= count) return;
    const int first = (int)first64, last = count - first > per ? first + per : count;
    int item = first + wv;
    if (item >= last) return;
    // plane of the item whose descriptor is fetched next: [pbeg, pend) = its item range (scalar; items only move forward)
    int p = 0;
    {
        int lo = 0, hi = a.nplanes;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (rfl(s_pref[mid]) <= item) lo = mid; else hi = mid;
        }
        p = lo;
    }
    int pbeg = rfl(s_pref[p]), pend = rfl(s_pref[p + 1]);
    struct Desc { int plane; unsigned xy; int coef; };
    auto fetch = [&](int it) -> Desc {               // scalar: the work list through the scalar cache
        if (it > last - 1) it = last - 1;                  // (clamped: the loads past the end are harmless and unused; still >= every earlier item of this wave)
        while (it >= pend) { p++; pbeg = pend; pend = rfl(s_pref[p + 1]); }
        int ps = p;                                        // (p also addresses LDS, which parks it in a vector register: pin a scalar copy for the arithmetic)
        asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(ps) : "v"(p));
        const int b = ps / 3, l = ps - 3 * b;
        const LeafWork *wp = a.work + ((long long)b * wstride + sel(l, lwoff[0], lwoff[1], lwoff[2]) + (long long)(it - pbeg));
        // The work list is constant for the life of this kernel: read through the constant address space, a uniform address makes it a SCALAR load
        // -- the words arrive in scalar registers, and the load is counted on the scalar counter, so that waiting for a descriptor never means
        // waiting for the pixel loads issued around it (vector loads retire in order on one counter).
        typedef const __attribute__((address_space(4))) LeafWork *ConstWork;
        ConstWork cw = (ConstWork)(unsigned long long)wp;
        Desc d;
        d.plane = ps;
        d.xy = cw->xy;
        d.coef = cw->coef;
        return d;
    };
    auto landed = [&](const Desc &d) { return d; };
    const unsigned lane_tiled = (unsigned)((i >> 2) * 16 + gq * 4 + (i & 3));
    auto load_x = [&](const Desc &d, float (&x)[4]) {      // d scalar
        const int b = d.plane / 3, layer = d.plane - b * 3;
        const int w = sel(layer, lw[0], lw[1], lw[2]), h = sel(layer, lh[0], lh[1], lh[2]);
        const int dy = (int)(d.xy & 0xffffu), dz = (int)(d.xy >> 16);
        const float *src = a.norm + ((long long)b * g.pstride + sel(layer, lpoff[0], lpoff[1], lpoff[2]));
        if (dz + S <= h && dy + S <= w) {
            if (TILED) {                                   // rows 4 s + gq: block row dz / 4 + s, row gq of the block; column dy + i
                const float *row0 = src + ((long long)(dz >> 2) * (w >> 2) + (dy >> 2)) * 16;
#pragma unroll
                for (int s = 0; s < 4; s++) x[s] = (row0 + (long long)s * (w >> 2) * 16)[lane_tiled];
            } else {
                const float *row0 = src + ((long long)dz * w + dy);
                const unsigned lane_rm = (unsigned)(gq * w + i);
#pragma unroll
                for (int s = 0; s < 4; s++) x[s] = (row0 + (long long)s * 4 * w)[lane_rm];
            }
        } else {                                           // clipped at the plane border: np.pad(reflect) indices
            const int hc = min(S, h - dz), wc = min(S, w - dy);
            const int col = dy + reflect_pad_idx(i, wc);
#pragma unroll
            for (int s = 0; s < 4; s++) x[s] = src[plane_elem(TILED ? 1 : 0, w, dz + reflect_pad_idx(4 * s + gq, hc), col)];
        }
    };
    // Bytes in flight are what bounds this kernel once the instruction count is down (a leaf is 1 KiB per wave; with one leaf of look-ahead
    // the kernel ran at 3.7 TB/s): the pixels of the next TWO leaves and the descriptor of the third are in flight under a leaf's work.  Three
    // buffers, the loop unrolled three-fold so that no register with a load outstanding is ever copied.
    Desc d0 = landed(fetch(item)), d1, d2;
    float x0[4], x1[4], x2[4];
    load_x(d0, x0);
    d1 = landed(fetch(item + 4));
    Desc dv = fetch(item + 8);
    load_x(d1, x1);
    auto step = [&](const float (&x_cur)[4], const Desc &d_cur, float (&x_fill)[4], Desc &d_fill) {
        // (the descriptor load goes out BEFORE the pixel loads: the counter of outstanding loads retires in order, so waiting for a descriptor
        // one leaf later must not imply waiting for pixels requested after it)
        d_fill = landed(dv);                                   // leaf item + 8: requested one leaf ago
        dv = fetch(item + 12);
        load_x(d_fill, x_fill);
        const int b = d_cur.plane / 3, layer = d_cur.plane - b * 3;
        floatx4 pacc = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int s = 0; s < 4; s++) pacc = __builtin_amdgcn_mfma_f32_16x16x4f32(x_cur[s], dB[s], pacc, 0, 0, 0);
        // (__float_as_uint, not __builtin_bit_cast: hipcc 7.2 folds a bit_cast of an ext-vector ELEMENT to element 0)
        const auto p02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(pacc[0]), __float_as_uint(pacc[2]), false, false);
        const auto p13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(pacc[1]), __float_as_uint(pacc[3]), false, false);
        const unsigned p02a = p02[0], p02b = p02[1], p13a = p13[0], p13b = p13[1];
        const auto q01 = __builtin_amdgcn_permlane16_swap(p02a, p13a, false, false);
        const auto q23 = __builtin_amdgcn_permlane16_swap(p02b, p13b, false, false);
        const unsigned t0 = q01[0], t1 = q01[1], t2 = q23[0], t3 = q23[1];
        const float t[4] = { __uint_as_float(t0), __uint_as_float(t1), __uint_as_float(t2), __uint_as_float(t3) };
        floatx4 y = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int s = 0; s < 4; s++) y = __builtin_amdgcn_mfma_f32_16x16x4f32(t[s], dB[s], y, 0, 0, 0);
        const long long out_base = (long long)b * q.coeff_stride + sel(layer, lcoff[0], lcoff[1], lcoff[2]) + d_cur.coef;
        if (WANT_DCT) {
#pragma unroll
            for (int r = 0; r < 4; r++) a.dct_f32[out_base + (4 * gq + r) * S + i] = y[r];
        }
        const floatx4 qf = *reinterpret_cast<const floatx4 *>(&sQf[layer * SS + lane * 4]), rq = *reinterpret_cast<const floatx4 *>(&sQr[layer * SS + lane * 4]);
        const float ymax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(y[0]), __builtin_fabsf(y[1])), __builtin_fmaxf(__builtin_fabsf(y[2]), __builtin_fabsf(y[3])));
        // one range test per leaf for the float32 quantiser (quantise_f32: q <= 2^22, |y / q| < 2^18)
        if (!slow_q && !__any(!(ymax < 131072.0f))) {
            // quantise_f32 with the reciprocal looked up: k = rint(y / q) is at most one off, the remainder r = y - k q is exact, and
            // only 2 |r| >= q needs the correction -- one test per leaf
            float kf[4], rem[4];
            unsigned long long fix = 0;           // lanes with a remainder at or beyond q / 2 (ballots: the comparisons write scalar masks, the scalar unit ors them)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                kf[r] = __builtin_rintf(y[r] * rq[r]);
                rem[r] = __builtin_fmaf(-kf[r], qf[r], y[r]);
                const float ar = __builtin_fabsf(rem[r]);
                fix |= __builtin_amdgcn_ballot_w64(ar + ar >= qf[r]);
            }
            int ki[4];
#pragma unroll
            for (int r = 0; r < 4; r++) ki[r] = (int)kf[r];
            if (fix != 0) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float ar = __builtin_fabsf(rem[r]), h = 0.5f * qf[r];
                    if (ar > h || (ar == h && (ki[r] & 1))) ki[r] += rem[r] > 0.f ? 1 : -1;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) out_slab[zz[r]] = ki[r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) out_slab[zz[r]] = quantise_f64(y[r], (int)qf[r]);
        }
        // (LDS operations of one wave execute in order: the reads below see this wave's writes, and the next leaf's writes
        // come after these reads)
        const int4 o = reinterpret_cast<const int4 *>(out_slab)[lane];
        reinterpret_cast<int4 *>(a.coeffs + out_base)[lane] = o;
    };
    while (true) {
        step(x0, d0, x2, d2); item += 4; if (item >= last) break;
        step(x1, d1, x0, d0); item += 4; if (item >= last) break;
        step(x2, d2, x1, d1); item += 4; if (item >= last) break;
    }
}
template <bool WANT_DCT>
__device__ __forceinline__ void dct16_body(const Geom &g, const QtGeom &q, const DctArgs &a, long long max_items, unsigned bid, unsigned nb)
{
    __shared__ Dct16Lds L;
    if (g.tiled) dct16_impl<WANT_DCT, true>(L, g, q, a, max_items, bid, nb);
    else dct16_impl<WANT_DCT, false>(L, g, q, a, max_items, bid, nb);
}
template <bool WANT_DCT>
__global__ __launch_bounds__(256) void k_dct16_mfma(Geom g, QtGeom q, DctArgs a, long long max_items)
{
    dct16_body<WANT_DCT>(g, q, a, max_items, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// S = 256, 512, 1024 (aej_bigblock.h): T = D.X into this workgroup's scratch, then Y = T.D^T, quantise, zigzag scatter
// ------------------------------------------------------------------------------------------------
template <int S, bool WANT_DCT>
__global__ __launch_bounds__(256) void k_dct_big(Geom g, QtGeom q, DctArgs a, long long max_items)
{
    __shared__ BigTileLds L;
    __shared__ LayerTab lt;
    extern __shared__ int s_pref[];
    dct_prologue(g, q, a, s_pref, lt);
    long long count = s_pref[a.nplanes];
    if (count > max_items) count = max_items;
    const long long wstride = q.work_stride[a.k];
    float *T = a.scratch + (long long)blockIdx.x * S * S;
    for (long long item = blockIdx.x; item < count; item += gridDim.x) {
        const int4 cur = fetch_item(a, wstride, lt, s_pref, item);
        const int b = cur.x / 3, layer = cur.x - b * 3;
        const int w = lt.w[layer], h = lt.h[layer];
        const float *src = a.norm + (long long)b * g.pstride + lt.poff[layer];
        const int hc = min(S, h - cur.z), wc = min(S, w - cur.y);
        const float *D = a.D;
        big_product<S>(L,
            [&](int i, int k) { return D[i * S + k]; },
            [&](int k, int j) { return src[plane_elem(g.tiled, w, cur.z + reflect_pad_idx(k, hc), cur.y + reflect_pad_idx(j, wc))]; },
            [&](int i, int j, float v) { T[i * S + j] = v; });
        big_scratch_sync();
        const long long out_base = (long long)b * q.coeff_stride + lt.coff[layer] + cur.w;
        const int *qm = a.qm[layer];
        big_product<S>(L,
            [&](int i, int k) { return T[i * S + k]; },
            [&](int k, int j) { return D[j * S + k]; },
            [&](int i, int j, float v) {
                const int ridx = i * S + j;
                if (WANT_DCT) a.dct_f32[out_base + ridx] = v;
                a.coeffs[out_base + a.zzinv[ridx]] = quantise(v, qm ? qm[ridx] : 1);
            });
        big_scratch_sync();      // the next leaf overwrites T
    }
}

// LDS-DMA (global_load_lds): per-lane global source, LDS destination = wave-uniform base (M0) + lane * size.  Issued from inline
// assembly on purpose: the compiler treats the builtin as a store to LDS that any later LDS read may alias and drains it
// (s_waitcnt vmcnt(0)) before the very next ds_read -- which would expose the whole prefetch latency once per leaf.  Here the
// kernel itself orders the transfers (wait_vmem_but + lds_barrier); M0 is saved / restored inside the statement.
__device__ __forceinline__ void glds16(const float *g, float *lds_wave_base)
{
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}
__device__ __forceinline__ void glds4(const float *g, float *lds_wave_base)
{
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}

// ------------------------------------------------------------------------------------------------
// large blocks: S in {32, 64, 128}; one workgroup of (S/32)^2 waves per leaf, one 32x32 output tile per wave.
// P = X^T.D^T  (P[r][c] = T[c][r]) with A[i][k] = X[k][I0+i] read from LDS rows, B[k][j] = D[J0+j][k] held
// in S/2 registers for the whole kernel; P goes to LDS as stored, and Y = T.D^T reads A[i][k] = P[k][I0+i]
// with the same row pattern and the same B registers.  Both LDS read patterns are 32 consecutive floats per
// half-wave: conflict-free.  The quantisers of the lane's 16 outputs and the next leaf's descriptor are
// loaded before the MFMA chains so that their latency hides under them.
// ------------------------------------------------------------------------------------------------
template <int S>
struct MfmaCfg {
    static constexpr int NT = S / 32;                       // 32x32 output tiles per side
    static constexpr int TPW = S == 128 ? 2 : 1;            // tiles per wave (same tile column, so D registers are shared)
    static constexpr int NWAVES = NT * NT / TPW;
    static constexpr int NTHREADS = NWAVES * 64;
    static constexpr int MINW = S == 32 ? 4 : S == 64 ? 3 : 2;   // waves per SIMD the register budget is sized for
    static constexpr int NXB = S == 128 ? 1 : 2;            // X buffers (3 x 64 KiB would not fit for S = 128)
};

// scalar-base form: every lane adds its own byte offset to one wave-uniform 64-bit base (no per-lane 64-bit address arithmetic)
__device__ __forceinline__ void glds16_s(const float *base_uniform, unsigned lane_byte_off, float *lds_wave_base)
{
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
    const unsigned long long b = (unsigned long long)(size_t)base_uniform;
    const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)b), bhi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    const unsigned long long sb = ((unsigned long long)bhi << 32) | blo;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_byte_off), "s"(dst), "s"(sb) : "memory");
}

// X -> LDS by LDS-DMA (no VGPR staging): full leaves 1 KiB (256 floats) per wave instruction, clipped or unaligned ones one
// dword per lane with the np.pad(reflect) index map applied to the source address.  `xoff` = this lane's element offsets
// (row * w + column) of its pieces of a full leaf for the layer width `w` they were computed for (dct_x_offsets).
template <int S, int NWAVES>
struct XOffsets { unsigned off[S * S / 256 / NWAVES]; int w; };

template <int S, int NWAVES>
__device__ __forceinline__ void dct_x_offsets(XOffsets<S, NWAVES> &xo, int w, int tiled, int wave, int lane)
{
    constexpr int ROWS = 256 / S;                 // rows per wave instruction
#pragma unroll
    for (int t = 0; t < S * S / 256 / NWAVES; t++) {
        const int chunk = t * NWAVES + wave;
        const int r = chunk * ROWS + lane / (S / 4), c = (lane % (S / 4)) * 4;
        xo.off[t] = (unsigned)plane_elem(tiled, w, r, c) * 4u;        // (tiled: a wave instruction's 4 rows x 64 columns are one contiguous KiB)
    }
    xo.w = w;
}

template <int S, int NWAVES>
__device__ __forceinline__ void dct_load_x(const float *src, int w, int h, int tiled, const int4 &d, float *sX, int wave, int lane, XOffsets<S, NWAVES> &xo)
{
    constexpr int SS = S * S;
    const int hc = min(S, h - d.z), wc = min(S, w - d.y);
    if (hc == S && wc == S && (w & 3) == 0) {
        if (xo.w != w) dct_x_offsets<S, NWAVES>(xo, w, tiled, wave, lane);     // the layer changed (wave-uniform)
        const float *leaf = src + plane_elem(tiled, w, d.z, d.y);
#pragma unroll
        for (int t = 0; t < SS / 256 / NWAVES; t++) glds16_s(leaf, xo.off[t], sX + (t * NWAVES + wave) * 256);
    } else {
#pragma unroll 4
        for (int t = 0; t < SS / 64 / NWAVES; t++) {
            const int chunk = t * NWAVES + wave;
            const int idx = chunk * 64 + lane;
            const int r = idx / S, c = idx - r * S;
            glds4(src + plane_elem(tiled, w, d.z + reflect_pad_idx(r, hc), d.y + reflect_pad_idx(c, wc)), sX + chunk * 64);
        }
    }
}

// Synchronisation of the MFMA kernels.  __syncthreads() is "s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier": it would drain, three
// times per leaf, every global-memory operation of the wave -- the LDS-DMA prefetch of the NEXT leaf and the coefficient stores
// of the previous one included -- and expose their latency.  What the three hand-offs through LDS need is the wave's LDS
// operations only; the one thing that has to have LANDED from global memory, the current leaf's X, is waited for with a
// COUNTED vmcnt that leaves the younger stores in flight (vector-memory operations retire in issue order).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N> __device__ __forceinline__ void wait_vmem_but() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }


template <int S, bool WANT_DCT>
__device__ __forceinline__ void dct_mfma_leaves(const Geom &g, const QtGeom &q, const DctArgs &a, long long max_items, const LeafWork *__restrict__ work /* = a.work, read-only: scalar loads */,
                                                unsigned bid, unsigned nb)
{
    using C = MfmaCfg<S>;
    constexpr int NT = C::NT, TPW = C::TPW, NWAVES = C::NWAVES, NTHREADS = C::NTHREADS;
    constexpr int SS = S * S;
    constexpr int NXB = C::NXB;        // X double-buffered: the next leaf's X is fetched by LDS-DMA while this one is transformed
    constexpr int NCOPY = SS / (NTHREADS * 4);                 // 16-byte coefficient stores per thread and leaf
    constexpr int NYOUNG = NCOPY + (WANT_DCT ? 16 * TPW : 0);  // global stores a wave issues AFTER the prefetch of the next X
    static_assert(NYOUNG < 60, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sXb = smem;                 // [NXB][S][S]  X, the current one later reused as int staging for the zigzag scatter
    float *sP = smem + NXB * SS;       // [S][S]  P = T^T
    int4 *s_desc = reinterpret_cast<int4 *>(smem + (NXB + 1) * SS);                      // [2][kDescChunk] leaf descriptors
    LayerTab &lt = *reinterpret_cast<LayerTab *>(s_desc + 2 * kDescChunk);
    int *s_pref = reinterpret_cast<int *>(s_desc + 2 * kDescChunk) + (sizeof(LayerTab) + 3) / 4;   // [nplanes + 1]
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wj = wave % NT, wi0 = wave / NT;       // tile column; first tile row (the others are wi0 + t * NT / TPW)
    const int J0 = wj * 32;
    const int li = lane & 31, lh = lane >> 5;

    // B operand registers: D[J0 + li][2*s + lh]
    float dreg[S / 2];
#pragma unroll
    for (int s = 0; s < S / 2; s++) dreg[s] = a.D[(J0 + li) * S + 2 * s + lh];

    dct_prologue(g, q, a, s_pref, lt);
    // The waits inside the loop are written by hand (inline assembly the compiler's wait-count bookkeeping cannot see), so the
    // loads issued so far are consumed HERE: an empty statement that takes each register as an operand makes the compiler
    // wait for it now instead of guarding its first use in the loop with counted waits that would also catch the prefetch.
#pragma unroll
    for (int s = 0; s < S / 2; s++) asm volatile("" : "+v"(dreg[s]));
    long long count = s_pref[a.nplanes];
    if (count > max_items) count = max_items;
    const long long wstride = q.work_stride[a.k];
    const long long step = nb;
    long long item = bid;
    // leaf k of this workgroup is work item bid + k * nb; chunk c = leaves [c * kDescChunk, (c + 1) * kDescChunk)
    auto load_chunk = [&](long long c) {           // threads 0 .. kDescChunk - 1: one descriptor each, into s_desc[c & 1]
        if (tid < kDescChunk) {
            const long long it = (long long)bid + (c * kDescChunk + tid) * step;
            int4 d = make_int4(0, 0, 0, 0);
            if (it < count) d = fetch_item(a, wstride, lt, s_pref, it);
            s_desc[(c & 1) * kDescChunk + tid] = d;
        }
    };
    load_chunk(0);
    load_chunk(1);
    __syncthreads();
    long long k = 0;                               // index of the current leaf of this workgroup
    // A leaf descriptor is the same in every lane: its fields are moved to scalar registers (readfirstlane) as soon as they are
    // read from LDS, so that everything derived from them -- plane, layer, addresses, the layer's geometry (selected from the
    // kernel arguments, not looked up in LDS) -- is scalar-ALU work.  Vector instructions of this wave compete for issue slots with
    // the MFMAs of the other waves on the SIMD; measured, the ~150 vector instructions of address arithmetic per leaf cost
    // 2 000 cycles there.
    struct LeafU { int plane, x, y, coef; };
    auto uniform = [](const int4 &d) {
        return LeafU{ __builtin_amdgcn_readfirstlane(d.x), __builtin_amdgcn_readfirstlane(d.y), __builtin_amdgcn_readfirstlane(d.z),
                      __builtin_amdgcn_readfirstlane(d.w) };
    };
    auto plane_of = [&](const LeafU &d, int &b, int &layer, int &w, int &h, const float *&src) {
        b = d.plane / 3;
        layer = d.plane - 3 * b;
        w = layer == 0 ? g.w[0] : layer == 1 ? g.w[1] : g.w[2];
        h = layer == 0 ? g.h[0] : layer == 1 ? g.h[1] : g.h[2];
        const long long poff = layer == 0 ? g.poff[0] : layer == 1 ? g.poff[1] : g.poff[2];
        src = a.norm + (long long)b * g.pstride + poff;
    };
    LeafU cur = uniform(s_desc[0]), nxt = uniform(s_desc[1]);
    int pb = 0;
    XOffsets<S, NWAVES> xo;
    xo.w = -1;
    if (NXB == 2 && item < count) {
        int b0, l0, w0, h0;
        const float *src0;
        plane_of(cur, b0, l0, w0, h0, src0);
        dct_load_x<S, NWAVES>(src0, w0, h0, g.tiled, make_int4(cur.plane, cur.x, cur.y, cur.coef), sXb, wave, lane, xo);
        wait_vmem_but<0>();
    }
    // quantisers of this lane's outputs (as floats: they are < 2^24), kept in registers while consecutive leaves belong to the same
    // layer (the work lists are ordered by plane, so the layer changes a few hundred times per launch)
    float qf[TPW][16];
    int q_layer = -1;
    bool q_slow = false;               // some quantiser of the layer is too large for the float32 quantiser (never for the codec's own tables)
    for (; item < count; item += step) {
        int b, layer, lw, lhh;
        const float *src;
        plane_of(cur, b, layer, lw, lhh, src);
        const long long coff = layer == 0 ? q.coeff_off[0] : layer == 1 ? q.coeff_off[1] : q.coeff_off[2];
        const long long out_base = (long long)b * q.coeff_stride + coff + cur.coef;
        float *sX = sXb + pb * SS;
        if (S == 128 || layer != q_layer) {   // wave-uniform (the 128 x 128 kernel has no registers to spare: it reloads per leaf)
            // (selected without indexing the kernel-argument array: that would be a dependent kernarg load per leaf)
            const int *qm = layer == 0 ? a.qm[0] : layer == 1 ? a.qm[1] : a.qm[2];
            int qmax = 0;
#pragma unroll
            for (int t = 0; t < TPW; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int qi = qm[((wi0 + t * (NT / TPW)) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * S + J0 + li];
                    qmax = max(qmax, qi);
                    qf[t][r] = (float)qi;
                }
            q_slow = __any(qmax > (1 << 22));
            q_layer = layer;
            if (S != 128) {                       // rare path: leave no load pending across the prefetch below
#pragma unroll
                for (int t = 0; t < TPW; t++)
#pragma unroll
                    for (int r = 0; r < 16; r++) asm volatile("" : "+v"(qf[t][r]));
            }
        }
        if (NXB == 1) {
            dct_load_x<S, NWAVES>(src, lw, lhh, g.tiled, make_int4(cur.plane, cur.x, cur.y, cur.coef), sX, wave, lane, xo);
            wait_vmem_but<0>();
        }
        lds_barrier();                    // X of this leaf has landed: every wave waited for its own pieces before arriving
        if (NXB == 2 && item + step < count) {
            // prefetch the next leaf into the other buffer: it lands under this leaf's MFMA chains
            int bn, ln, wn, hn;
            const float *srcn;
            plane_of(nxt, bn, ln, wn, hn, srcn);
            dct_load_x<S, NWAVES>(srcn, wn, hn, g.tiled, make_int4(nxt.plane, nxt.x, nxt.y, nxt.coef), sXb + (pb ^ 1) * SS, wave, lane, xo);
        }
        // entering a chunk: the chunk before it is finished, its buffer takes the chunk after this one (needed 64 leaves from now)
        if ((k & (kDescChunk - 1)) == 0 && k > 0) load_chunk(k / kDescChunk + 1);
        floatx16 acc[TPW];
#pragma unroll
        for (int t = 0; t < TPW; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
        mfma_chain<S, TPW, kMfmaPF>(sX, (NT / TPW) * 32, wi0 * 32 + li, lh, dreg, acc);
        // accumulator layout: row = (r & 3) + 8 * (r >> 2) + 4 * lh, col = li
#pragma unroll
        for (int t = 0; t < TPW; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                sP[((wi0 + t * (NT / TPW)) * 32 + row) * S + J0 + li] = acc[t][r];
            }
        lds_barrier();

#pragma unroll
        for (int t = 0; t < TPW; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
        mfma_chain<S, TPW, kMfmaPF>(sP, (NT / TPW) * 32, wi0 * 32 + li, lh, dreg, acc);
        int *sQ = reinterpret_cast<int *>(sX);   // every wave finished reading sX before the barrier above
        // one range test per leaf for the float32 quantiser (|y| < 2^17 and q >= 1 give |y / q| < 2^18, its proven range)
        float ymax = 0.f;
#pragma unroll
        for (int t = 0; t < TPW; t++)
#pragma unroll
            for (int r = 0; r < 16; r += 2) ymax = __builtin_fmaxf(__builtin_fmaxf(ymax, __builtin_fabsf(acc[t][r])), __builtin_fabsf(acc[t][r + 1]));
        const bool slow = q_slow || __any(!(ymax < 131072.0f));
        // Most coefficients of a large leaf are far below q / 2: when that holds for the whole wave the result is 0 whatever the
        // quotient's last bits are (0.499f leaves the float32 rounding of the product orders of magnitude of margin) and the
        // quantiser is skipped -- per output register, because vector instructions of this wave only issue while no float32 MFMA
        // of another wave occupies the SIMD (the float32 MFMA runs at exactly the vector rate), so instructions NOT executed
        // are what counts: one vote for all sixteen registers followed by a branch-free quantiser was measured 8 % slower.
#pragma unroll
        for (int t = 0; t < TPW; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int row = (wi0 + t * (NT / TPW)) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (WANT_DCT) a.dct_f32[out_base + row * S + J0 + li] = acc[t][r];
                const float y = acc[t][r], qq = qf[t][r];
                int v = 0;
                if (!__all(__builtin_fabsf(y) < 0.499f * qq)) v = slow ? quantise_f64(y, (int)qq) : quantise_f32(y, qq);
                sQ[zigzag_pos<S>(row, J0 + li)] = v;
            }
        lds_barrier();
        for (int idx = tid * 4; idx < SS; idx += NTHREADS * 4)
            *reinterpret_cast<int4 *>(a.coeffs + out_base + idx) = *reinterpret_cast<const int4 *>(sQ + idx);
        if (NXB == 2) wait_vmem_but<NYOUNG>();   // all but this leaf's stores: in particular the next leaf's X has landed
        else lds_barrier();               // single buffer: the next leaf's DMA must not overwrite sQ before every wave has read it
        k++;
        cur = nxt;
        // the descriptor after next: written to LDS at least one barrier ago (chunk k / 64 + 1 is stored when leaf k's chunk starts)
        nxt = uniform(s_desc[(k + 1) & (2 * kDescChunk - 1)]);
        pb ^= (NXB - 1);
    }
}

// the kernel proper is a shell around the leaf loop above: with the arguments reaching the loop by reference the compiler's schedule of the
// 64 x 64 instantiation needs 128 vector registers instead of 146 (no spills either way), i.e. a 128- instead of a 152-register allocation
template <int S, bool WANT_DCT>
__global__ __launch_bounds__(MfmaCfg<S>::NTHREADS, MfmaCfg<S>::MINW) void k_dct_mfma(Geom g, QtGeom q, DctArgs a, long long max_items, const LeafWork *__restrict__ work)
{
    dct_mfma_leaves<S, WANT_DCT>(g, q, a, max_items, work, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// 64 x 64, one WAVE per leaf: no barrier and no LDS hand-off inside the leaf loop.
//
// The four-wave kernel above spends 4 096 of ~16 000 cycles per leaf and SIMD in its MFMA chains: a float32 MFMA occupies the SIMD's
// vector pipe, so the vector / LDS phases between the chains (P through LDS, quantise + scatter, copy-out) wait for the other waves'
// MFMAs, and every one of them ends in a workgroup barrier that waits for the slowest wave.  Here a wave owns the whole leaf:
//   * chain 1, P = X^T.D^T as four 32 x 32 tiles (wi = block of x, wj = block of u): A = X[2s + lh][32 wi + li] straight from global
//     memory into 64 registers (two 128-byte row segments per load), B = D[32 wj + li][2s + lh] from an operand-ordered LDS table
//     shared by the workgroup; the same operands in the same order as the four-wave kernel, so the sums are bit-identical;
//   * the accumulators ARE chain 2's A operands: the MFMA leaves output row 8g + 4 lh + j in register 4g + j of lane half lh, so the
//     lanes of chain 1's A operand carry the columns of X in the order 8g + 2j + h (for lane 8g + 4h + j) -- register r of a P tile
//     then holds x = 2r + lh, exactly what step r of chain 2 reads (the first version moved the data with eight
//     `v_permlane32_swap` per tile: 32 vector instructions per leaf that each waited for an MFMA of the other wave);
//   * chain 2, Y = T.D^T, tiles (iu, jv): A = the swapped P tiles (s / 16, iu), B = the same LDS operands;
//   * X of the NEXT leaf is requested as soon as chain 1 has consumed the registers, and lands under chain 2 and the epilogue;
//   * epilogue: quantisers and zigzag positions come from LDS tables transposed to [v][u] (a lane's four consecutive rows are one
//     16- / 8-byte read); the first 1 024 zigzag positions go through a 4 KiB per-wave slab and leave as 16-byte stores; the other
//     3 072 are zero in nearly every 64 x 64 leaf (a leaf that large has no edge in it) and are then written as zeros directly --
//     a leaf with a non-zero coefficient up there takes the same slab path once per quarter.
// Two waves per SIMD (<= 256 registers): while one is in its epilogue or waiting for memory the other's MFMAs have the pipe.
// ------------------------------------------------------------------------------------------------
constexpr int kW64Waves = 8;
constexpr int kW64Stride = 68;          // elements per row of the transposed tables: 16-byte aligned rows, conflict-free 16-byte reads
struct __attribute__((aligned(16))) Wave64Lds {
    float dop[2][32][64];               // D operands: [wj][s][lane] = D[32 wj + li][2 s + lh]
    float qT[3][64][kW64Stride];        // quantisers as floats, [layer][v][u]
    unsigned short zT[64][kW64Stride];  // zigzag position of (u, v), [v][u]
    float qlo[3][64][16];               // 0.499 x min of the four quantisers [v][4 g .. 4 g + 3]
    float qlo16[3][64][4];              // ... of the sixteen rows a lane holds of tile row iu: [v][2 iu + lh]
    int slab[kW64Waves][1024];
    LayerTab lt;
    int q_slow[4];
};

// four k-steps of the 2 x 2 tile product: acc[i][j] += A_i(:, 2s .. 2s+1) * B_j(2s .. 2s+1, :), B operands from the LDS table; the
// operands of the NEXT four steps are requested first (scheduling barriers keep that order), so that every wait finds data that was
// requested 16 MFMAs earlier and no more than eight operand registers are live
struct Wave64B { float b[2][4]; };
__device__ __forceinline__ Wave64B wave64_b(const Wave64Lds &L, int s0, int lane)
{
    Wave64B r;
#pragma unroll
    for (int i = 0; i < 4; i++) { r.b[0][i] = L.dop[0][s0 + i][lane]; r.b[1][i] = L.dop[1][s0 + i][lane]; }
    return r;
}

template <bool WANT_DCT>
__global__ __launch_bounds__(kW64Waves * 64, 1) void k_dct64_wave(Geom g, QtGeom q, DctArgs a, long long max_items,
                                                                 const LeafWork *__restrict__ work /* = a.work, read-only: scalar loads */)
{
    constexpr int S = 64;
    __shared__ Wave64Lds L;
    extern __shared__ int s_pref[];      // [nplanes + 1]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;      // (the wave index as a SCALAR: everything derived from it -- the
                                                                                     // item, the leaf, its geometry -- then stays in scalar registers)
    const int li = lane & 31, lh = lane >> 5;
    // tables (once per workgroup; the loops are unrolled so that their global loads are in flight together: left as loops, hipcc issues
    // one load per iteration and waits for it -- two dozen serialised L2 round trips, a leaf and a half of time)
    {
        constexpr int NT = kW64Waves * 64;
        static_assert((2 * 32 * 64) % NT == 0 && (S * S) % NT == 0, "table loops assume whole rounds");
        float dv[2 * 32 * 64 / NT];
#pragma unroll
        for (int i = 0; i < 2 * 32 * 64 / NT; i++) {          // coalesced: consecutive threads read consecutive k of one row of D
            const int idx = tid + i * NT, row = idx >> 6, kk = idx & 63;
            dv[i] = a.D[row * S + kk];
        }
        int qv[3][S * S / NT];
#pragma unroll
        for (int l = 0; l < 3; l++)
#pragma unroll
            for (int i = 0; i < S * S / NT; i++) qv[l][i] = a.qm[l] ? a.qm[l][tid + i * NT] : 1;
#pragma unroll
        for (int i = 0; i < 2 * 32 * 64 / NT; i++) {
            const int idx = tid + i * NT, row = idx >> 6, kk = idx & 63;
            L.dop[row >> 5][kk >> 1][(kk & 1) * 32 + (row & 31)] = dv[i];          // [wj][s][lane] = D[32 wj + li][2 s + lh]
        }
        if (tid < 4) L.q_slow[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int l = 0; l < 3; l++) {
            int qmax = 0;
#pragma unroll
            for (int i = 0; i < S * S / NT; i++) {
                const int idx = tid + i * NT, u = idx >> 6, v = idx & 63;
                qmax = max(qmax, qv[l][i]);
                L.qT[l][v][u] = (float)qv[l][i];
                if (l == 0) L.zT[v][u] = (unsigned short)zigzag_pos<S>(u, v);
            }
            if (qmax > (1 << 22)) L.q_slow[l] = 1;
        }
    }
    __syncthreads();
    // 0.499 x the smallest quantiser of each group of four consecutive rows: "all four coefficients quantise to 0" in one comparison
    for (int idx = tid; idx < 3 * 64 * 16; idx += kW64Waves * 64) {
        const int l = idx >> 10, v = (idx >> 4) & 63, u4 = idx & 15;
        const float *qq = &L.qT[l][v][4 * u4];
        L.qlo[l][v][u4] = 0.499f * __builtin_fminf(__builtin_fminf(qq[0], qq[1]), __builtin_fminf(qq[2], qq[3]));
    }
    __syncthreads();
    for (int idx = tid; idx < 3 * 64 * 4; idx += kW64Waves * 64) {
        const int l = idx >> 8, v = (idx >> 2) & 63, k4 = idx & 3, iu = k4 >> 1, hh = k4 & 1;
        float mn = L.qlo[l][v][8 * iu + hh];
        for (int gq = 1; gq < 4; gq++) mn = __builtin_fminf(mn, L.qlo[l][v][8 * iu + 2 * gq + hh]);
        L.qlo16[l][v][k4] = mn;
    }
    dct_prologue(g, q, a, s_pref, L.lt);       // ends with a barrier
    // (scalar: a value loaded from LDS counts as divergent, and so would every branch on it; 32-bit, because a 64-bit comparison is a
    // vector instruction whose result makes the branch -- and everything defined behind it -- divergent again)
    int count = __builtin_amdgcn_readfirstlane(s_pref[a.nplanes]);
    if ((long long)count > max_items) count = (int)max_items;
    const long long wstride = q.work_stride[a.k];
    const int step = (int)gridDim.x * kW64Waves;
    int item = (int)blockIdx.x * kW64Waves + wave;
    struct LeafU { int plane, x, y, coef; };
    // A wave's items ascend, so the plane of an item is found by walking a running index up the per-plane prefix (usually zero or
    // one step; the binary search of fetch_item costs ~60 vector instructions, and vector instructions outside the MFMA chains are
    // what this kernel has to ration: while the other wave of the SIMD is in a chain they issue once per MFMA)
    int pl = 0;
    auto leaf_at = [&](int it) {
        if (it >= count) return LeafU{ 0, 0, 0, 0 };
        while (pl + 1 < a.nplanes && __builtin_amdgcn_readfirstlane(s_pref[pl + 1]) <= it) pl++;
        pl = __builtin_amdgcn_readfirstlane(pl);        // (the LDS addresses above keep it in a vector register; what follows is scalar arithmetic)
        const int b = pl / 3, l = pl - 3 * b;
        const long long lw = L.lt.woff[l];        // (the LDS copy: indexing the kernel-argument array with a.k would be a dependent global load per leaf)
        const long long woff = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(lw >> 32)) << 32) |
                                           (unsigned)__builtin_amdgcn_readfirstlane((int)lw));
        const long long idx = (long long)b * wstride + woff + (long long)(it - __builtin_amdgcn_readfirstlane(s_pref[pl]));
        const LeafWork d = work[idx];      // a scalar load (counted by lgkmcnt): it does not wait for the vector loads and stores in flight
        const unsigned xy = (unsigned)__builtin_amdgcn_readfirstlane((int)d.xy);
        return LeafU{ pl, (int)(xy & 0xffffu), (int)(xy >> 16), __builtin_amdgcn_readfirstlane(d.coef) };
    };
    float xr[2][32];
    // Lane li of an A operand carries column pli, a permutation inside each group of eight (8 g + 4 h + j -> 8 g + 2 j + h): the MFMA
    // puts row m = 8 g + 4 lh + j of its output into accumulator register 4 g + j of half lh, so with this order register r of a P tile
    // holds x = 2 r + lh -- it IS chain 2's A operand of step r, no data movement in between.  (The sums do not change: a lane's
    // column only decides which output row the same chain of products lands in.)
    const int pli = (li & 24) | ((li & 3) << 1) | ((li >> 2) & 1);
    // X of one leaf into the A-operand registers: xr[wi][s] = X[2 s + lh][32 wi + li], np.pad(reflect) applied to leaves clipped by the
    // plane's border.  (The leaf and its geometry are scalar values -- see `wave` and `count` above -- so the branch between the two
    // forms is a scalar branch, not a pair of masked regions.)
    auto load_x = [&](const LeafU &d) {
        const int b = d.plane / 3, layer = d.plane - 3 * b;
        const int w = layer == 0 ? g.w[0] : layer == 1 ? g.w[1] : g.w[2];
        const int h = layer == 0 ? g.h[0] : layer == 1 ? g.h[1] : g.h[2];
        const long long poff = layer == 0 ? g.poff[0] : layer == 1 ? g.poff[1] : g.poff[2];
        const float *src = a.norm + (long long)b * g.pstride + poff;
        const int hc = min(S, h - d.y), wc = min(S, w - d.x);
        typedef const char __attribute__((address_space(1))) *gbytes;          // (an integer cast to a plain pointer would give flat loads)
        typedef const float __attribute__((address_space(1))) *gfloat;
        unsigned long long base = (unsigned long long)(size_t)(src + plane_elem(g.tiled, w, d.y, d.x));
        const unsigned rowb = (unsigned)w * 4u;
        if (hc == S && wc == S) {
            // the unclipped leaf: scalar row bases, one constant byte offset per lane -- two scalar and two memory instructions per step.
            // Row-major: step s reads rows 2s / 2s + 1 (lane half) at column pli.  Tiled: row 2s + lh is row 2 (s & 1) + lh of block row
            // s >> 1, column pli is element pli & 3 of block pli >> 2; the second tile's columns are 8 blocks further.
            const unsigned off = g.tiled ? (unsigned)(pli >> 2) * 64u + (unsigned)(pli & 3) * 4u + (lh ? 16u : 0u) : (lh ? rowb : 0u) + (unsigned)pli * 4u;
            const unsigned second = g.tiled ? 512u : 128u;
            const unsigned long long even_step = g.tiled ? 32ull : 2ull * rowb, odd_step = g.tiled ? 4ull * rowb - 32ull : 2ull * rowb;
#pragma unroll
            for (int s = 0; s < 32; s++) {
                // (readfirstlane pins the row base to scalar registers: without it the compiler adds the lane offset first and carries
                // 32 64-bit vector addresses)
                const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)base), bhi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
                const gbytes rowp = (gbytes)(size_t)(((unsigned long long)bhi << 32) | blo);
                xr[0][s] = *(gfloat)(rowp + off);
                xr[1][s] = *(gfloat)(rowp + (off + second));
                base += (s & 1) ? odd_step : even_step;
            }
        } else {
            // clipped by the plane's border (the last row / column of leaves of a plane): np.pad(reflect) row by row
            const int cc0 = reflect_pad_idx(pli, wc), cc1 = reflect_pad_idx(32 + pli, wc);
            int r = 0, dir = hc > 1 ? 1 : 0;
            auto advance = [&]() {
                r += dir;
                if (r == hc - 1 && dir > 0) dir = -1;
                else if (r == 0 && dir < 0) dir = 1;
            };
#pragma unroll
            for (int s = 0; s < 32; s++) {
                const int r0 = r;
                advance();
                const int r1 = r;
                advance();
                const int rr = lh ? r1 : r0;
                const gfloat plane = (gfloat)(size_t)src;
                xr[0][s] = plane[plane_elem(g.tiled, w, d.y + rr, d.x + cc0)];
                xr[1][s] = plane[plane_elem(g.tiled, w, d.y + rr, d.x + cc1)];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    if (item >= count) return;
    LeafU cur = leaf_at(item);
    load_x(cur);
    int *slab = L.slab[wave];
    for (; item < count; item += step) {
        const LeafU nxt = leaf_at(item + step);
        const int b = cur.plane / 3, layer = cur.plane - 3 * b;
        const long long coff = layer == 0 ? q.coeff_off[0] : layer == 1 ? q.coeff_off[1] : q.coeff_off[2];
        const long long out_base = (long long)b * q.coeff_stride + coff + cur.coef;
        // ---- chain 1: P tiles [wi][wj]
        floatx16 P[2][2];
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) P[t >> 1][t & 1][r] = 0.f;
        {
            Wave64B bc = wave64_b(L, 0, lane), bn = bc;
#pragma unroll
            for (int s0 = 0; s0 < 32; s0 += 4) {
                if (s0 + 4 < 32) bn = wave64_b(L, s0 + 4, lane);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    P[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[0][s0 + i], bc.b[0][i], P[0][0], 0, 0, 0);
                    P[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[0][s0 + i], bc.b[1][i], P[0][1], 0, 0, 0);
                    P[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[1][s0 + i], bc.b[0][i], P[1][0], 0, 0, 0);
                    P[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[1][s0 + i], bc.b[1][i], P[1][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                bc = bn;
            }
        }
        // the registers of X are free: the next leaf's pixels fly while this one is finished
        if (item + step < count) load_x(nxt);
        __builtin_amdgcn_sched_barrier(0);
        // (P[wi][wj][r] is chain 2's A operand of step 16 wi + r: see pli)
        // ---- chain 2: Y tiles [iu][jv]
        floatx16 Y[2][2];
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) Y[t >> 1][t & 1][r] = 0.f;
        {
            Wave64B bc = wave64_b(L, 0, lane), bn = bc;
#pragma unroll
            for (int s0 = 0; s0 < 32; s0 += 4) {
                if (s0 + 4 < 32) bn = wave64_b(L, s0 + 4, lane);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int s = s0 + i;
                    const float a0 = P[s >> 4][0][s & 15], a1 = P[s >> 4][1][s & 15];
                    Y[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bc.b[0][i], Y[0][0], 0, 0, 0);
                    Y[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bc.b[1][i], Y[0][1], 0, 0, 0);
                    Y[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bc.b[0][i], Y[1][0], 0, 0, 0);
                    Y[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bc.b[1][i], Y[1][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                bc = bn;
            }
        }
        // ---- epilogue.  Accumulator register 4 gq + j of tile (iu, jv): u = 32 iu + 8 gq + 4 lh + j, v = 32 jv + li.
        // A group (tile, gq) whose smallest diagonal 32 iu + 32 jv + 8 gq is 45 or more lies entirely beyond zigzag position 1 024.
        const float (*qT)[kW64Stride] = L.qT[layer];
        const float (*qlo)[16] = L.qlo[layer];
        // One vote per group of four rows: "every lane's four coefficients quantise to 0" (|y| < 0.499 x the smallest of the four
        // quantisers: the float32 rounding of the product has orders of magnitude of margin).  True for all but a few groups of a
        // leaf this large; the first quarter of the output is zeroed up front and only non-zero values are written into it.
        bool redo = L.q_slow[layer] != 0;   // wave-uniform: some value needs the float64 quantiser (never with the codec's own tables and image-range input)
        bool upper = false;                 // wave-uniform: a non-zero coefficient at zigzag position >= 1 024
#pragma unroll
        for (int c = 0; c < 4; c++) reinterpret_cast<int4 *>(slab)[c * 64 + lane] = make_int4(0, 0, 0, 0);
        int hi_lane = 0;                    // per lane: a non-zero value beyond position 1 024, or one outside the float32 quantiser's range
        if (WANT_DCT) {
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    a.dct_f32[out_base + (32 * (t >> 1) + 8 * (r >> 2) + 4 * lh + (r & 3)) * S + 32 * (t & 1) + li] = Y[t >> 1][t & 1][r];
        }
        // (every vote is a round trip vector -> scalar -> branch that the other wave's MFMAs stretch to hundreds of cycles: as few as possible)
        auto tile_max = [&](int iu, int jv) {
            float m16 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) m16 = __builtin_fmaxf(__builtin_fmaxf(m16, __builtin_fabsf(Y[iu][jv][r])), __builtin_fabsf(Y[iu][jv][r + 1]));
            return m16;
        };
        auto group_max = [&](int iu, int jv, int gq) {
            return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(Y[iu][jv][4 * gq]), __builtin_fabsf(Y[iu][jv][4 * gq + 1])),
                                   __builtin_fmaxf(__builtin_fabsf(Y[iu][jv][4 * gq + 2]), __builtin_fabsf(Y[iu][jv][4 * gq + 3])));
        };
        auto quantise_group = [&](int iu, int jv, int gq, float m) {          // a group with a coefficient that does not quantise to 0
            const int u0 = 32 * iu + 8 * gq + 4 * lh, v = 32 * jv + li;
            if (32 * iu + 32 * jv + 8 * gq >= 45) {        // the whole group lies beyond position 1 024 (smallest diagonal >= 45)
                upper = true;
                return;
            }
            const float4 q4 = *reinterpret_cast<const float4 *>(&qT[v][u0]);
            const uint2 z2 = *reinterpret_cast<const uint2 *>(&L.zT[v][u0]);
            const float qq[4] = { q4.x, q4.y, q4.z, q4.w };
            const int zz[4] = { (int)(z2.x & 0xffffu), (int)(z2.x >> 16), (int)(z2.y & 0xffffu), (int)(z2.y >> 16) };
            hi_lane |= !(m < 131072.0f);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int val = quantise_f32(Y[iu][jv][4 * gq + j], qq[j]);
                if (val != 0) {
                    if (zz[j] < 1024) slab[zz[j]] = val;
                    else hi_lane |= 2;
                }
            }
        };
        auto groups_of = [&](int iu, int jv, int first) {                     // the groups of a tile, one vote each
#pragma unroll
            for (int gq = first; gq < 4; gq++) {
                const float m = group_max(iu, jv, gq);
                if (!__all(m < qlo[32 * jv + li][(32 * iu + 8 * gq + 4 * lh) >> 2])) quantise_group(iu, jv, gq, m);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // tile (0, 0): its first group holds the DC coefficient -- no vote; its other three groups share one
        quantise_group(0, 0, 0, group_max(0, 0, 0));
        {
            bool zero = true;
#pragma unroll
            for (int gq = 1; gq < 4; gq++) zero = zero && group_max(0, 0, gq) < qlo[li][(8 * gq + 4 * lh) >> 2];
            if (!__all(zero)) groups_of(0, 0, 1);
        }
        // the other three tiles share one vote
        {
            const bool zero = tile_max(0, 1) < L.qlo16[layer][32 + li][lh] && tile_max(1, 0) < L.qlo16[layer][li][2 + lh] &&
                              tile_max(1, 1) < L.qlo16[layer][32 + li][2 + lh];
            if (!__all(zero)) {
#pragma unroll
                for (int t = 1; t < 4; t++)
                    if (!__all(tile_max(t >> 1, t & 1) < L.qlo16[layer][32 * (t & 1) + li][2 * (t >> 1) + lh])) groups_of(t >> 1, t & 1, 0);
            }
        }
        if (__any(hi_lane & 1)) redo = true;
        if (__any(hi_lane & 2)) upper = true;
        // (LDS operations of one wave execute in order: the reads below see this wave's writes, the next writes come after them)
#pragma unroll
        for (int c = 0; c < 4; c++)
            reinterpret_cast<int4 *>(a.coeffs + out_base)[c * 64 + lane] = reinterpret_cast<const int4 *>(slab)[c * 64 + lane];
        if (!upper && !redo) {
#pragma unroll
            for (int c = 0; c < 12; c++) reinterpret_cast<int4 *>(a.coeffs + out_base + 1024)[c * 64 + lane] = make_int4(0, 0, 0, 0);
        } else {
            // rare: every coefficient beyond position 1 024 (every coefficient, if the float64 quantiser is needed) straight to its place.
            // A tile at a time through the slab, so that the loop over its registers need not be unrolled.
            if (redo) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the stores above have landed before other lanes overwrite them
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int iu = t >> 1, jv = t & 1;
#pragma unroll
                for (int r = 0; r < 16; r++) reinterpret_cast<float *>(slab)[r * 64 + lane] = Y[iu][jv][r];
#pragma unroll 1
                for (int r = 0; r < 16; r++) {
                    const float y = reinterpret_cast<const float *>(slab)[r * 64 + lane];
                    const int u = 32 * iu + 8 * (r >> 2) + 4 * lh + (r & 3), v = 32 * jv + li;
                    const float qq = qT[v][u];
                    const int zz = L.zT[v][u];
                    const bool slow = redo && (L.q_slow[layer] != 0 || __any(!(__builtin_fabsf(y) < 131072.0f)));
                    const int val = slow ? quantise_f64(y, (int)qq) : quantise_f32(y, qq);
                    if (redo || zz >= 1024) a.coeffs[out_base + zz] = val;
                }
            }
        }
        cur = nxt;
    }
}

template <bool WANT_DCT>
static void launch_dct64_wave(hipStream_t st, const Geom &g, const QtGeom &q, const DctArgs &a, long long max_items)
{
    const size_t pref = (size_t)(a.nplanes + 1) * sizeof(int);
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    const long long want = (max_items + kW64Waves - 1) / kW64Waves;
    const int blocks = (int)(want < 1 ? 1 : want > cus ? cus : want);
    hipLaunchKernelGGL((k_dct64_wave<WANT_DCT>), dim3(blocks), dim3(kW64Waves * 64), pref, st, g, q, a, max_items,
                       a.work);
}

// ------------------------------------------------------------------------------------------------
// stand-alone entry: leaf table -> per-size work lists
// ------------------------------------------------------------------------------------------------
struct WorkPtrs { LeafWork *w[kMaxSizes]; };

__global__ __launch_bounds__(256) void k_work_from_leaves(const int *__restrict__ leaves, long long n, int bmin_log2, int plane, WorkPtrs wp,
                                                          int *__restrict__ work_count)
{
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int4 lf = reinterpret_cast<const int4 *>(leaves)[i];
    int k = (31 - __clz(lf.z)) - bmin_log2;
    if (k < 0 || k >= kMaxSizes || !wp.w[k]) return;
    int pos = atomicAdd(&work_count[plane * kMaxSizes + k], 1);
    wp.w[k][pos] = pack_work(lf.x, lf.y, lf.w);
}

void launch_work_from_leaves(hipStream_t st, const int *leaves, long long n, int bmin, int plane, LeafWork *const *work, int *work_count)
{
    WorkPtrs wp;
    for (int k = 0; k < kMaxSizes; k++) wp.w[k] = work[k];
    if (n <= 0) return;
    hipLaunchKernelGGL(k_work_from_leaves, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, leaves, n, ilog2(bmin), plane, wp, work_count);
}

// The MFMA kernels are persistent (a workgroup walks the work list with a grid stride), so the grid must not exceed what is
// resident at once -- a second, partly filled round of workgroups would run alone at the end (4096 workgroups on 3072 slots:
// 67 % efficiency).  The residency comes from the occupancy query for the actual dynamic-LDS size.
template <int S, bool WANT_DCT>
static void launch_mfma_t(hipStream_t st, const Geom &g, const QtGeom &q, const DctArgs &a, long long max_items)
{
    size_t lds = (size_t)(MfmaCfg<S>::NXB + 1) * S * S * sizeof(float) + 2 * kDescChunk * sizeof(int4) + sizeof(LayerTab) + 8 +
                 (size_t)(a.nplanes + 1) * sizeof(int);
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dct_mfma<S, WANT_DCT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    static size_t occ_lds = ~(size_t)0;
    static int slots = 0;
    if (lds != occ_lds) {
        int per_cu = 0, dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_dct_mfma<S, WANT_DCT>, MfmaCfg<S>::NTHREADS, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        slots = per_cu * cus;
        occ_lds = lds;
    }
    const int blocks = (int)(max_items < slots ? (max_items < 1 ? 1 : max_items) : slots);
    hipLaunchKernelGGL((k_dct_mfma<S, WANT_DCT>), dim3(blocks), dim3(MfmaCfg<S>::NTHREADS), lds, st, g, q, a, max_items,
                       a.work);
}

// ------------------------------------------------------------------------------------------------
// Latency-sized calls (one image: the GUI's use, src/gui/main_frame.py:145): the five block sizes 4 .. 64 in ONE launch.  Launched one
// after the other their times add up -- 58 of the 204 us of a 1080p call, each kernel a short chain of dependent steps on a mostly idle
// chip (profiles/r05_latency_chain.txt) -- while side by side they take as long as the longest.  The grid is the five kernels' grids end to
// end; a workgroup runs the body of the size its index falls into (same code, same results: the bodies are shared with the per-size
// kernels).  The 32 x 32 body is a one-wave workgroup: the other three waves of its workgroups leave at once.
// ------------------------------------------------------------------------------------------------
struct DctMulti {
    DctArgs a[5];              // sizes 4, 8, 16, 32, 64
    long long max_items[5];
    int first[6];              // first workgroup of each size's share ([5] = grid size); an absent size has an empty share
};

__global__ __launch_bounds__(256) void k_dct_multi(Geom g, QtGeom q, DctMulti m)
{
    const unsigned b = blockIdx.x;
    if (b < (unsigned)m.first[1]) dct4_body<false>(g, q, m.a[0], m.max_items[0], b - m.first[0], m.first[1] - m.first[0]);
    else if (b < (unsigned)m.first[2]) dct8_body<false>(g, q, m.a[1], m.max_items[1], b - m.first[1], m.first[2] - m.first[1]);
    else if (b < (unsigned)m.first[3]) dct16_body<false>(g, q, m.a[2], m.max_items[2], b - m.first[2], m.first[3] - m.first[2]);
    else if (b < (unsigned)m.first[4]) {
        if (threadIdx.x < MfmaCfg<32>::NTHREADS) dct_mfma_leaves<32, false>(g, q, m.a[3], m.max_items[3], m.a[3].work, b - m.first[3], m.first[4] - m.first[3]);
    } else dct_mfma_leaves<64, false>(g, q, m.a[4], m.max_items[4], m.a[4].work, b - m.first[4], m.first[5] - m.first[4]);
}

// args[k] / max_items[k] for size bmin << k; returns 0 when the launch was made, 1 when this call is not one for it (the caller then
// launches size by size)
int launch_dct_multi(hipStream_t st, const Geom &g, const QtGeom &q, const DctArgs *args, const long long *max_items)
{
    if (q.bmin < 4 || q.bmax > 64 || args[0].dct_f32 || args[0].nplanes > kMaxPlanes) return 1;
    DctMulti m;
    int nb = 0;
    size_t lds = 0;
    for (int slot = 0; slot < 5; slot++) {
        const int size = 4 << slot;
        m.first[slot] = nb;
        if (size < q.bmin || size > q.bmax) { m.a[slot] = args[0]; m.max_items[slot] = 0; continue; }
        const int k = ilog2(size) - ilog2(q.bmin);
        m.a[slot] = args[k];
        m.max_items[slot] = max_items[k];
        if (max_items[k] <= 0) continue;
        auto cap = [&](long long per_block, int hi) { const long long bl = (max_items[k] + per_block - 1) / per_block; return (int)(bl < 1 ? 1 : bl > hi ? hi : bl); };
        const size_t pref = (size_t)(args[k].nplanes + 1) * sizeof(int);
        // (the merged kernel holds the registers and the LDS of its largest body -- two workgroups per CU: the shares are sized so that the
        // whole grid is resident at once, 512 workgroups, roughly in proportion to the time the sizes take; every body walks its list with
        // a grid stride)
        if (size == 4) { nb += cap(64 * 8, 48); lds = std::max(lds, pref); }
        else if (size == 8) { nb += cap(32, 64); lds = std::max(lds, pref); }
        else if (size == 16) { nb += cap(4, 112); lds = std::max(lds, pref); }
        else {
            const size_t need = size == 32 ? (size_t)(MfmaCfg<32>::NXB + 1) * 32 * 32 * sizeof(float) : (size_t)(MfmaCfg<64>::NXB + 1) * 64 * 64 * sizeof(float);
            lds = std::max(lds, need + 2 * kDescChunk * sizeof(int4) + sizeof(LayerTab) + 8 + pref);
            nb += cap(1, size == 32 ? 96 : 192);
        }
    }
    m.first[5] = nb;
    if (nb == 0) return 0;
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dct_multi), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    hipLaunchKernelGGL(k_dct_multi, dim3((unsigned)nb), dim3(256), lds, st, g, q, m);
    return 0;
}

int launch_dct(hipStream_t st, int size, const Geom &g, const QtGeom &q, const DctArgs &a, long long max_items, const Tuning &t)
{
    if (max_items <= 0) return 0;                  // an empty work list is not an error
    if (a.nplanes > kMaxPlanes) return -1;         // the per-plane prefix table would not fit the launch's LDS
    const size_t pref = (size_t)(a.nplanes + 1) * sizeof(int);
    const int small_wgs = t.dct_small_workgroups;      // tuning (aej_set_option): residency of the grid-stride kernels
    auto cap = [&](long long per_block, int hi) {
        long long b = (max_items + per_block - 1) / per_block;
        if (small_wgs > 0 && size <= 16) hi = small_wgs;
        return (int)(b < 1 ? 1 : b > hi ? hi : b);
    };
    const bool wd = a.dct_f32 != nullptr;
#define AEJ_SMALL(S, PER, HI)                                                                                              \
    if (wd) hipLaunchKernelGGL((k_dct_small<S, true>), dim3(cap(PER, HI)), dim3(256), pref, st, g, q, a, max_items);       \
    else hipLaunchKernelGGL((k_dct_small<S, false>), dim3(cap(PER, HI)), dim3(256), pref, st, g, q, a, max_items)
#define AEJ_MFMA(S)                                                     \
    if (wd) launch_mfma_t<S, true>(st, g, q, a, max_items);              \
    else launch_mfma_t<S, false>(st, g, q, a, max_items)
    switch (size) {
    case 2: AEJ_SMALL(2, 128, 2048); break;
    case 4:
        if (wd) hipLaunchKernelGGL((k_dct4<true>), dim3(cap(64 * 8, 8192)), dim3(256), pref, st, g, q, a, max_items);
        else hipLaunchKernelGGL((k_dct4<false>), dim3(cap(64 * 8, 8192)), dim3(256), pref, st, g, q, a, max_items);
        break;
    case 8:
        if (wd) hipLaunchKernelGGL((k_dct8_shfl<true>), dim3(cap(32, 4096)), dim3(256), pref, st, g, q, a, max_items);
        else hipLaunchKernelGGL((k_dct8_shfl<false>), dim3(cap(32, 4096)), dim3(256), pref, st, g, q, a, max_items);
        break;
    case 16:
        if (wd) hipLaunchKernelGGL((k_dct16_mfma<true>), dim3(cap(4, 2048)), dim3(256), pref, st, g, q, a, max_items);
        else hipLaunchKernelGGL((k_dct16_mfma<false>), dim3(cap(4, 2048)), dim3(256), pref, st, g, q, a, max_items);
        break;
    case 32: AEJ_MFMA(32); break;
    case 64: {
        // Two kernels.  One wave per leaf (k_dct64_wave) is the faster one on an otherwise idle device (0.79 against 0.86 ms for 64 x 4K), but
        // its workgroups take a whole CU each (122 KiB LDS, 2 x 242 registers per SIMD lane): beside the kernels of other sub-batches it
        // waits for CUs to drain and then shares them with nobody -- the pipelined 64 x 4K step is 6.70 ms with it and 6.45 ms with the
        // four-wave kernel, whose workgroups are a third of a CU.  So the caller says whether the launch will have company.
        const bool force_four = t.dct64_kernel == 4, force_wave = t.dct64_kernel == 1;      // aej_set_option "dct64_kernel"
        // (latency-sized calls -- at most a few hundred leaves can exist: the one-wave kernel's workgroup first fills 122 KiB of LDS tables,
        // 0.038 against 0.027 ms for one 1080p image)
        const bool tiny = max_items < 2048;
        if (wd || force_four || ((a.crowded || tiny) && !force_wave)) { AEJ_MFMA(64); }       // (the float32 DCT output, a debugging aid, stays with the four-wave kernel)
        else launch_dct64_wave<false>(st, g, q, a, max_items);
        break;
    }
    case 128: AEJ_MFMA(128); break;
#define AEJ_BIG(S)                                                                                                               \
    if (!a.scratch) return -1;   /* callers reserve it whenever the settings allow this size */                                 \
    if (wd) hipLaunchKernelGGL((k_dct_big<S, true>), dim3(cap(1, big_blocks(S))), dim3(256), pref, st, g, q, a, max_items);      \
    else hipLaunchKernelGGL((k_dct_big<S, false>), dim3(cap(1, big_blocks(S))), dim3(256), pref, st, g, q, a, max_items)
    case 256: AEJ_BIG(256); break;
    case 512: AEJ_BIG(512); break;
    case 1024: AEJ_BIG(1024); break;
#undef AEJ_BIG
    default: return -1;
    }
#undef AEJ_SMALL
#undef AEJ_MFMA
    return 0;
}

}  // namespace aej
