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
__device__ __forceinline__ int quantise(float y, int q)
{
    double v = (double)y / (double)q;
    return (int)rint(v);
}

// The same for the large (MFMA) blocks, where most coefficients are far below q / 2 and the result is 0 whatever the
// quotient's last bits are: when that holds for the whole wave the float64 division (about 15 VALU instructions) is skipped.
// 0.499f leaves the float32 rounding of the product and of q orders of magnitude of margin.  (Not used for blocks <= 16:
// their waves rarely qualify and the test costs more than it saves, measured.)
__device__ __forceinline__ int quantise_sparse(float y, int q)
{
    const bool tiny = fabsf(y) < 0.499f * (float)q;
    if (__all(tiny)) return 0;
    return quantise(y, q);
}

// Work items of one block size are the concatenation, over planes (b, l), of that plane's Morton-ordered leaf list.
// s_pref[p] = number of items in planes < p (built once per workgroup by wave 0); an item index is mapped back to
// (plane, index in plane) with a binary search.  Per-layer geometry is copied to LDS once so that the per-leaf code
// never indexes kernel arguments dynamically (that would be a dependent kernarg load per leaf).
struct LayerTab {
    int w[3], h[3];
    long long poff[3], coff[3], woff[3];
};

__device__ __forceinline__ int wave_incl_scan_i(int v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

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
            carry += __shfl(inc, 63);
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
    return reinterpret_cast<const int4 *>(a.work)[idx];
}

// Same mapping for a caller whose item indices only grow: scan forward from the plane of the previous item (`p`, updated)
// instead of bisecting -- usually one or two LDS reads on the critical path instead of eight dependent ones.
__device__ __forceinline__ int4 fetch_item_fwd(const DctArgs &a, long long work_stride, const LayerTab &lt, const int *s_pref, long long item, int &p)
{
    while (p + 1 < a.nplanes && (long long)s_pref[p + 1] <= item) p++;
    const int b = p / 3, l = p - 3 * b;
    const long long idx = (long long)b * work_stride + lt.woff[l] + (item - s_pref[p]);
    return reinterpret_cast<const int4 *>(a.work)[idx];
}

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
            for (int k = 0; k < S; k++) x[k] = src[(long long)(cur.z + reflect_pad_idx(k, hc)) * w + col];
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
#pragma unroll
            for (int jj = 0; jj < S; jj++) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < S; k++) acc = __builtin_fmaf(t[k], sD[jj * S + k], acc);
                const int ridx = j * S + jj;
                if (WANT_DCT) a.dct_f32[out_base + ridx] = acc;
                sQ[slot * SS + sZ[ridx]] = quantise(acc, sQm[layer * SS + ridx]);
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
// 4 x 4 blocks (the most numerous leaves): one THREAD per leaf, everything in registers.  Four 16-byte row loads, both
// products as the same k-ordered fma chains, 16 quotients, zigzag as a compile-time permutation, four 16-byte stores
// (consecutive leaves of a Morton-ordered list have consecutive coefficient offsets, so a wave writes 4 KiB contiguously).
// No LDS transpose, no barriers; 25 instead of 61 VALU instructions per pixel.
// ------------------------------------------------------------------------------------------------
template <bool WANT_DCT>
__global__ __launch_bounds__(256) void k_dct4(Geom g, QtGeom q, DctArgs a, long long max_items)
{
    __shared__ float sD[16];
    __shared__ int sQm[3 * 16];
    __shared__ LayerTab lt;
    extern __shared__ int s_pref[];
    const int tid = threadIdx.x;
    if (tid < 16) sD[tid] = a.D[tid];
    if (tid < 48) sQm[tid] = a.qm[tid / 16] ? a.qm[tid / 16][tid % 16] : 1;
    dct_prologue(g, q, a, s_pref, lt);       // ends with a barrier
    long long count = s_pref[a.nplanes];
    if (count > max_items) count = max_items;
    const long long wstride = q.work_stride[a.k];
    float D[4][4];
#pragma unroll
    for (int i = 0; i < 16; i++) D[i >> 2][i & 3] = sD[i];
    for (long long item = (long long)blockIdx.x * 256 + tid; item < count; item += (long long)gridDim.x * 256) {
        const int4 cur = fetch_item(a, wstride, lt, s_pref, item);
        const int b = cur.x / 3, layer = cur.x - b * 3;
        const int w = lt.w[layer], h = lt.h[layer];
        const float *src = a.norm + (long long)b * g.pstride + lt.poff[layer];
        const int hc = min(4, h - cur.z), wc = min(4, w - cur.y);
        float x[4][4];
        if (hc == 4 && wc == 4 && (w & 3) == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float4 v = *reinterpret_cast<const float4 *>(src + (long long)(cur.z + r) * w + cur.y);
                x[r][0] = v.x; x[r][1] = v.y; x[r][2] = v.z; x[r][3] = v.w;
            }
        } else {                              // clipped at the plane border (np.pad reflect) or unaligned rows
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int c = 0; c < 4; c++)
                    x[r][c] = src[(long long)(cur.z + reflect_pad_idx(r, hc)) * w + cur.y + reflect_pad_idx(c, wc)];
        }
        float T[4][4];                        // T[i][j] = sum_k D[i][k] X[k][j]
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 4; k++) acc = __builtin_fmaf(D[i][k], x[k][j], acc);
                T[i][j] = acc;
            }
        const long long out_base = (long long)b * q.coeff_stride + lt.coff[layer] + cur.w;
        int out[16];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {   // Y[i][jj] = sum_k T[i][k] D[jj][k]
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 4; k++) acc = __builtin_fmaf(T[i][k], D[jj][k], acc);
                if (WANT_DCT) a.dct_f32[out_base + i * 4 + jj] = acc;
                out[zigzag_pos<4>(i, jj)] = quantise(acc, sQm[layer * 16 + i * 4 + jj]);
            }
        int4 *dst = reinterpret_cast<int4 *>(a.coeffs + out_base);     // offsets are sums of squares of sizes >= 2: multiples of 4
#pragma unroll
        for (int r = 0; r < 4; r++) dst[r] = make_int4(out[4 * r], out[4 * r + 1], out[4 * r + 2], out[4 * r + 3]);
    }
}

// ------------------------------------------------------------------------------------------------
// S = 256 (aej_bigblock.h): T = D.X into this workgroup's scratch, then Y = T.D^T, quantise, zigzag scatter
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
            [&](int k, int j) { return src[(long long)(cur.z + reflect_pad_idx(k, hc)) * w + cur.y + reflect_pad_idx(j, wc)]; },
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

// LDS-DMA (global_load_lds): per-lane global source, LDS destination = wave-uniform base + lane * size.
__device__ __forceinline__ void glds16(const float *g, float *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const float *g, float *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 4, 0, 0);
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
};

// X -> LDS by LDS-DMA (no VGPR staging): full leaves 1 KiB (256 floats) per wave instruction, clipped or unaligned ones one
// dword per lane with the np.pad(reflect) index map applied to the source address
template <int S, int NWAVES>
__device__ __forceinline__ void dct_load_x(const float *src, int w, int h, const int4 &d, float *sX, int wave, int lane)
{
    constexpr int SS = S * S;
    const int hc = min(S, h - d.z), wc = min(S, w - d.y);
    if (hc == S && wc == S && (w & 3) == 0) {
        constexpr int ROWS = 256 / S;                 // rows per wave instruction
#pragma unroll
        for (int t = 0; t < SS / 256 / NWAVES; t++) {
            const int chunk = t * NWAVES + wave;
            const int r = chunk * ROWS + lane / (S / 4), c = (lane % (S / 4)) * 4;
            glds16(src + (long long)(d.z + r) * w + d.y + c, sX + chunk * 256);
        }
    } else {
#pragma unroll 4
        for (int t = 0; t < SS / 64 / NWAVES; t++) {
            const int chunk = t * NWAVES + wave;
            const int idx = chunk * 64 + lane;
            const int r = idx / S, c = idx - r * S;
            glds4(src + (long long)(d.z + reflect_pad_idx(r, hc)) * w + d.y + reflect_pad_idx(c, wc), sX + chunk * 64);
        }
    }
}

template <int S, bool WANT_DCT>
__global__ __launch_bounds__(MfmaCfg<S>::NTHREADS, MfmaCfg<S>::MINW) void k_dct_mfma(Geom g, QtGeom q, DctArgs a, long long max_items)
{
    using C = MfmaCfg<S>;
    constexpr int NT = C::NT, TPW = C::TPW, NWAVES = C::NWAVES, NTHREADS = C::NTHREADS;
    constexpr int SS = S * S;
    constexpr int NXB = S == 64 ? 2 : 1;          // X double-buffered (next leaf prefetched by LDS-DMA): pays only for 64x64 leaves
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sXb = smem;                 // [NXB][S][S]  X, the current one later reused as int staging for the zigzag scatter
    float *sP = smem + NXB * SS;       // [S][S]  P = T^T
    LayerTab &lt = *reinterpret_cast<LayerTab *>(smem + (NXB + 1) * SS);
    int *s_pref = reinterpret_cast<int *>(smem + (NXB + 1) * SS) + (sizeof(LayerTab) + 3) / 4;   // [nplanes + 1]
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
    long long count = s_pref[a.nplanes];
    if (count > max_items) count = max_items;
    const long long wstride = q.work_stride[a.k];
    const long long step = gridDim.x;
    long long item = blockIdx.x;
    int4 cur = make_int4(0, 0, 0, 0), nxt = make_int4(0, 0, 0, 0);
    int plane_hint = 0;
    if (item < count) cur = fetch_item_fwd(a, wstride, lt, s_pref, item, plane_hint);
    if (item + step < count) nxt = fetch_item_fwd(a, wstride, lt, s_pref, item + step, plane_hint);
    int pb = 0;
    if (NXB == 2 && item < count) {
        const int b0 = cur.x / 3, l0 = cur.x - b0 * 3;
        dct_load_x<S, NWAVES>(a.norm + (long long)b0 * g.pstride + lt.poff[l0], lt.w[l0], lt.h[l0], cur, sXb, wave, lane);
    }
    for (; item < count; item += step) {
        const int b = cur.x / 3, layer = cur.x - b * 3;
        const long long out_base = (long long)b * q.coeff_stride + lt.coff[layer] + cur.w;
        const int *qm = a.qm[layer];
        float *sX = sXb + pb * SS;
        if (NXB == 1) dct_load_x<S, NWAVES>(a.norm + (long long)b * g.pstride + lt.poff[layer], lt.w[layer], lt.h[layer], cur, sX, wave, lane);
        __syncthreads();                  // X of this leaf has landed (the barrier drains the LDS-DMA queue)
        if (NXB == 2 && item + step < count) {
            // prefetch the next leaf into the other buffer: it lands under this leaf's first MFMA chain
            const int bn = nxt.x / 3, ln = nxt.x - bn * 3;
            dct_load_x<S, NWAVES>(a.norm + (long long)bn * g.pstride + lt.poff[ln], lt.w[ln], lt.h[ln], nxt, sXb + (pb ^ 1) * SS, wave, lane);
        }
        int4 nn = make_int4(0, 0, 0, 0);
        if (item + 2 * step < count) nn = fetch_item_fwd(a, wstride, lt, s_pref, item + 2 * step, plane_hint);

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
        __syncthreads();

        // quantisers of this lane's outputs: loaded here so that their latency hides under the second MFMA chain
        int qv[TPW][16];
#pragma unroll
        for (int t = 0; t < TPW; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) qv[t][r] = qm[((wi0 + t * (NT / TPW)) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * S + J0 + li];
#pragma unroll
        for (int t = 0; t < TPW; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
        mfma_chain<S, TPW, kMfmaPF>(sP, (NT / TPW) * 32, wi0 * 32 + li, lh, dreg, acc);
        int *sQ = reinterpret_cast<int *>(sX);   // every wave finished reading sX before the barrier above
#pragma unroll
        for (int t = 0; t < TPW; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int row = (wi0 + t * (NT / TPW)) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (WANT_DCT) a.dct_f32[out_base + row * S + J0 + li] = acc[t][r];
                sQ[zigzag_pos<S>(row, J0 + li)] = quantise_sparse(acc[t][r], qv[t][r]);
            }
        __syncthreads();
        for (int idx = tid * 4; idx < SS; idx += NTHREADS * 4)
            *reinterpret_cast<int4 *>(a.coeffs + out_base + idx) = *reinterpret_cast<const int4 *>(sQ + idx);
        if (NXB == 1) __syncthreads();    // single buffer: the next leaf's DMA must not overwrite sQ before it is copied out
        cur = nxt; nxt = nn; pb ^= (NXB - 1);
    }
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
    reinterpret_cast<int4 *>(wp.w[k])[pos] = make_int4(plane, lf.x, lf.y, lf.w);
}

void launch_work_from_leaves(hipStream_t st, const int *leaves, long long n, int bmin, int plane, LeafWork *const *work, int *work_count)
{
    WorkPtrs wp;
    for (int k = 0; k < kMaxSizes; k++) wp.w[k] = work[k];
    if (n <= 0) return;
    hipLaunchKernelGGL(k_work_from_leaves, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, leaves, n, ilog2(bmin), plane, wp, work_count);
}

template <int S, bool WANT_DCT>
static void launch_mfma_t(hipStream_t st, const Geom &g, const QtGeom &q, const DctArgs &a, long long max_items, int blocks)
{
    size_t lds = (size_t)(S == 64 ? 3 : 2) * S * S * sizeof(float) + sizeof(LayerTab) + 8 + (size_t)(a.nplanes + 1) * sizeof(int);
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dct_mfma<S, WANT_DCT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    hipLaunchKernelGGL((k_dct_mfma<S, WANT_DCT>), dim3(blocks), dim3(MfmaCfg<S>::NTHREADS), lds, st, g, q, a, max_items);
}

int launch_dct(hipStream_t st, int size, const Geom &g, const QtGeom &q, const DctArgs &a, long long max_items)
{
    if (max_items <= 0) return 0;                  // an empty work list is not an error
    if (a.nplanes > kMaxPlanes) return -1;         // the per-plane prefix table would not fit the launch's LDS
    const size_t pref = (size_t)(a.nplanes + 1) * sizeof(int);
    auto cap = [&](long long per_block, int hi) {
        long long b = (max_items + per_block - 1) / per_block;
        return (int)(b < 1 ? 1 : b > hi ? hi : b);
    };
    const bool wd = a.dct_f32 != nullptr;
#define AEJ_SMALL(S, PER, HI)                                                                                              \
    if (wd) hipLaunchKernelGGL((k_dct_small<S, true>), dim3(cap(PER, HI)), dim3(256), pref, st, g, q, a, max_items);       \
    else hipLaunchKernelGGL((k_dct_small<S, false>), dim3(cap(PER, HI)), dim3(256), pref, st, g, q, a, max_items)
#define AEJ_MFMA(S, HI)                                                                 \
    if (wd) launch_mfma_t<S, true>(st, g, q, a, max_items, cap(1, HI));                  \
    else launch_mfma_t<S, false>(st, g, q, a, max_items, cap(1, HI))
    switch (size) {
    case 2: AEJ_SMALL(2, 128, 2048); break;
    case 4:
        if (wd) hipLaunchKernelGGL((k_dct4<true>), dim3(cap(256, 8192)), dim3(256), pref, st, g, q, a, max_items);
        else hipLaunchKernelGGL((k_dct4<false>), dim3(cap(256, 8192)), dim3(256), pref, st, g, q, a, max_items);
        break;
    case 8: AEJ_SMALL(8, 32, 4096); break;
    case 16: AEJ_SMALL(16, 16, 4096); break;
    case 32: AEJ_MFMA(32, 4096); break;
    case 64: AEJ_MFMA(64, 768); break;
    case 128: AEJ_MFMA(128, 256); break;
    case 256:
        if (!a.scratch) return -1;   // callers reserve it whenever the settings allow this size
        if (wd) hipLaunchKernelGGL((k_dct_big<256, true>), dim3(cap(1, kBigBlocks)), dim3(256), pref, st, g, q, a, max_items);
        else hipLaunchKernelGGL((k_dct_big<256, false>), dim3(cap(1, kBigBlocks)), dim3(256), pref, st, g, q, a, max_items);
        break;
    default: return -1;
    }
#undef AEJ_SMALL
#undef AEJ_MFMA
    return 0;
}

}  // namespace aej
