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

namespace aej {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// np.pad(mode='reflect') index map for a block clipped to n valid samples (jpeg.py:399-402)
__device__ __forceinline__ int reflect_pad_idx(int i, int n)
{
    if (i < n) return i;
    if (n <= 1) return 0;
    int p = 2 * (n - 1);
    int j = i % p;
    return j >= n ? p - j : j;
}

__device__ __forceinline__ int quantise(float y, int q)
{
    double v = (double)y / (double)q;
    return (int)rint(v);
}

// Work items of one block size are the concatenation, over planes (b, l), of that plane's Morton-ordered leaf list.
// s_pref[p] = number of items in planes < p (built once per workgroup by wave 0); an item index is mapped back to
// (plane, index in plane) with a binary search.
__device__ __forceinline__ int wave_incl_scan_i(int v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

__device__ __forceinline__ void build_plane_prefix(const DctArgs &a, int *s_pref)
{
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

__device__ __forceinline__ int4 fetch_item(const DctArgs &a, const QtGeom &q, const int *s_pref, long long item)
{
    int lo = 0, hi = a.nplanes;          // largest p with s_pref[p] <= item
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if ((long long)s_pref[mid] <= item) lo = mid; else hi = mid;
    }
    const int b = lo / 3, l = lo - 3 * b;
    const long long idx = (long long)b * q.work_stride[a.k] + q.work_off[l][a.k] + (item - s_pref[lo]);
    return reinterpret_cast<const int4 *>(a.work)[idx];
}

// ------------------------------------------------------------------------------------------------
// small blocks: S in {2, 4, 8, 16}; 256 threads = 256/S leaves, S threads per leaf
// ------------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void k_dct_small(Geom g, QtGeom q, DctArgs a, long long max_items)
{
    constexpr int LPB = 256 / S;
    constexpr int SS = S * S;
    __shared__ float sT[LPB * S * (S + 1)];
    __shared__ int sQ[LPB * SS];
    __shared__ float sD[SS];
    __shared__ int sZ[SS];
    __shared__ int sQm[3 * SS];
    extern __shared__ int s_pref[];
    const int tid = threadIdx.x;
    for (int i = tid; i < SS; i += 256) { sD[i] = a.D[i]; sZ[i] = a.zzinv[i]; }
    for (int i = tid; i < 3 * SS; i += 256) sQm[i] = a.qm[i / SS] ? a.qm[i / SS][i % SS] : 1;
    build_plane_prefix(a, s_pref);
    long long count = s_pref[a.nplanes];
    if (count > max_items) count = max_items;
    const int slot = tid / S, j = tid % S;
    for (long long base = (long long)blockIdx.x * LPB; base < count; base += (long long)gridDim.x * LPB) {
        const long long item = base + slot;
        const bool active = item < count;
        int layer = 0;
        int4 wk = make_int4(0, 0, 0, 0);
        if (active) {
            wk = fetch_item(a, q, s_pref, item);
            const int b = wk.x / 3;
            layer = wk.x - b * 3;
            const int w = g.w[layer], h = g.h[layer];
            const float *src = a.norm + (long long)b * g.pstride + g.poff[layer];
            const int hc = min(S, h - wk.z), wc = min(S, w - wk.y);
            const int col = wk.y + reflect_pad_idx(j, wc);
            float x[S];
#pragma unroll
            for (int k = 0; k < S; k++) x[k] = src[(long long)(wk.z + reflect_pad_idx(k, hc)) * w + col];
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
            const int b = wk.x / 3;
            out_base = (long long)b * q.coeff_stride + q.coeff_off[layer] + wk.w;
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
                if (a.dct_f32) a.dct_f32[out_base + ridx] = acc;
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
// large blocks: S in {32, 64, 128}; one workgroup of (S/32)^2 waves per leaf, one 32x32 output tile per wave.
// P = X^T.D^T  (P[r][c] = T[c][r]) with A[i][k] = X[k][I0+i] read from LDS rows, B[k][j] = D[J0+j][k] held
// in S/2 registers for the whole kernel; P goes to LDS as stored, and Y = T.D^T reads A[i][k] = P[k][I0+i]
// with the same row pattern and the same B registers.  Both LDS read patterns are 32 consecutive floats per
// half-wave: conflict-free.
// ------------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__((S / 32) * (S / 32) * 64) void k_dct_mfma(Geom g, QtGeom q, DctArgs a, long long max_items)
{
    constexpr int NT = S / 32;
    constexpr int NTHREADS = NT * NT * 64;
    constexpr int SS = S * S;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sX = smem;            // [S][S]  X, later reused as int staging for the zigzag scatter
    float *sP = smem + SS;       // [S][S]  P = T^T
    int *s_pref = reinterpret_cast<int *>(smem + 2 * SS);   // [nplanes + 1]
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wi = wave / NT, wj = wave % NT;
    const int I0 = wi * 32, J0 = wj * 32;
    const int li = lane & 31, lh = lane >> 5;

    // B operand registers: D[J0 + li][2*s + lh]
    float dreg[S / 2];
#pragma unroll
    for (int s = 0; s < S / 2; s++) dreg[s] = a.D[(J0 + li) * S + 2 * s + lh];

    build_plane_prefix(a, s_pref);
    long long count = s_pref[a.nplanes];
    if (count > max_items) count = max_items;
    for (long long item = blockIdx.x; item < count; item += gridDim.x) {
        const int4 wk = fetch_item(a, q, s_pref, item);
        const int b = wk.x / 3, layer = wk.x - b * 3;
        const int w = g.w[layer], h = g.h[layer];
        const float *src = a.norm + (long long)b * g.pstride + g.poff[layer];
        const int hc = min(S, h - wk.z), wc = min(S, w - wk.y);
        const long long out_base = (long long)b * q.coeff_stride + q.coeff_off[layer] + wk.w;
        const int *qm = a.qm[layer];

        for (int idx = tid; idx < SS; idx += NTHREADS) {
            int r = idx / S, c = idx - r * S;
            sX[idx] = src[(long long)(wk.z + reflect_pad_idx(r, hc)) * w + wk.y + reflect_pad_idx(c, wc)];
        }
        __syncthreads();

        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < S / 2; s++) {
            float av = sX[(2 * s + lh) * S + I0 + li];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, dreg[s], acc, 0, 0, 0);
        }
        // accumulator layout: row = (r & 3) + 8 * (r >> 2) + 4 * lh, col = li
#pragma unroll
        for (int r = 0; r < 16; r++) {
            int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            sP[(I0 + row) * S + J0 + li] = acc[r];
        }
        __syncthreads();

#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < S / 2; s++) {
            float av = sP[(2 * s + lh) * S + I0 + li];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, dreg[s], acc, 0, 0, 0);
        }
        int *sQ = reinterpret_cast<int *>(sX);   // every wave finished reading sX before the barrier above
#pragma unroll
        for (int r = 0; r < 16; r++) {
            int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            int ridx = (I0 + row) * S + J0 + li;
            if (a.dct_f32) a.dct_f32[out_base + ridx] = acc[r];
            sQ[a.zzinv[ridx]] = quantise(acc[r], qm[ridx]);
        }
        __syncthreads();
        for (int idx = tid * 4; idx < SS; idx += NTHREADS * 4)
            *reinterpret_cast<int4 *>(a.coeffs + out_base + idx) = *reinterpret_cast<const int4 *>(sQ + idx);
        __syncthreads();
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

template <int S>
static void launch_mfma_t(hipStream_t st, const Geom &g, const QtGeom &q, const DctArgs &a, long long max_items, int blocks)
{
    constexpr int NT = S / 32;
    size_t lds = (size_t)2 * S * S * sizeof(float) + (size_t)(a.nplanes + 1) * sizeof(int);
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dct_mfma<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    hipLaunchKernelGGL(k_dct_mfma<S>, dim3(blocks), dim3(NT * NT * 64), lds, st, g, q, a, max_items);
}

void launch_dct(hipStream_t st, int size, const Geom &g, const QtGeom &q, const DctArgs &a, long long max_items)
{
    if (max_items <= 0 || a.nplanes > kMaxPlanes) return;
    const size_t pref = (size_t)(a.nplanes + 1) * sizeof(int);
    auto cap = [&](long long per_block, int hi) {
        long long b = (max_items + per_block - 1) / per_block;
        return (int)(b < 1 ? 1 : b > hi ? hi : b);
    };
    switch (size) {
    case 2: hipLaunchKernelGGL(k_dct_small<2>, dim3(cap(128, 2048)), dim3(256), pref, st, g, q, a, max_items); break;
    case 4: hipLaunchKernelGGL(k_dct_small<4>, dim3(cap(64, 4096)), dim3(256), pref, st, g, q, a, max_items); break;
    case 8: hipLaunchKernelGGL(k_dct_small<8>, dim3(cap(32, 4096)), dim3(256), pref, st, g, q, a, max_items); break;
    case 16: hipLaunchKernelGGL(k_dct_small<16>, dim3(cap(16, 4096)), dim3(256), pref, st, g, q, a, max_items); break;
    case 32: launch_mfma_t<32>(st, g, q, a, max_items, cap(1, 4096)); break;
    case 64: launch_mfma_t<64>(st, g, q, a, max_items, cap(1, 1024)); break;
    case 128: launch_mfma_t<128>(st, g, q, a, max_items, cap(1, 256)); break;
    default: break;
    }
}

}  // namespace aej
