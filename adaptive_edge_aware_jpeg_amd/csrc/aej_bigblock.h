// aej_bigblock.h -- S x S products for block sizes whose operands do not fit LDS (S = 256, the reference GUI's maximum):
// a plain tiled product, one workgroup of 256 threads per leaf, 64 x 64 output tiles, 16-wide k slices staged through LDS, a
// 4 x 4 micro-tile per thread.  Every output element is the k-ordered fma chain from +0 of the numerics contract, so these
// kernels agree bit for bit with the MFMA / VALU kernels of the smaller sizes and with the CPU oracle.  Leaves of this size only
// occur in flat regions at least 256 pixels wide, so the path is written for clarity, not speed.
#pragma once
#include <hip/hip_runtime.h>

namespace aej {

struct BigTileLds {
    float a[64][17];     // A(i0 + r, k0 + c)
    float b[16][65];     // B(k0 + r, j0 + c)
};

// C(i, j) = sum_k A(i, k) * B(k, j) for i, j in [0, S); loadA(i, k), loadB(k, j) read the operands, storeC(i, j, v) takes results
template <int S, typename FA, typename FB, typename FC>
__device__ __forceinline__ void big_product(BigTileLds &L, FA loadA, FB loadB, FC storeC)
{
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    for (int tile = 0; tile < (S / 64) * (S / 64); tile++) {
        const int i0 = (tile / (S / 64)) * 64, j0 = (tile % (S / 64)) * 64;
        float acc[4][4];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int v = 0; v < 4; v++) acc[u][v] = 0.f;
        for (int k0 = 0; k0 < S; k0 += 16) {
            for (int idx = tid; idx < 64 * 16; idx += 256) { const int r = idx >> 4, c = idx & 15; L.a[r][c] = loadA(i0 + r, k0 + c); }
            for (int idx = tid; idx < 16 * 64; idx += 256) { const int r = idx >> 6, c = idx & 63; L.b[r][c] = loadB(k0 + r, j0 + c); }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 16; kk++) {
                float av[4], bv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) av[u] = L.a[ty * 4 + u][kk];
#pragma unroll
                for (int v = 0; v < 4; v++) bv[v] = L.b[kk][tx * 4 + v];
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int v = 0; v < 4; v++) acc[u][v] = __builtin_fmaf(av[u], bv[v], acc[u][v]);
            }
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int v = 0; v < 4; v++) storeC(i0 + ty * 4 + u, j0 + tx * 4 + v, acc[u][v]);
    }
}

// the first product's result goes through a per-workgroup global scratch: make it visible to the other waves of the workgroup
// (release, barrier, acquire -- the acquire drops lines of the previous leaf's scratch contents from the vector L1)
__device__ __forceinline__ void big_scratch_sync()
{
    __threadfence();
    __syncthreads();
    __threadfence();
}

}  // namespace aej
