// aej_mfma.h -- the float32 MFMA accumulation chain shared by the DCT (dct.hip) and IDCT (decode.hip) kernels of the 32 / 64 / 128 blocks
#pragma once
#include <hip/hip_runtime.h>

namespace aej {

typedef float floatx16 __attribute__((ext_vector_type(16)));

#ifndef AEJ_MFMA_PF
#define AEJ_MFMA_PF 4
#endif
constexpr int kMfmaPF = AEJ_MFMA_PF;      // A operands requested this many MFMAs ahead

// One chain of S / 2 dependent MFMAs per tile: acc += A(:, 2s .. 2s+1) * B(2s .. 2s+1, :).  The A operands come from LDS; the
// compiler's own schedule is "read, wait, two MFMAs, read, wait, ..." on one register pair, which exposes the LDS latency
// once per pair.  Here the operands of the next PF steps are requested before the current PF MFMAs are issued (a scheduling
// barrier keeps that order), so each wait finds data that was requested PF x 64 cycles earlier.
template <int S, int TPW, int PF>
__device__ __forceinline__ void mfma_chain(const float *sA, int tile_stride, int col0, int lh, const float (&dreg)[S / 2], floatx16 (&acc)[TPW])
{
    float cur[TPW][PF], nxt[TPW][PF];
#pragma unroll
    for (int t = 0; t < TPW; t++)
#pragma unroll
        for (int i = 0; i < PF; i++) cur[t][i] = sA[(2 * i + lh) * S + col0 + t * tile_stride];
#pragma unroll
    for (int s0 = 0; s0 < S / 2; s0 += PF) {
        if (s0 + PF < S / 2) {
#pragma unroll
            for (int t = 0; t < TPW; t++)
#pragma unroll
                for (int i = 0; i < PF; i++) nxt[t][i] = sA[(2 * (s0 + PF + i) + lh) * S + col0 + t * tile_stride];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < PF; i++)
#pragma unroll
            for (int t = 0; t < TPW; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[t][i], dreg[s0 + i], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TPW; t++)
#pragma unroll
            for (int i = 0; i < PF; i++) cur[t][i] = nxt[t][i];
    }
}

}  // namespace aej
