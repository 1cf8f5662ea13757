// Hardware check (GPU box): hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma16_order.hip -o /tmp/mfma16 && /tmp/mfma16
//  1. does v_mfma_f32_16x16x4_f32 accumulate its four k values as the k-ordered fma chain acc = fma(a_k, b_k, acc), k ascending
//     (lane group 0, 1, 2, 3), i.e. the DCT contract?  Compared bit for bit against that chain on the host.
//  2. v_permlane32_swap + v_permlane16_swap as the 4 x 4 transpose between the four 16-lane groups and four registers that turns
//     the C layout of one product (row 4 g + r) into the A layout of the next (k = 4 s + g).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef float floatx4 __attribute__((ext_vector_type(4)));

__global__ void k_prod(float *C, const float *A, const float *B)     // C[16][16] = A[16][16] . B[16][16]
{
    const int l = threadIdx.x, g = l >> 4, i = l & 15;
    floatx4 acc = { 0.f, 0.f, 0.f, 0.f };
    for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * 16 + 4 * s + g], B[(4 * s + g) * 16 + i], acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) C[(4 * g + r) * 16 + i] = acc[r];
}

__global__ void k_swap(unsigned *out)
{
    const int l = threadIdx.x, g = l >> 4;
    unsigned R[4];
    for (int r = 0; r < 4; r++) R[r] = (unsigned)(g * 16 + r) * 256u + (unsigned)(l & 15);     // (group, reg) tag
    auto p02 = __builtin_amdgcn_permlane32_swap(R[0], R[2], false, false);
    auto p13 = __builtin_amdgcn_permlane32_swap(R[1], R[3], false, false);
    auto q01 = __builtin_amdgcn_permlane16_swap(p02[0], p13[0], false, false);
    auto q23 = __builtin_amdgcn_permlane16_swap(p02[1], p13[1], false, false);
    out[0 * 64 + l] = q01[0]; out[1 * 64 + l] = q01[1]; out[2 * 64 + l] = q23[0]; out[3 * 64 + l] = q23[1];
}

int main()
{
    float hA[256], hB[256], hC[256], *dA, *dB, *dC;
    unsigned *dS, hS[256];
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 1024); hipMalloc(&dS, 1024);
    int bad_chain = 0, bad_pair = 0, bad_rev = 0;
    srand(1);
    for (int trial = 0; trial < 2000; trial++) {
        for (int i = 0; i < 256; i++) {
            hA[i] = ((float)rand() / RAND_MAX - 0.5f) * (trial & 1 ? 254.f : 1.f);
            hB[i] = ((float)rand() / RAND_MAX - 0.5f) * (trial & 2 ? 1e3f : 1.f);
        }
        hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
        k_prod<<<1, 64>>>(dC, dA, dB);
        hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++) {
                float chain = 0.f, rev = 0.f, pair = 0.f;
                for (int k = 0; k < 16; k++) chain = fmaf(hA[i * 16 + k], hB[k * 16 + j], chain);
                for (int s = 0; s < 4; s++) { for (int k = 3; k >= 0; k--) rev = fmaf(hA[i * 16 + 4 * s + k], hB[(4 * s + k) * 16 + j], rev); }
                for (int s = 0; s < 4; s++) {            // pairwise tree inside one instruction
                    float p0 = fmaf(hA[i * 16 + 4 * s + 1], hB[(4 * s + 1) * 16 + j], hA[i * 16 + 4 * s] * hB[(4 * s) * 16 + j]);
                    float p1 = fmaf(hA[i * 16 + 4 * s + 3], hB[(4 * s + 3) * 16 + j], hA[i * 16 + 4 * s + 2] * hB[(4 * s + 2) * 16 + j]);
                    pair = pair + (p0 + p1);
                }
                float got = hC[i * 16 + j];
                if (memcmp(&got, &chain, 4)) bad_chain++;
                if (memcmp(&got, &rev, 4)) bad_rev++;
                if (memcmp(&got, &pair, 4)) bad_pair++;
            }
    }
    printf("mfma_f32_16x16x4_f32 vs k-ascending fma chain: %d mismatches of %d; vs descending-within-instruction: %d; vs pairwise: %d\n", bad_chain, 2000 * 256, bad_rev, bad_pair);
    k_swap<<<1, 64>>>(dS);
    hipMemcpy(hS, dS, 1024, hipMemcpyDeviceToHost);
    int bad_t = 0;
    for (int s = 0; s < 4; s++)
        for (int l = 0; l < 64; l++) {
            const int g = l >> 4;
            const unsigned want = (unsigned)(s * 16 + g) * 256u + (unsigned)(l & 15);     // new (group g, reg s) = old (group s, reg g)
            if (hS[s * 64 + l] != want) { if (bad_t < 8) printf("swap: reg %d lane %d got %x want %x\n", s, l, hS[s * 64 + l], want); bad_t++; }
        }
    printf("permlane32_swap + permlane16_swap 4x4 group/register transpose: %d mismatches\n", bad_t);
    return (bad_chain || bad_t) ? 1 : 0;
}
