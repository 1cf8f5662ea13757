// valu_issue.hip -- measures the per-SIMD issue cost (shader cycles per wave64 instruction) of the VALU / LDS instruction
// kinds the stencil kernels of this path are made of, at 1, 2, 4 and 8 waves per SIMD.  Diagnostic only (not product):
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_issue.hip -o /tmp/valu_issue && /tmp/valu_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

// 8 independent chains, 16 x 8 = 128 instructions per loop iteration
#define BODY(INS)                                                                                               \
    REP16(asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                   \
                       : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                       : "v"(b), "v"(c));)

#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_ADDF(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define I_MULF(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define I_ADDU(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define I_CVTUB(n) "v_cvt_f32_ubyte1 %" #n ", %" #n "\n"
#define I_CVTI(n) "v_cvt_i32_f32 %" #n ", %" #n "\n"
#define I_SDWA(n) "v_sub_u32_sdwa %" #n ", %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define I_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define I_LSHL(n) "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
#define I_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define I_MED3(n) "v_med3_i32 %" #n ", %" #n ", %8, %9\n"
#define I_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define I_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 8, 8\n"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 1, %8\n"
#define I_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define I_PKADD16(n) "v_pk_add_u16 %" #n ", %" #n ", %8\n"

template <int OP>
__global__ __launch_bounds__(512) void k_scalar(float *out, int iters, long long *cyc)
{
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = (float)(threadIdx.x + i) * 1e-3f;
    float b = 1.0001f, c = 1e-7f;
    if (OP >= 3 && OP != 8) { b = __int_as_float(0x01020304 + threadIdx.x); c = __int_as_float(0x07060504); }
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (OP == 0) { BODY(I_FMA) }
        if (OP == 1) { BODY(I_ADDF) }
        if (OP == 2) { BODY(I_MULF) }
        if (OP == 3) { BODY(I_ADDU) }
        if (OP == 4) { BODY(I_PERM) }
        if (OP == 5) { BODY(I_CVTUB) }
        if (OP == 6) { BODY(I_CVTI) }
        if (OP == 7) { BODY(I_SDWA) }
        if (OP == 8) { BODY(I_RCP) }
        if (OP == 9) { BODY(I_LSHL) }
        if (OP == 10) { BODY(I_AND) }
        if (OP == 11) { BODY(I_MED3) }
        if (OP == 12) { BODY(I_MULLO) }
        if (OP == 13) { BODY(I_BFE) }
        if (OP == 14) { BODY(I_LSHLADD) }
        if (OP == 15) { BODY(I_ADD3) }
        if (OP == 16) { BODY(I_PKADD16) }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// packed f32: operands are 64-bit register pairs
typedef float float2v __attribute__((ext_vector_type(2)));
#define BODY2(INS)                                                                                              \
    REP16(asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                   \
                       : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                       : "v"(b), "v"(c));)
#define I_PKFMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_PKADD(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
#define I_PKMUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
#define I_PKFMA_SEL(n) "v_pk_fma_f32 %" #n ", %8, %9, %" #n " op_sel_hi:[1,0,1]\n"

template <int OP>
__global__ __launch_bounds__(512) void k_packed(float *out, int iters, long long *cyc)
{
    float2v a[8];
    for (int i = 0; i < 8; i++) a[i] = float2v{ (float)(threadIdx.x + i) * 1e-3f, (float)i };
    float2v b = { 1.0001f, 0.9999f }, c = { 1e-7f, 2e-7f };
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (OP == 0) { BODY2(I_PKFMA) }
        if (OP == 1) { BODY2(I_PKADD) }
        if (OP == 2) { BODY2(I_PKMUL) }
        if (OP == 3) { BODY2(I_PKFMA_SEL) }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// LDS: ds_read_b32 / ds_read_u8 with per-lane addresses (conflict-free stride, random, same word)
template <int OP>
__global__ __launch_bounds__(512) void k_lds(float *out, int iters, long long *cyc, int mode)
{
    __shared__ unsigned int s[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = i * 2654435761u;
    __syncthreads();
    unsigned int addr[8];
    for (int i = 0; i < 8; i++) {
        unsigned int lane = threadIdx.x & 63;
        unsigned int idx = mode == 0 ? lane + 64 * i : mode == 1 ? ((lane * 2654435761u + i * 40503u) >> 7) & 2047 : 7 + i;
        addr[i] = (unsigned int)(size_t)(&s[0]) + idx * 4;     // LDS byte address (low 32 bits of the generic pointer offset)
    }
    unsigned int acc = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            unsigned int v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("ds_read_b32 %0, %1" : "=v"(v[i]) : "v"(addr[i]));
                if (OP == 1) asm volatile("ds_read_u8 %0, %1" : "=v"(v[i]) : "v"(addr[i]));
                if (OP == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(*(unsigned long long *)&v[i & 6]) : "v"(addr[i] & ~7u));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("" :: "v"(v[i]));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <typename K>
static void run(const char *name, K kern, int wpsimd, int iters, int ninstr_per_iter)
{
    const int threads = 64 * 4 * (wpsimd > 2 ? 2 : wpsimd);         // waves per workgroup = 4 * min(wpsimd, 2)
    const int wg_per_cu = wpsimd > 2 ? wpsimd / 2 : 1;
    const int blocks = 256 * wg_per_cu;
    float *out; long long *cyc;
    hipMalloc(&out, (size_t)blocks * threads * 4);
    hipMalloc(&cyc, (size_t)blocks * (threads / 64) * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern(blocks, threads, out, 8, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern(blocks, threads, out, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h((size_t)blocks * (threads / 64));
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= (double)h.size();
    const double per_wave = mean / ((double)iters * ninstr_per_iter);
    // wall-clock view: instructions per SIMD per second
    const double simd_instr = (double)iters * ninstr_per_iter * wpsimd;       // wave-instructions per SIMD
    printf("%-28s waves/SIMD %d: %6.2f memtime-ticks per instr per wave, %6.2f per SIMD-instr;  wall %7.3f ms -> %6.2f ns per SIMD-instr\n", name, wpsimd,
           per_wave, per_wave / wpsimd, ms, ms * 1e6 / simd_instr);
    hipFree(out); hipFree(cyc);
}

#define RUN_S(NAME, OP) for (int w : { 1, 2, 4, 8 }) run(NAME, [](int bl, int th, float *o, int it, long long *c) { hipLaunchKernelGGL((k_scalar<OP>), dim3(bl), dim3(th), 0, 0, o, it, c); }, w, 2000, 128)
#define RUN_P(NAME, OP) for (int w : { 1, 2, 4, 8 }) run(NAME, [](int bl, int th, float *o, int it, long long *c) { hipLaunchKernelGGL((k_packed<OP>), dim3(bl), dim3(th), 0, 0, o, it, c); }, w, 2000, 128)
#define RUN_L(NAME, OP, MODE) for (int w : { 1, 2, 4, 8 }) run(NAME, [](int bl, int th, float *o, int it, long long *c) { hipLaunchKernelGGL((k_lds<OP>), dim3(bl), dim3(th), 0, 0, o, it, c, MODE); }, w, 500, 128)

int main()
{
    RUN_S("v_fma_f32", 0);
    RUN_S("v_add_f32", 1);
    RUN_S("v_mul_f32", 2);
    RUN_S("v_add_u32", 3);
    RUN_S("v_perm_b32", 4);
    RUN_S("v_cvt_f32_ubyte1", 5);
    RUN_S("v_cvt_i32_f32", 6);
    RUN_S("v_sub_u32_sdwa", 7);
    RUN_S("v_rcp_f32", 8);
    RUN_S("v_lshlrev_b32", 9);
    RUN_S("v_and_b32", 10);
    RUN_S("v_med3_i32", 11);
    RUN_S("v_mul_lo_u32", 12);
    RUN_S("v_bfe_u32", 13);
    RUN_S("v_lshl_add_u32", 14);
    RUN_S("v_add3_u32", 15);
    RUN_S("v_pk_add_u16", 16);
    RUN_P("v_pk_fma_f32", 0);
    RUN_P("v_pk_add_f32", 1);
    RUN_P("v_pk_mul_f32", 2);
    RUN_P("v_pk_fma_f32 op_sel_hi", 3);
    RUN_L("ds_read_b32 stride-1", 0, 0);
    RUN_L("ds_read_b32 random", 0, 1);
    RUN_L("ds_read_b32 broadcast", 0, 2);
    RUN_L("ds_read_u8 random", 1, 1);
    RUN_L("ds_read_b64 stride", 2, 0);
    return 0;
}
