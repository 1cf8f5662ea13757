// pk_f32.hip -- issue rate of the packed float32 VALU instructions next to their scalar forms (diagnostic, not product):
// is one v_pk_fma_f32 (two lanes' worth of fma per lane) as cheap as one v_fma_f32?  Decides whether the bilateral filter's
// (sum, wsum) updates are worth packing.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/pk_f32.hip -o /tmp/pk_f32 && /tmp/pk_f32
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define BODY(INS)                                                                                               \
    REP16(asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                   \
                       : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                       : "v"(b), "v"(c));)
#define I0(n) "v_pk_fma_f32 %" #n ", %8, %9, %" #n "\n"
#define I1(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
#define I2(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
#define I3(n) "v_pk_mov_b32 %" #n ", %8, %9 op_sel:[1,0]\n"
#define I4(n) "v_pk_fma_f32 %" #n ", %8, %9, %" #n " op_sel:[0,0,0] op_sel_hi:[1,0,1]\n"
#define I5(n) "v_fma_f32 %" #n ", %8, %9, %" #n "\n"

template <int OP>
__global__ __launch_bounds__(512) void k(float *out, int iters)
{
    float2v a[8];
    for (int i = 0; i < 8; i++) a[i] = float2v{ (float)(threadIdx.x + i) * 1e-3f, 1.0f };
    float2v b = { 1.0001f, 0.9999f }, c = { 1e-7f, 2e-7f };
    for (int it = 0; it < iters; it++) {
#define CASE(N) if (OP == N) { BODY(I##N) }
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4)
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(512) void k5(float *out, int iters)
{
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = (float)(threadIdx.x + i) * 1e-3f;
    float b = 1.0001f, c = 1e-7f;
    for (int it = 0; it < iters; it++) { BODY(I5) }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static void run(const char *name, K kern)
{
    for (int wpsimd : { 2, 4, 8 }) {
        const int threads = 64 * 4 * (wpsimd > 2 ? 2 : wpsimd), wg_per_cu = wpsimd > 2 ? wpsimd / 2 : 1, blocks = 256 * wg_per_cu, iters = 2000;
        float *out;
        (void)hipMalloc(&out, (size_t)blocks * threads * 4);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        kern<<<blocks, threads>>>(out, 8);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        kern<<<blocks, threads>>>(out, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)iters * 128 * wpsimd;       // 16 x 8 per iteration per wave
        printf("%-44s waves/SIMD %d: wall %7.3f ms -> %6.2f ns per SIMD-instr\n", name, wpsimd, ms, ms * 1e6 / instr_per_simd);
        (void)hipFree(out);
    }
}

int main()
{
    run("v_fma_f32", k5);
    run("v_pk_fma_f32", k<0>);
    run("v_pk_add_f32", k<1>);
    run("v_pk_mul_f32", k<2>);
    run("v_pk_mov_b32", k<3>);
    run("v_pk_fma_f32 op_sel (hi lane takes src1.lo)", k<4>);
    return 0;
}
