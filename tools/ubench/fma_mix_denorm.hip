#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float mix_lo(unsigned p, float w, float acc) {
    float r; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(w), "v"(acc)); return r; }
__device__ __forceinline__ float mix_hi(unsigned p, float w, float acc) {
    float r; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(w), "v"(acc)); return r; }
__device__ __forceinline__ int sub_hi_lo(unsigned a, unsigned b) {   // a.hi - b.lo
    int r; asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "=v"(r) : "v"(a), "v"(b)); return r; }
__global__ void k(const unsigned *p, const float *w, const float *acc, float *o, int *d, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    o[2*i] = mix_lo(p[i], w[i], acc[i]); o[2*i+1] = mix_hi(p[i], w[i], acc[i]); d[i] = sub_hi_lo(p[i], p[i]);
}
// compiler path
__global__ void k2(const unsigned *p, const float *w, const float *acc, float *o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    unsigned v = p[i];
    _Float16 lo = __builtin_bit_cast(_Float16, (unsigned short)(v & 0xffff)), hi = __builtin_bit_cast(_Float16, (unsigned short)(v >> 16));
    o[2*i] = __builtin_fmaf((float)lo, w[i], acc[i]); o[2*i+1] = __builtin_fmaf((float)hi, w[i], acc[i]);
}
int main() {
    const int n = 1 << 16;
    unsigned *p; float *w, *acc, *o; int *d;
    (void)hipMallocManaged(&p, n*4); (void)hipMallocManaged(&w, n*4); (void)hipMallocManaged(&acc, n*4); (void)hipMallocManaged(&o, n*8); (void)hipMallocManaged(&d, n*4);
    unsigned s = 12345;
    for (int i = 0; i < n; i++) { s = s*1664525u+1013904223u; unsigned lo = (s>>8)%1021, hi = (s>>20)%1021; p[i] = lo | hi<<16;
        s = s*1664525u+1013904223u; w[i] = (float)(s>>8) / 16777216.0f * (i%7==0 ? 1e-30f : 1.f); s = s*1664525u+1013904223u; acc[i] = (i%5==0) ? 0.f : (float)(s>>8) / 16777216.0f * 1e-4f; }
    k<<<n/256,256>>>(p,w,acc,o,d,n); (void)hipDeviceSynchronize();
    int bad = 0, badd = 0;
    for (int i = 0; i < n; i++) { unsigned lo = p[i]&0xffff, hi = p[i]>>16;
        float e0 = __builtin_fmaf((float)lo * 5.9604644775390625e-8f, w[i], acc[i]), e1 = __builtin_fmaf((float)hi * 5.9604644775390625e-8f, w[i], acc[i]);
        if (e0 != o[2*i] || e1 != o[2*i+1]) { if (bad < 5) printf("mismatch %d: %u %u w %g acc %g got %g %g want %g %g\n", i, lo, hi, w[i], acc[i], o[2*i], o[2*i+1], e0, e1); bad++; }
        if (d[i] != (int)hi - (int)lo) badd++; }
    printf("asm: bad %d badd %d of %d\n", bad, badd, n);
    k2<<<n/256,256>>>(p,w,acc,o,n); (void)hipDeviceSynchronize(); bad = 0;
    for (int i = 0; i < n; i++) { unsigned lo = p[i]&0xffff, hi = p[i]>>16;
        float e0 = __builtin_fmaf((float)lo * 5.9604644775390625e-8f, w[i], acc[i]), e1 = __builtin_fmaf((float)hi * 5.9604644775390625e-8f, w[i], acc[i]);
        if (e0 != o[2*i] || e1 != o[2*i+1]) bad++; }
    printf("compiler: bad %d\n", bad);
}
