// corun.hip -- does an HBM-streaming kernel with a SMALL per-CU footprint run beside an issue-bound kernel with a LARGE one
// (the blur kernel's: 512 threads, 128 VGPRs, 77 KiB LDS, two workgroups per CU) at no cost to either?  (diagnostic, not product)
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/corun.hip -o /tmp/corun && /tmp/corun
// A = persistent streaming kernel: `wgs` workgroups of 256 threads per CU, each thread keeps UNROLL x 16-byte loads in flight,
//     reads R bytes and writes R / 2 (the colour-plane kernel's 12 : 7.5 ratio, roughly).
// B = issue-bound kernel: dependent-free fma streams + LDS reads, 77 KiB static LDS, launch bounds (512, 4) -> 128 VGPRs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int UNROLL>
__global__ __launch_bounds__(256) void k_stream(const float4 *__restrict__ in, float4 *__restrict__ out, long long n16)
{
    const long long stride = (long long)gridDim.x * 256 * UNROLL;
    for (long long base = (long long)blockIdx.x * 256 * UNROLL + threadIdx.x; base < n16; base += stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) { long long i = base + (long long)u * 256; v[u] = i < n16 ? in[i] : make_float4(0, 0, 0, 0); }
#pragma unroll
        for (int u = 0; u < UNROLL; u += 2) {
            long long i = base + (long long)u * 256;
            if (i < n16) out[i >> 1] = make_float4(v[u].x + v[u + 1].x, v[u].y * v[u + 1].y, v[u].z - v[u + 1].z, v[u].w + v[u + 1].w);
        }
    }
}

__global__ __launch_bounds__(512, 4) void k_issue(float *out, int iters)
{
    __shared__ float lds[77 * 256];
    for (int i = threadIdx.x; i < 77 * 256; i += 512) lds[i] = (float)i * 1e-4f;
    __syncthreads();
    float a[16];
    for (int i = 0; i < 16; i++) a[i] = (float)(threadIdx.x + i) * 1e-3f;
    const float b = 1.0001f, c = 1e-7f;
    int idx = threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = __builtin_fmaf(a[i], b, c);
            a[r] += lds[idx & (77 * 256 - 1)];
            idx += 517;
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += a[i];
    out[(long long)blockIdx.x * 512 + threadIdx.x] = s;
}

static float elapsed(hipEvent_t a, hipEvent_t b) { float ms; (void)hipEventElapsedTime(&ms, a, b); return ms; }

int main()
{
    const long long bytes = 6LL << 30;               // 6 GiB read, 3 GiB written per launch
    const long long n16 = bytes / 16;
    float4 *in, *outA; float *outB;
    (void)hipMalloc(&in, bytes); (void)hipMalloc(&outA, bytes / 2); (void)hipMalloc(&outB, 4096LL * 512 * 4);
    (void)hipMemset(in, 0, bytes);
    hipStream_t sa, sb; (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    hipEvent_t e0, e1, e2, e3; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&e2); (void)hipEventCreate(&e3);
    const int b_blocks = 2048, b_iters = 1400;      // sized to ~2 ms alone
    auto runB = [&](hipStream_t s) { hipLaunchKernelGGL(k_issue, dim3(b_blocks), dim3(512), 0, s, outB, b_iters); };
    for (int wgs : { 1, 2, 4, 8, 32 }) {
        auto runA = [&](hipStream_t s) { hipLaunchKernelGGL(k_stream<8>, dim3(256 * wgs), dim3(256), 0, s, in, outA, n16); };
        runA(sa); runB(sb); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, sa); runA(sa); (void)hipEventRecord(e1, sa); (void)hipDeviceSynchronize();
        const float tA = elapsed(e0, e1);
        (void)hipEventRecord(e0, sb); runB(sb); (void)hipEventRecord(e1, sb); (void)hipDeviceSynchronize();
        const float tB = elapsed(e0, e1);
        // together: B first (it fills the CUs), A right behind it on the other stream
        (void)hipEventRecord(e0, sb); runB(sb); (void)hipEventRecord(e1, sb);
        (void)hipEventRecord(e2, sa); runA(sa); (void)hipEventRecord(e3, sa);
        (void)hipDeviceSynchronize();
        const float tB2 = elapsed(e0, e1), tA2 = elapsed(e2, e3), span = elapsed(e0, e3) > elapsed(e0, e1) ? elapsed(e0, e3) : elapsed(e0, e1);
        // together, A first
        (void)hipEventRecord(e2, sa); runA(sa); (void)hipEventRecord(e3, sa);
        (void)hipEventRecord(e0, sb); runB(sb); (void)hipEventRecord(e1, sb);
        (void)hipDeviceSynchronize();
        const float tB3 = elapsed(e0, e1), tA3 = elapsed(e2, e3);
        printf("stream kernel %2d WG/CU: alone %.3f ms = %.2f TB/s (R+W) | issue kernel alone %.3f ms | together (B first): A %.3f B %.3f span %.3f (serial %.3f) | "
               "(A first): A %.3f B %.3f\\n", wgs, tA, (bytes * 1.5) / tA / 1e9, tB, tA2, tB2, span, tA + tB, tA3, tB3);
    }
    return 0;
}
