// Do kernels launched with hipExtAnyOrderLaunch overlap on ONE stream on gfx950?  (hip_ext.h says the flag is not supported on GFX9xx.)
// Two single-workgroup kernels that each spin ~200 us, back to back on one stream: in order they take ~400 us, overlapped ~200 us.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/anyorder.hip -o /tmp/anyorder && /tmp/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

__global__ void spin(long long cycles, int *out)
{
    const long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(10);
    if (threadIdx.x == 0) *out = 1;
}

int main()
{
    int *d;
    hipMalloc(&d, 64);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const long long cycles = 20000;      // s_memtime ticks at 100 MHz: 200 us
    for (int flags = 0; flags < 2; flags++) {
        for (int rep = 0; rep < 3; rep++) {
            hipStreamSynchronize(st);
            const auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < 4; k++) {
                int *o = d + k;
                long long c = cycles;
                void *args[] = { &c, &o };
                hipError_t e = hipExtLaunchKernel(reinterpret_cast<const void *>(&spin), dim3(1), dim3(64), args, 0, st, nullptr, nullptr, k == 0 ? 0 : flags);
                if (e != hipSuccess) printf("launch: %s\n", hipGetErrorString(e));
            }
            hipStreamSynchronize(st);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("flags %d: four 200-us kernels on one stream took %.0f us\n", flags, us);
        }
    }
    return 0;
}
