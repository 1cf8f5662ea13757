#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
    unsigned v = threadIdx.x + 100;
    unsigned up = (unsigned)__builtin_amdgcn_update_dpp((int)7777, (int)v, 0x138, 0xf, 0xf, false);   // wave_shr:1 -> lane i gets lane i-1
    unsigned dn = (unsigned)__builtin_amdgcn_update_dpp((int)8888, (int)v, 0x130, 0xf, 0xf, false);   // wave_shl:1 -> lane i gets lane i+1
    o[threadIdx.x] = up; o[64 + threadIdx.x] = dn;
}
int main() { unsigned *o; (void)hipMallocManaged(&o, 512); k<<<1,64>>>(o); (void)hipDeviceSynchronize();
  printf("up: %u %u %u ... %u %u | dn: %u %u ... %u %u %u\n", o[0], o[1], o[2], o[31], o[32], o[64], o[65], o[64+31], o[64+62], o[64+63]); }
