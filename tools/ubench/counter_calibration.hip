// counter_calibration.hip -- what FETCH_SIZE / WRITE_SIZE report for the access shapes this library uses (diagnostic, not product).
//
// MI355X_MICROARCH.md calibrates the gfx950 counters for ONE shape only (16 B per lane, streaming: FETCH_SIZE reads half the bytes,
// WRITE_SIZE reads them exactly) and says: "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern before trusting an absolute".  Every kernel below touches a known number of bytes of a buffer far larger than the 256 MiB
// Infinity Cache, exactly once unless its name says otherwise; tools/profiling/counter_calibration.py runs the binary under
// `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) and prints counter / bytes per kernel.
//
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/counter_calibration.hip -o tools/ubench/counter_calibration.out
//
// Shapes (the kernels of csrc/ that use them):
//   rd16 / rd8 / rd4 / rd2 / rd1   N bytes per lane, lanes contiguous                        (colour / DCT / blur prefetch / lists)
//   rd4_tile                        k_sobel_nms_reg's own dwords: a wave owns a 64 x 64 byte tile of a W-byte-wide plane, a load
//                                   instruction = 4 segments of 64 B, 16 rows apart; tiles dealt as the kernel does (4 per workgroup)
//   rd4_tile_halo                   ... plus the halo dword left / right of every segment and the 4 overlap rows of every 16-row band,
//                                   i.e. exactly the kernel's loads (plane bytes fetched once = the algorithmic figure)
//   rd4_tile_halo_xcd               the same with each XCD given a contiguous range of tiles (csrc/canny.hip xcd_remap)
//   rd8_words                       bit-plane words: 8 B per lane, 64 lanes contiguous (hysteresis / quadtree)
//   wr16 / wr4 / wr2q / wr1         stores: 16 B, 4 B per lane contiguous; one 2-byte piece from every fourth lane (Sobel's bit-plane
//                                   pieces); 1 B per lane contiguous
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ unsigned long long g_sink;

template <typename T> __device__ __forceinline__ unsigned fold(T v);
template <> __device__ __forceinline__ unsigned fold<uint4>(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }
template <> __device__ __forceinline__ unsigned fold<uint2>(uint2 v) { return v.x ^ v.y; }
template <> __device__ __forceinline__ unsigned fold<unsigned>(unsigned v) { return v; }
template <> __device__ __forceinline__ unsigned fold<unsigned short>(unsigned short v) { return v; }
template <> __device__ __forceinline__ unsigned fold<unsigned char>(unsigned char v) { return v; }

// contiguous lanes, grid-stride: n elements of T
template <typename T>
__device__ __forceinline__ void rd_body(const T *__restrict__ p, long long n)
{
    unsigned acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) acc ^= fold<T>(p[i]);
    if (acc == (unsigned)n) g_sink = acc;      // (a run-time value: a constant the narrow types cannot reach lets the compiler drop the loads)
}
template <typename T>
__device__ __forceinline__ void wr_body(T *__restrict__ p, long long n, T v)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = v;
}
// (plain names, so that the profiler's kernel names are easy to match)
__global__ __launch_bounds__(256) void k_rd16(const uint4 *p, long long n) { rd_body(p, n); }
__global__ __launch_bounds__(256) void k_rd8(const uint2 *p, long long n) { rd_body(p, n); }
__global__ __launch_bounds__(256) void k_rd4(const unsigned *p, long long n) { rd_body(p, n); }
__global__ __launch_bounds__(256) void k_rd2(const unsigned short *p, long long n) { rd_body(p, n); }
__global__ __launch_bounds__(256) void k_rd1(const unsigned char *p, long long n) { rd_body(p, n); }
__global__ __launch_bounds__(256) void k_wr16(uint4 *p, long long n, uint4 v) { wr_body(p, n, v); }
__global__ __launch_bounds__(256) void k_wr4(unsigned *p, long long n, unsigned v) { wr_body(p, n, v); }
__global__ __launch_bounds__(256) void k_wr1(unsigned char *p, long long n, unsigned char v) { wr_body(p, n, v); }
// one 2-byte piece from every fourth lane, pieces contiguous (k_sobel_nms_reg's stores: 4 pieces = one 64-bit word of a bit-plane row)
__global__ __launch_bounds__(256) void k_wr2q(unsigned short *__restrict__ p, long long n_pieces)
{
    const int lane = threadIdx.x & 63;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) >> 2; i < n_pieces; i += ((long long)gridDim.x * 256) >> 2)
        if ((lane & 3) == 3) p[i] = (unsigned short)i;
}

__device__ __forceinline__ long long xcd_remap(long long id, long long n)
{
    const long long q = n / 8, r = n % 8, xcd = id % 8, k = id / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// MODE 0: own dwords only (every byte once); 1: + halo dwords + the 4 overlap rows per band (the Sobel kernel's loads); XCD: contiguous tiles per XCD
template <int MODE, bool XCD>
__device__ __forceinline__ void rd4_tile_body(const unsigned char *__restrict__ src, int w, int h, int planes)
{
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    const int ntx = w / 64, nty = h / 64;
    const long long wg = XCD ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const long long T = wg * 4 + (threadIdx.x >> 6);
    if (T >= (long long)ntx * nty * planes) return;
    const int pl = (int)(T / (ntx * nty)), t = (int)(T - (long long)pl * ntx * nty), ty = t / ntx, tx = t - ty * ntx;
    const unsigned char *plane = src + (long long)pl * w * h;
    unsigned acc = 0;
    const int rows = MODE == 0 ? 16 : 20, r0 = MODE == 0 ? 0 : -2;
    const int halo_col = j == 0 ? tx * 64 - 4 : j == 15 ? tx * 64 + 64 : tx * 64 + 4 * j;
#pragma unroll 4
    for (int u = 0; u < rows; u++) {
        int y = ty * 64 + 16 * q + r0 + u;
        y = y < 0 ? 0 : y >= h ? h - 1 : y;
        acc ^= *reinterpret_cast<const unsigned *>(plane + (long long)y * w + tx * 64 + 4 * j);
        if (MODE == 1) {
            int c = halo_col < 0 ? 0 : halo_col > w - 4 ? w - 4 : halo_col;
            acc ^= *reinterpret_cast<const unsigned *>(plane + (long long)y * w + c) * 3u;
        }
    }
    if (acc == (unsigned)planes) g_sink = acc;
}

__global__ __launch_bounds__(256) void k_rd4_tile(const unsigned char *s, int w, int h, int p) { rd4_tile_body<0, false>(s, w, h, p); }
__global__ __launch_bounds__(256) void k_rd4_tile_halo(const unsigned char *s, int w, int h, int p) { rd4_tile_body<1, false>(s, w, h, p); }
__global__ __launch_bounds__(256) void k_rd4_tile_halo_xcd(const unsigned char *s, int w, int h, int p) { rd4_tile_body<1, true>(s, w, h, p); }

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 3;
    const long long N = 1LL << 31;                   // 2 GiB buffer (8 x the Infinity Cache)
    unsigned char *buf;
    CK(hipMalloc(&buf, N));
    CK(hipMemset(buf, 1, N));
    const int W = 3840, H = 2160, planes = (int)(N / ((long long)W * H));       // 4K luma planes (the bench's): 258 of them
    const long long plane_bytes = (long long)W * H * planes;
    const long long tiles = (long long)(W / 64) * (H / 64) * planes;           // (2160 = 33.75 x 64: the last 48 rows of a plane are not read)
    const double tile_bytes = (double)tiles * 64 * 64;
    const int grid = 256 * 8;
    printf("buffer %lld B, reps %d, tiles %lld\n", N, reps, tiles);
    for (int r = 0; r < reps; r++) {
        hipLaunchKernelGGL(k_rd16, dim3(grid), dim3(256), 0, 0, (const uint4 *)buf, N / 16);
        hipLaunchKernelGGL(k_rd8, dim3(grid), dim3(256), 0, 0, (const uint2 *)buf, N / 8);
        hipLaunchKernelGGL(k_rd4, dim3(grid), dim3(256), 0, 0, (const unsigned *)buf, N / 4);
        hipLaunchKernelGGL(k_rd2, dim3(grid), dim3(256), 0, 0, (const unsigned short *)buf, N / 2 / 4);      // a quarter of the buffer
        hipLaunchKernelGGL(k_rd1, dim3(grid), dim3(256), 0, 0, (const unsigned char *)buf, N / 4);
        hipLaunchKernelGGL(k_rd4_tile, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, 0, buf, W, H, planes);
        hipLaunchKernelGGL(k_rd4_tile_halo, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, 0, buf, W, H, planes);
        hipLaunchKernelGGL(k_rd4_tile_halo_xcd, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, 0, buf, W, H, planes);
        hipLaunchKernelGGL(k_wr16, dim3(grid), dim3(256), 0, 0, (uint4 *)buf, N / 16, make_uint4(1, 2, 3, 4));
        hipLaunchKernelGGL(k_wr4, dim3(grid), dim3(256), 0, 0, (unsigned *)buf, N / 4, 7u);
        hipLaunchKernelGGL(k_wr2q, dim3(grid), dim3(256), 0, 0, (unsigned short *)buf, N / 2 / 4);
        hipLaunchKernelGGL(k_wr1, dim3(grid), dim3(256), 0, 0, buf, N / 4, (unsigned char)9);
        CK(hipDeviceSynchronize());
    }
    // the byte counts the Python driver divides the counters by (kernel-name substring, bytes per launch, R or W)
    printf("CASE k_rd16 %.0f R\nCASE k_rd8 %.0f R\nCASE k_rd4 %.0f R\nCASE k_rd2 %.0f R\nCASE k_rd1 %.0f R\n", (double)N, (double)N, (double)N, (double)N / 4, (double)N / 4);
    printf("CASE k_rd4_tile %.0f R\nCASE k_rd4_tile_halo %.0f R\nCASE k_rd4_tile_halo_xcd %.0f R\n", tile_bytes, tile_bytes, tile_bytes);
    printf("CASE k_wr16 %.0f W\nCASE k_wr4 %.0f W\nCASE k_wr2q %.0f W\nCASE k_wr1 %.0f W\n", (double)N, (double)N, (double)N / 4, (double)N / 4);
    (void)plane_bytes;
    CK(hipFree(buf));
    return 0;
}
