// aej_common.h -- structures shared by the HIP translation units of libaejpeg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aej {

constexpr int kMaxSizes = 8;      // block sizes 2..256
constexpr int kClaheTiles = 4;    // tileGridSize (4,4), edge_detection.py:36
constexpr int kHystTile = 64;     // hysteresis tile edge (pixels)
constexpr int kBlurTW = 64, kBlurTH = 32;  // output tile of the fused CLAHE/Gauss/bilateral kernel
constexpr int kQtChunk = 256;     // min-cells per quadtree chunk (16x16, one Morton-aligned square)

// Geometry of one batch: every image has the same H x W; up to 3 layers per image.
struct Geom {
    int B;            // images
    int nl;           // layers in use (3, or 1 for the stand-alone stage entry points)
    int H, W;         // full-resolution image
    int h[3], w[3];   // layer shapes (jpeg.py:676-686)
    int rh[3], rw[3]; // down-sampling ratios (jpeg.py:62-147)
    long long poff[3];   // element offset of layer l inside one image's plane storage
    long long pstride;   // plane elements per image = sum h*w
    int tiled;           // the normalised float32 planes (what the DCT kernels read) are stored in 4 x 4 blocks, see plane_elem(); 0 = row-major
    // CLAHE tile geometry per layer (clahe.cpp): padded size / 4
    int ctw[3], cth[3];
    // edge bit-planes (weak / strong): one 64-bit word per 64 pixels of a row, stored TILE-MAJOR: the 64 row-words of a
    // 64x64-pixel tile are contiguous (512 B), see bp_index(); rows are padded to a multiple of 64
    int wpr[3];               // words per row = ceil(w / 64)
    long long bpoff[3];       // word offset of layer l inside one image's bit-plane storage
    long long bpstride;       // words per image
};

// Quadtree geometry per layer.
struct QtGeom {
    int bmin, bmax;
    int cell;          // min(bmin, root) -- edge of a level-0 cell in pixels
    int root[3];       // root size in pixels
    int ncell[3];      // root / cell  (cells per side, power of two)
    int ltot[3];       // log2(ncell)
    long long pyr_off[3];     // byte offset of layer l's pyramid inside one image's pyramid storage
    long long pyr_stride;     // pyramid bytes per image
    int nchunk[3];            // number of 256-cell chunks = max(1, ncell^2/256)
    long long chunk_off[3];   // offset (in chunks) of layer l inside one image's chunk arrays
    long long chunk_stride;   // chunks per image
    // output layout (elements)
    long long coeff_off[3], coeff_stride;
    long long leaf_off[3], leaf_stride;
    long long state_off[3], state_stride;
    long long coeff_cap[3], leaf_cap[3], state_cap[3];
    // per-size DCT work lists: the list of plane (b, l) for size index k starts at element
    // b * work_stride[k] + work_off[l][k] of work[k] and holds work_count[(b*3+l)*kMaxSizes + k] leaves in Morton order
    int nsizes;
    long long work_off[3][kMaxSizes], work_stride[kMaxSizes];
};
// Per-context tuning / A-B options (aej_set_option, include/aej.h): the only way a caller changes which kernel or launch shape serves a
// stage.  Nothing in the library reads the environment.
struct Tuning {
    int color_strip = 1;           // "color_strip": 0 = never the persistent strip colour kernel (the 128 x 16 kernel of rounds 1-2 instead)
    int color_strip_rows = 0;      // "color_strip_rows": strip height of that kernel, 0 = automatic (16 .. 64 by batch size)
    int color_workgroups = 0;      // "color_workgroups": workgroups of the persistent launch, 0 = automatic (256 for the matrix spaces)
    int planes_row_major = 0;      // "planes_row_major": 1 = keep the normalised planes row-major (default: 4 x 4 blocks where the strip kernel can write them)
    int dct64_kernel = 0;          // "dct64_kernel": 0 = by company (DctArgs::crowded), 1 = one wave per leaf, 4 = four waves per leaf
    int dct_small_workgroups = 0;  // "dct_small_workgroups": cap on the grids of the 4 / 8 / 16 kernels, 0 = automatic
    int sobel_lds = 0;             // "sobel_lds": 1 = the LDS-tiled Sobel / NMS kernel of rounds 1-2 for every shape
    int dct_multi = 1;             // "dct_multi": 1 = calls of at most 8 Mpx run the DCTs of sizes 4 .. 64 as one launch (0: one launch per size)
    int sobel_xcd = 1;             // "sobel_xcd": 1 = each XCD works on a contiguous range of the register Sobel kernel's tiles (0: round-robin, rounds 3-4)
};

constexpr int kChunkInts = 4 + kMaxSizes;   // per-chunk record: nsym, nleaf, ncoef, pad, leaves per size
constexpr int kMaxPlanes = 3072;            // planes (3 x images) one DCT launch can address

// One unit of DCT work (a leaf), appended by the quadtree emit kernel.
// accumulator slots (doubles) per image pair of the evaluation metrics (metrics.hip): 0 = sum of squared differences;
// kMetricSlotGrey + {0, 1} = ssim / cs sums of the grey SSIM map; kMetricSlotScales + (scale * 3 + channel) * 2 + {0, 1} likewise per scale
constexpr int kMetricSlots = 40;
constexpr int kMetricSlotGrey = 2;
constexpr int kMetricSlotScales = 4;

// One unit of DCT work (a leaf) in a per-size, per-plane list segment: 8 bytes.  The plane is NOT stored: a reader finds an item's
// segment from the per-plane prefix counts anyway, and that IS the plane (round 4: 16-byte entries carried it a second time; on natural
// images, nearly all 4 x 4 leaves, the lists were half of what the quadtree's emit pass wrote and a ninth of what the 4 x 4 DCT read).
struct LeafWork {
    unsigned xy;    // origin in the layer: x | y << 16 (layers are at most 65535 pixels wide / high: make_geom)
    int coef;       // coefficient offset inside the layer's coefficient array
};
__host__ __device__ __forceinline__ LeafWork pack_work(int x, int y, int coef)
{
    LeafWork w;
    w.xy = (unsigned)x | ((unsigned)y << 16);
    w.coef = coef;
    return w;
}
// -> (plane, x, y, coef), the form the kernels work with
__device__ __forceinline__ int4 unpack_work(int plane, const LeafWork &w) { return make_int4(plane, (int)(w.xy & 0xffffu), (int)(w.xy >> 16), w.coef); }

struct DctTables {
    // per size index k (size = bmin << k)
    const float *D[kMaxSizes];        // [s][s] DCT-II basis, float32 (row k = frequency)
    const int *zzinv[kMaxSizes];      // [s*s] zigzag position of raster index
    const int *qm[3][kMaxSizes];      // [s*s] quantisation matrix per layer
};

// Element index of (y, x) in a plane of width w.  Row-major, or -- Geom::tiled, only for planes whose sides are multiples of 4 -- in 4 x 4
// blocks: blocks row-major, the four rows of a block consecutive.  A quadtree leaf of any size is then a few contiguous runs (4 x 4: one
// 64-byte sector; 8 x 8: two 128-byte lines; 64 x 64: sixteen 1 KiB runs) instead of one short piece per row of lines it shares with leaves
// of other sizes: the small-block DCT kernels read 2.5 x their pixels from row-major planes.
__host__ __device__ inline long long plane_elem(int tiled, int w, int y, int x)
{
    return tiled ? ((long long)(y >> 2) * (w >> 2) + (x >> 2)) * 16 + ((y & 3) << 2) + (x & 3) : (long long)y * w + x;
}

__host__ __device__ inline long long bp_index(int y, int xw, int wpr) { return ((long long)(y >> 6) * wpr + xw) * 64 + (y & 63); }
__host__ __device__ inline long long bp_words(int h, int wpr) { return (long long)((h + 63) / 64) * wpr * 64; }

inline int ilog2(int v) { int l = 0; while ((1 << l) < v) l++; return l; }

// Wave-wide inclusive prefix sum by DPP: four shifts inside the rows of 16 lanes, then the two row broadcasts (gfx9 has row_bcast15 /
// row_bcast31) -- six vector instructions, no LDS crossbar (`__shfl_up` is ds_bpermute: six dependent LDS round trips per scan, in
// kernels and prologues that are latency-bound; quadtree stage 0.528 -> 0.503 ms).  Lanes that would read from outside their row take
// the `old` operand, 0.  All 64 lanes must be active.
__device__ __forceinline__ int wave_scan_incl(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2 and 3
    return v;
}
// Wave-wide maximum by the same DPP steps (lanes that would read from outside their row take 0: for values >= 0).  All 64 lanes must be active.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    auto mx = [](unsigned a, int b) { return a > (unsigned)b ? a : (unsigned)b; };
    v = mx(v, __builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));      // row_shr:1
    v = mx(v, __builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));      // row_shr:2
    v = mx(v, __builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));      // row_shr:4
    v = mx(v, __builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));      // row_shr:8
    v = mx(v, __builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));      // row_bcast:15 -> rows 1 and 3
    v = mx(v, __builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));      // row_bcast:31 -> rows 2 and 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ int wave_total(int v) { return __builtin_amdgcn_readlane(wave_scan_incl(v), 63); }

}  // namespace aej

#define AEJ_HIP_CHECK(expr)                                                       \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) return aej::hip_fail(ctx, _e, #expr, __FILE__, __LINE__); \
    } while (0)
