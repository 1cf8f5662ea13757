// aej_launch.h -- host-side launch functions exported by the .hip translation units.
#pragma once
#include "aej_common.h"

namespace aej {

// color.hip
int launch_color_convert(hipStream_t st, int space, const float *rgb, float *out, long long n);
// whether the encode path may keep this geometry's normalised planes in 4 x 4 blocks (Geom::tiled): the strip kernel writes them, every DCT kernel reads both forms
bool color_planes_can_tile(const Geom &g, int space, bool in_u8, const Tuning &t);
int launch_color_planes(hipStream_t st, int space, const void *rgb, bool in_u8, const Geom &g, const float *mid, const float *scale,
                        float *raw, float *norm, unsigned char *u8, int *tile_hist, const Tuning &t);
// tables of cv.resize(INTER_AREA) for the chroma layers when the ratios are not exact 2x2 (device arrays)
struct AreaTabs {
    int mode;                  // 0: exact 2x2, 1: other integer ratios (isx, isy), 2: general tables
    int isx, isy;
    const int *xoff, *xsi, *yoff, *ysi;
    const float *xal, *yal;
};
int launch_color_planes_generic(hipStream_t st, int space, const void *rgb, bool in_u8, const Geom &g, const float *mid, const float *scale,
                                const AreaTabs &tabs, float *raw, float *norm, unsigned char *u8, int *tile_hist);
void launch_plane_u8(hipStream_t st, const float *plane, const Geom &g, unsigned char *u8, int *tile_hist);

// canny.hip
struct CannyBuffers {
    unsigned char *u8a;     // [B][pstride] scaled uint8 in; NMS / hysteresis map out
    unsigned char *u8b;     // [B][pstride] bilateral output
    int *tile_hist;         // [B][3][16][256]
    unsigned char *lut;     // [B][3][16][256]
    int *blur_hist;         // [B][3][256]
    int *thr;               // [B][3][2]
    unsigned long long *weak;    // [B][bpstride] NMS candidates (bit-plane)
    unsigned long long *strong;  // [B][bpstride] strong edges, grown by the hysteresis passes -> final edge map
    int *hflags;            // [2][B * tiles] "look at this tile again" / "queued" flags, one parity per launch of the hysteresis
    int *hlist;             // [hyst_ring_slots()] the work queue of the last launch: a ring of tile + 1 (0 = empty slot)
    int *pass_count;        // [kHystCounters] the queue's tail / head / done counters (canny.hip kQTail ...)
    const float *space_w;   // [13]
    const float *color_w;   // [256]
    // run-time hyper-parameters of EdgeDetection.canny (edge_detection.py:31-40; aej_set_canny_params)
    double low_q = 0.10 * 100, high_q = 0.30 * 100;     // np.percentile arguments
    double clip_limit = 0.75;                           // CLAHE clipLimit (<= 0: no clipping)
    int l2 = 1;                                         // cv.Canny L2gradient
    // optional stage dumps (stand-alone entry point only)
    unsigned char *dump_clahe, *dump_gauss;
};
constexpr int kHystCounters = 128;
long long hyst_tiles_per_image(const Geom &g);
int hyst_ring_slots(const Geom &g);
void launch_clahe_pad_hist(hipStream_t st, const Geom &g, const CannyBuffers &cb);
void launch_clahe_lut(hipStream_t st, const Geom &g, const CannyBuffers &cb);
void launch_clahe_blur(hipStream_t st, const Geom &g, const CannyBuffers &cb);
void launch_thresholds(hipStream_t st, const Geom &g, const CannyBuffers &cb);
void launch_sobel_nms(hipStream_t st, const Geom &g, const CannyBuffers &cb, const Tuning &t);
void launch_hysteresis(hipStream_t st, const Geom &g, const CannyBuffers &cb);      // pass over every tile + device-side drain of the work queue
void launch_zero(hipStream_t st, void *p16, size_t bytes_multiple_of_16);
void launch_bits_to_edge(hipStream_t st, const Geom &g, const unsigned long long *strong, unsigned char *edge01);
void launch_bits_to_map(hipStream_t st, const Geom &g, const unsigned long long *weak, const unsigned long long *strong, unsigned char *map);

// quadtree.hip
struct QtBuffers {
    unsigned char *pyr;     // [B][pyr_stride] (only levels >= 5 are used)
    const unsigned long long *edge_bits;   // [B][bpstride] final edge bit-plane
    int *chunk_cnt;         // [B][chunk_stride][kChunkInts]: counts, then exclusive offsets after the scan
    unsigned short *lane_code;   // [B][chunk_stride][64]: what the count pass found per lane (quadtree.hip pack_lane)
    int *leaves;            // out [B][leaf_stride][4]
    unsigned char *states;  // out [B][state_stride]
    long long *counts;      // out [B][3][4]
    LeafWork *work[kMaxSizes];   // per-size work lists (null when no DCT follows)
    int *work_count;        // [B*3][kMaxSizes] leaves per plane and size (written by the scan pass); null without DCT
    long long work_cap[kMaxSizes];
    int *overflow;          // [1] set to 1 when a capacity would be exceeded
};
void launch_qt_cells(hipStream_t st, const Geom &g, const QtGeom &q, const unsigned long long *edge_bits, const QtBuffers &qb);
void launch_pack_edge_bits(hipStream_t st, const Geom &g, const unsigned char *edge, unsigned long long *bits);
void launch_qt_count(hipStream_t st, const Geom &g, const QtGeom &q, const QtBuffers &qb);
void launch_qt_scan(hipStream_t st, const Geom &g, const QtGeom &q, const QtBuffers &qb);
void launch_qt_emit(hipStream_t st, const Geom &g, const QtGeom &q, const QtBuffers &qb);

// dct.hip
constexpr int kBigBlocks = 128;      // workgroups (and scratch slots of 256 KiB) of the 256 x 256 kernels
constexpr int kMaxBlock = 1024;      // largest block size with a kernel (the reference takes any power of two, jpeg.py:216-219; its GUI stops at 256)
// sizes >= 256 run one workgroup per leaf with the first product in a global scratch slot of S x S floats per workgroup
constexpr int big_blocks(int S) { return S <= 256 ? kBigBlocks : S == 512 ? 64 : 32; }
constexpr long long big_scratch_floats_for(int S) { return S >= 256 ? (long long)big_blocks(S) * S * S : 0; }
struct DctArgs {
    const float *norm;        // [B][pstride] normalised planes
    int *coeffs;              // out [B][coeff_stride]
    float *dct_f32;           // optional
    const LeafWork *work;     // work lists for this size (per-plane segments, see QtGeom)
    float *scratch;           // [big_blocks(S)][S * S] intermediate product of the kernels for S >= 256 (sized for the largest S), else null
    const int *work_count;    // [nplanes][kMaxSizes]
    int k;                    // size index
    int nplanes;
    const float *D;           // [s][s]
    const int *zzinv;         // [s*s]
    const int *qm[3];         // [s*s] per layer
    int crowded = 0;          // other kernels are expected beside this launch (sub-batches, calls in flight): prefer kernels that share a CU
};
int launch_dct(hipStream_t st, int size, const Geom &g, const QtGeom &q, const DctArgs &a, long long max_items, const Tuning &t);   // 0, or -1 when no kernel serves the request
// latency-sized calls: the sizes 4 .. 64 in one launch (args / max_items indexed by size index); 0 = launched, 1 = not a call for it
int launch_dct_multi(hipStream_t st, const Geom &g, const QtGeom &q, const DctArgs *args, const long long *max_items);
// builds a work list from a leaf table (stand-alone aej_dct_quant_zigzag)
void launch_work_from_leaves(hipStream_t st, const int *leaves, long long n, int bmin, int plane, LeafWork *const *work, int *work_count);


// decode.hip
struct IdctArgs {
    const int *coeffs;        // [B][coeff_stride] zigzag-ordered quantised coefficients
    float *planes;            // out [B][pstride] de-normalised layers
    const LeafWork *work;
    float *scratch;           // [big_blocks(S)][S * S] for the kernels of S >= 256, else null
    const int *work_count;    // [nplanes][kMaxSizes]
    int k, nplanes;
    const float *D;           // [s][s]
    const int *zz, *zzinv;    // zigzag order and its inverse
    const int *qm[3];
    float mid[3], scale[3];
};
void launch_work_from_tables(hipStream_t st, const Geom &g, const QtGeom &q, const int *leaves, const long long *counts, LeafWork *const *work,
                             int *work_count, int *bad /* [1], zeroed by the caller: set when the tables do not fit the plan */);
int launch_idct(hipStream_t st, int size, const Geom &g, const QtGeom &q, const IdctArgs &a, long long max_items);
int launch_color_inverse(hipStream_t st, int space, const float *in, float *out, long long n);
int launch_upsample_color(hipStream_t st, int space, const Geom &g, const float *planes, float *rgb);

// deflate.hip -- opt-in GPU entropy stage: the layers' int32 coefficients as zlib streams (jpeg.py:588-590, 659)
unsigned long long deflate_stream_bound(unsigned long long raw_bytes);
unsigned long long deflate_workspace_bytes(int batch, const long long *coeff_cap /* [3] */);
// LZ77 match search + parse of every stream: tokens into the workspace, symbol histogram (device [3][320], may be null)
void launch_deflate_parse(hipStream_t st, const int *coeffs, const long long *counts, int batch, long long coeff_stride, const long long *coeff_off,
                          const long long *coeff_cap, int *hist, void *workspace);
// the streams from the parse in the workspace (reuse_parse) or from a fresh one
void launch_deflate(hipStream_t st, const int *coeffs, const long long *counts, int batch, long long coeff_stride, const long long *coeff_off,
                    const long long *coeff_cap, const unsigned *tables /* [3][448] or null */, int reuse_parse, unsigned char *out,
                    unsigned long long stream_stride, long long *sizes, void *workspace);

// metrics.hip
void launch_metric_prep(hipStream_t st, const float *a, const float *b, int B, long long npx, double *acc, unsigned char *ga, unsigned char *gb);
void launch_metric_pool_grey(hipStream_t st, const unsigned char *ga, const unsigned char *gb, int B, int H, int W, int f, int hp, int wp, float *xa,
                             float *xb);
void launch_ssim_level(hipStream_t st, bool interleaved, const float *xa, const float *xb, int B, int C, int h, int w, const float *g11, double *acc,
                       int slot, bool want_ss, float *pool_a = nullptr, float *pool_b = nullptr);   // pool_*: scale 0 of even-sized images also writes scale 1
void launch_pool2_rgb(hipStream_t st, const float *ia, const float *ib, int B, int h, int w, int p, int h2, int w2, float *oa, float *ob);
void launch_pool2(hipStream_t st, bool interleaved, const float *in, int B, int C, int h, int w, int p, int h2, int w2, float *out);
void launch_metric_final(hipStream_t st, const double *acc, int B, long long npx, long long n_ssim, const long long *n_level, double *out);


}  // namespace aej
