/*
 * aej.h -- C ABI of libaejpeg_hip.so: the MI355X (gfx950) implementation of the adaptive-JPEG
 * ENCODE hot path of fevzibabaoglu/adaptive-edge-aware-jpeg.
 *
 * The reference is pure Python and has no FFI; its boundary for this path is the Python API
 * (src/jpeg/jpeg.py:240-272 Jpeg.compress, src/jpeg/edge_detection.py:28 EdgeDetection.canny,
 * src/jpeg/quadtree.py:71 QuadTree, src/color/conversion.py:95 convert).  Each entry point below
 * names the reference interface it replaces; INTEGRATION.md shows the ctypes binding a maintainer of
 * the reference would add.  Citations are path:line under the reference checkout.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HIP) unless the name ends in _host;
 *   - every function returns 0 on success or a negative aej_status; aej_last_error() gives the text;
 *   - no exceptions cross the ABI; nothing here allocates device memory after aej_create()
 *     except aej_set_settings() (tables) -- the caller owns inputs, outputs and the workspace;
 *   - a context is bound to one HIP device and one stream and is NOT thread-safe (like the
 *     reference's stateful Jpeg object, jpeg.py:256-259); distinct contexts are independent;
 *   - work is enqueued on the context's stream; the blocking calls synchronise it once, at their end (aej_encode_batch_begin
 *     returns without waiting); nothing inside a call depends on a value read back from the device.
 */
#ifndef AEJ_H
#define AEJ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AEJ_ABI_VERSION 3   /* 2: the hysteresis completes on the device -- the speculation entry points of version 1 are gone; aej_set_option.
                             * 3: the entropy stage keeps its parse in the workspace (aej_deflate_histogram / _batch signatures, table layout); aej_pack_u8_levels_host */

#if defined(__GNUC__)
#define AEJ_API __attribute__((visibility("default")))
#else
#define AEJ_API
#endif

typedef struct aej_ctx aej_ctx;

typedef enum aej_status {
    AEJ_OK = 0,
    AEJ_ERR_ARG = -1,        /* bad argument (ValueError on the Python side)          */
    AEJ_ERR_HIP = -2,        /* a HIP runtime call failed                             */
    AEJ_ERR_STATE = -3,      /* settings not set / plan mismatch                      */
    AEJ_ERR_CAPACITY = -4,   /* an output or the workspace is too small               */
    AEJ_ERR_UNSUPPORTED = -5 /* shape/ratio combination not built     */
} aej_status;

/* colour spaces: keys of JpegCompressionSettings.COLOR_SPACE_SETTINGS (jpeg.py:62-147) */
typedef enum aej_space {
    AEJ_YCBCR = 0, AEJ_YCOCG = 1, AEJ_YCOCG_R = 2, AEJ_OKLAB = 3, AEJ_ICTCP = 4, AEJ_ICACB = 5, AEJ_JZAZBZ = 6,
    AEJ_XYZ = 7 /* aej_color_convert / aej_color_convert_inverse only (conversion.py:63-68); not a codec space */
} aej_space;

#define AEJ_MAX_SIZES 8 /* a block-size range spans at most 8 powers of two between 2 and 1024 (the GUI offers 2..256, src/gui/main_frame.py:41-45;
                         Jpeg itself takes any power of two, jpeg.py:216-219) */

/* Shapes and capacities for one (batch, H, W) under the current settings.  Layer l of image b lives at
 * element offset  b*<x>_stride + <x>_off[l]  of the corresponding output array. */
typedef struct aej_plan {
    int32_t batch, H, W;
    int32_t layer_h[3], layer_w[3]; /* Jpeg._compute_downsampled_shapes, jpeg.py:676-686          */
    int32_t root_size[3];           /* QuadTree root size, quadtree.py:89-90                         */
    int64_t coeff_off[3], coeff_stride; /* int32 coefficients, capacity per layer = coeff_off[l+1]-coeff_off[l] */
    int64_t leaf_off[3], leaf_stride;   /* leaves, 4 x int32 each: x, y, size, coeff offset in layer  */
    int64_t state_off[3], state_stride; /* state symbols, one uint8 each: 0 leaf, 1 internal, 2 absent */
    uint64_t workspace_bytes;
} aej_plan;

/* ---- lifetime ------------------------------------------------------------------------------- */
AEJ_API int aej_abi_version(void);
AEJ_API aej_ctx *aej_create(int device, void *hip_stream /* hipStream_t or NULL = null stream */);
AEJ_API void aej_destroy(aej_ctx *ctx);
AEJ_API const char *aej_last_error(aej_ctx *ctx); /* host string owned by ctx (or a static one if ctx==NULL) */
AEJ_API int aej_synchronize(aej_ctx *ctx);
/* Re-binds the context to another stream of its device (the Python host follows torch's current stream with it, so
 * that the library's kernels are ordered behind whatever produced the caller's tensors); drains the stream it leaves. */
AEJ_API int aej_set_stream(aej_ctx *ctx, void *hip_stream);

/* Stage timing of aej_encode_batch: HIP events recorded on the context's stream around every stage of the
 * last call (measurement only; used by bench.py for the roofline figures). */
enum {
    AEJ_STAGE_CLEAR = 0, AEJ_STAGE_COLOR_PLANES, AEJ_STAGE_CLAHE_LUT, AEJ_STAGE_CLAHE_BLUR, AEJ_STAGE_THRESHOLDS,
    AEJ_STAGE_SOBEL_NMS, AEJ_STAGE_HYSTERESIS, AEJ_STAGE_QUADTREE, AEJ_STAGE_DCT_2, AEJ_STAGE_DCT_4, AEJ_STAGE_DCT_8,
    AEJ_STAGE_DCT_16, AEJ_STAGE_DCT_32, AEJ_STAGE_DCT_64, AEJ_STAGE_DCT_128, AEJ_STAGE_DCT_256, AEJ_STAGE_DCT_512, AEJ_STAGE_DCT_1024,
    AEJ_N_STAGES
};
/* Diagnostic.  The hysteresis of cv.Canny (edge_detection.py:85) is two launches whatever the image holds: a pass over every 64 x 64 tile,
 * then a device-side work queue of the tiles whose neighbourhood changed, drained to the fix-point by one persistent launch -- the host
 * guesses no pass count and reads nothing back.  out_host[2] = { whole-path calls since aej_create, tiles that went through that queue
 * in the last completed call }. */
AEJ_API int aej_get_hysteresis_stats(aej_ctx *ctx, int64_t *out_host);
/* Launch-latency path: aej_encode_batch can replay its whole launch sequence (about 20 kernel launches) as one
 * captured hipGraph, cached per (buffers, shape).  mode 0 = never (the default: on MI355X / ROCm 7.2 the replay of one 1080p encode measured 0.287 ms against 0.274 ms for
 * the eager launches), 1 = automatic (calls of at most 8 Mpx), 2 = whenever possible.  The graph runs on a private stream ordered behind the context's
 * stream; results are identical.  out_host[3] = { graph launches, graph captures, graphs cached }. */
AEJ_API int aej_set_graph_mode(aej_ctx *ctx, int mode);
/* Throughput path: aej_encode_batch cuts a large call into sub-batches (contiguous image ranges, the unit the reference's sweep
 * hands to one worker, metrics_computation.py:253) that run the whole chain on private streams, each one stage behind the previous,
 * so that the HBM-bound stages of one sub-batch (colour planes, DCT) run beside the issue-bound stages of another (blur, Sobel / NMS,
 * quadtree).  n = 0: automatic (the default: by call size -- 4 sub-batches from 384 Mpx / 16 images, 2 from 64 Mpx / 8 images -- when the
 * process runs with GPU_MAX_HW_QUEUES >= 8, so that every stream has a hardware queue of its own; with HIP's default of 4 queues: 2
 * sub-batches, and only while no other context has a call in flight), 1: never, 2..8: that many.
 * Outputs are identical; the call still returns with everything complete.  aej_encode_plan's workspace_bytes covers every split. */
AEJ_API int aej_set_sub_batches(aej_ctx *ctx, int n);
/* How many hardware queues the HIP runtime of this process maps its streams onto (GPU_MAX_HW_QUEUES as it was when the runtime
 * initialised; HIP's default is 4).  The library cannot see that value -- the environment may have been changed after the runtime
 * read it -- so the HOST states it: aej_create assumes 4 (the conservative schedule) and the Python package passes what it knows
 * (adaptive_edge_aware_jpeg_amd/_lib.py: the variable's value when the package was imported before the first HIP call, 4 otherwise).
 * aej_get_schedule_host: out_host[4] = { hardware queues assumed, sub-batches the automatic mode would use for (batch, H, W) right now,
 * 1 when fewer than 8 queues force the two-sub-batch schedule for a call the 4-sub-batch one would serve better, reserved }. */
AEJ_API int aej_set_hw_queues(aej_ctx *ctx, int n);
AEJ_API int aej_get_schedule_host(aej_ctx *ctx, int batch, int H, int W, int32_t *out_host);
/* Tuning and A / B options of one context.  The library reads NO environment variable: whatever changes which kernel or launch shape
 * serves a stage is said here, by the caller, per context (and reported back by aej_get_option).  Every option leaves the results
 * bit-identical; the defaults are the measured best on MI355X.  Unknown name or value outside the range: AEJ_ERR_ARG.
 *   name                    values             meaning
 *   "color_strip"           0 | 1 (default 1)  0: never the persistent strip colour kernel (the 128 x 16 kernel instead)
 *   "color_strip_rows"      0 | 16..64 (0)     strip height of that kernel; 0 = automatic
 *   "color_workgroups"      0..65536 (0)       workgroups of its persistent launch; 0 = automatic (256 for the matrix spaces)
 *   "planes_row_major"      0 | 1 (0)          1: normalised planes row-major in the workspace (default: 4 x 4 blocks where possible)
 *   "dct64_kernel"          0 | 1 | 4 (0)      64 x 64 DCT: 0 = by company (one wave per leaf alone, four beside other work), 1 / 4 force
 *   "dct_small_workgroups"  0..65536 (0)       cap on the grids of the 4 / 8 / 16 DCT kernels; 0 = automatic
 *   "sobel_lds"             0 | 1 (0)          1: the LDS-tiled Sobel / NMS kernel for every shape (default: register kernel when w % 4 == 0)
 *   "dct_multi"             0 | 1 (1)          1: calls of at most 8 Mpx run the DCTs of block sizes 4 .. 64 as ONE launch (latency), 0: one launch per size
 *   "sobel_xcd"             0 | 1 (1)          1: each XCD gets a contiguous range of the register Sobel kernel's tiles (0: round-robin)
 *   "sub_chain"             -1..3 (-1)         which stage of the previously enqueued part a part's colour stage waits for: 0 none, 1 colour,
 *                                              2 blur, 3 Sobel; -1 = 1 */
AEJ_API int aej_set_option(aej_ctx *ctx, const char *name, int64_t value);
AEJ_API int aej_get_option(aej_ctx *ctx, const char *name, int64_t *value_host);
/* The two halves of aej_encode_batch / aej_encode_batch_u8 (same arguments; rgb_is_u8 selects the ingest): _begin enqueues the whole
 * call and returns without waiting, _end waits for it and checks the device-side counters.  One call may
 * be in flight per context; until _end returns, the context's other entry points, the workspace and the output buffers must not be
 * used.  Two contexts on two streams, each with its own buffers, keep the GPU busy across calls: while one call drains (hysteresis
 * tail, DCT) the other's colour stage runs (bench.py times this pipeline; the strictly serial figure is reported beside it). */
AEJ_API int aej_encode_batch_begin(aej_ctx *ctx, const void *rgb, int rgb_is_u8, int batch, int H, int W, int32_t *coeffs, int32_t *leaves,
                                   uint8_t *states, int64_t *counts, float *dct_f32, void *workspace, uint64_t workspace_bytes);
AEJ_API int aej_encode_batch_end(aej_ctx *ctx);
AEJ_API int64_t aej_get_split_calls(aej_ctx *ctx);   /* calls since aej_create that ran as sub-batches */
AEJ_API int aej_get_graph_stats(aej_ctx *ctx, int64_t *out_host);
AEJ_API int aej_set_profiling(aej_ctx *ctx, int enable);
AEJ_API int aej_get_stage_ms(aej_ctx *ctx, float *ms_host /* [AEJ_N_STAGES] */);
AEJ_API const char *aej_stage_name(int stage);

/* ---- settings: JpegCompressionSettings + Jpeg.precompute_caches (jpeg.py:150-174, 216-238) ------
 * qmats_host: the integer quantisation matrices of Jpeg._get_quantization_matrix (jpeg.py:707-724),
 * built by the Python host, laid out [layer 0..2][size = bmin, 2*bmin, ..., bmax][size*size] int32.
 * Zigzag orders (jpeg.py:726-766) and DCT bases are derived inside the library. */
AEJ_API int aej_set_settings(aej_ctx *ctx, int space, int bmin, int bmax, const int32_t *qmats_host);

/* ---- whole path: Jpeg.compress up to and including the zigzag gather (jpeg.py:262-270, 579-588) --
 * Any image size up to 65535 pixels a side (AEJ_ERR_UNSUPPORTED above; the reference has no limit of its own), at most 1024 images a call. */
AEJ_API int aej_encode_plan(aej_ctx *ctx, int batch, int H, int W, aej_plan *plan_host);

/* rgb: [batch][H][W][3] float32 in [0,1] (Image.data, image.py:26-36).
 * coeffs / leaves / states: laid out as the plan says.  counts: [batch][3][4] int64 =
 * {n_coeffs, n_leaves, n_states, root_size}.  dct_f32 (optional, may be NULL): pre-quantisation
 * DCT values in raster order per leaf, same offsets as coeffs. */
AEJ_API int aej_encode_batch(aej_ctx *ctx, const float *rgb, int batch, int H, int W,
                     int32_t *coeffs, int32_t *leaves, uint8_t *states, int64_t *counts,
                     float *dct_f32, void *workspace, uint64_t workspace_bytes);

/* Same call for images that are still 8-bit: rgb_u8 is [batch][H][W][3] uint8 and the library forms
 * float32(v) / 255.0f itself (image.py:80, Image.load: `iio.imread(path).astype(np.float32) / 255.0`), so the
 * result is identical to aej_encode_batch on that float image while the input costs 3 B/px instead of 12. */
AEJ_API int aej_encode_batch_u8(aej_ctx *ctx, const uint8_t *rgb_u8, int batch, int H, int W,
                     int32_t *coeffs, int32_t *leaves, uint8_t *states, int64_t *counts,
                     float *dct_f32, void *workspace, uint64_t workspace_bytes);

/* ---- EvaluationMetrics(original, compressed).psnr() / .ssim() / .ms_ssim() (evaluation_metrics.py:50-89) for a batch
 * of image pairs.  img_a, img_b: [batch][H][W][3] float32 in [0,1].  out: device [batch][3] float64 = {psnr (dB), ssim of
 * the 8-bit grey images, ms_ssim}; entries not requested in `which` are NaN.  The metrics are piq 0.8.0's (requirements.txt:22)
 * with the reference's arguments; AEJ_ERR_ARG where piq raises ValueError (image smaller than the 11x11 window after
 * pooling, or smaller than 161x161 for MS-SSIM).  lpips() is not offered: its AlexNet weights are a download. */
enum { AEJ_METRIC_PSNR = 1, AEJ_METRIC_SSIM = 2, AEJ_METRIC_MS_SSIM = 4 };
AEJ_API uint64_t aej_metrics_workspace_bytes(int batch, int H, int W);
AEJ_API int aej_metrics_batch(aej_ctx *ctx, const float *img_a, const float *img_b, int batch, int H, int W, int which,
                      double *out, void *workspace, uint64_t workspace_bytes);

/* ---- stage entry points (same kernels; used by the Python mirrors and the parity tests) -------- */

/* color.convert("sRGB", space, x)  (conversion.py:95-124): rgb [n][3] -> out [n][3], float32 */
AEJ_API int aej_color_convert(aej_ctx *ctx, int space, const float *rgb, float *out, int64_t n);

/* Jpeg._convert_color_space + _downsample (jpeg.py:262-267): raw (un-normalised) float32 planes,
 * planes[l] has layer_h[l]*layer_w[l] elements at offset plane_off[l]; total = plane_stride per image */
AEJ_API int aej_color_planes(aej_ctx *ctx, const float *rgb, int batch, int H, int W, float *planes_raw,
                     float *planes_norm, uint8_t *planes_u8);

/* EdgeDetection.canny(img2d) (edge_detection.py:28-86): plane float32 [H][W] -> edge uint8 {0,1}.
 * stages (optional): 5*H*W uint8 = scaled, CLAHE, Gaussian, bilateral, NMS map (0 weak,1 none,2 strong);
 * thresholds (optional): 2 int32 = the integer low/high passed to the NMS test. */
AEJ_API uint64_t aej_canny_workspace_bytes(int H, int W);
/* The keyword arguments of EdgeDetection.canny (edge_detection.py:31-40) that are run-time values of the kernels; they apply to
 * aej_canny and to the Canny stage of aej_encode_batch until changed (NULL = the reference's defaults 0.10, 0.30, 0.75, 75, 75,
 * true, which is what Jpeg._block_split uses, jpeg.py:376).  aperture_size = 3, clahe_tile_grid = (4, 4),
 * bilateral_diameter = 5 and gaussian_kernel = 3 are structural (stencil shapes) and not selectable. */
typedef struct aej_canny_params {
    double canny_low_ratio, canny_high_ratio;     /* np.percentile(blur, ratio * 100) */
    double clahe_clip_limit;                      /* cv.createCLAHE(clipLimit=...); <= 0 disables clipping */
    double bilateral_sigma_color, bilateral_sigma_space;
    int use_l2_gradient;                          /* cv.Canny(L2gradient=...) */
} aej_canny_params;
AEJ_API int aej_set_canny_params(aej_ctx *ctx, const aej_canny_params *params);
AEJ_API int aej_canny(aej_ctx *ctx, const float *plane, int H, int W, uint8_t *edge, uint8_t *stages,
              int32_t *thresholds, void *workspace, uint64_t workspace_bytes);

/* QuadTree(edge, max_size, min_size).get_leaves_and_states() (quadtree.py:71-165).
 * edge: uint8 [H][W], non-zero == edge.  leaves: [cap][4] int32 (x, y, size, coeff offset);
 * states: uint8 symbols; counts: 4 int64 = {n_coeffs, n_leaves, n_states, root_size}. */
AEJ_API uint64_t aej_quadtree_workspace_bytes(int H, int W, int min_size, int max_size);
AEJ_API int aej_quadtree_capacity(int H, int W, int min_size, int max_size, int64_t *leaf_cap, int64_t *state_cap,
                          int64_t *coeff_cap);
AEJ_API int aej_quadtree(aej_ctx *ctx, const uint8_t *edge, int H, int W, int min_size, int max_size,
                 int32_t *leaves, uint8_t *states, int64_t *counts, void *workspace, uint64_t workspace_bytes);

/* gather + reflect pad + DCT + quantise + zigzag (jpeg.py:393-404, 471, 499-502, 579-588) for the leaves
 * of one layer.  norm: normalised plane [H][W]; leaves: [n][4] as produced by aej_quadtree;
 * coeffs: sum(size^2) int32; dct_f32 optional. Uses the quantisation matrices of `layer` from the
 * current settings. */
AEJ_API int aej_dct_quant_zigzag(aej_ctx *ctx, const float *norm, int H, int W, int layer, const int32_t *leaves,
                         int64_t n_leaves, int32_t *coeffs, float *dct_f32);

/* ---- OPT-IN entropy stage on the GPU: every layer's coefficient array as a zlib stream ---------------------------
 * The container's per-layer streams are `zlib.compress(coeffs.tobytes(), level=9)` in the reference (jpeg.py:588-590) and are read back
 * with `zlib.decompress` (jpeg.py:659) -- which accepts ANY conforming zlib stream.  With the hot path on the GPU that host call is the
 * whole of Jpeg.compress end to end, so the library can write the streams itself (csrc/deflate.hip): ONE deflate block per stream, LZ77
 * matches found by an exact hash-chain search over 32 KiB chunks (window: the chunk), tokens chosen by dynamic programming over a static
 * cost model, coded with RFC 1951's fixed code or -- when `tables` is given, complete for the stream and smaller -- a dynamic code per
 * layer that the HOST builds from the token histogram (aej_deflate_build_tables).  The bytes differ from zlib level 9's (about 1.1 x larger
 * on natural images with the dynamic codes: zlib level 6's size); the decoded coefficients are identical.  Never the default.
 * coeffs / counts: aej_encode_batch's outputs (counts is the DEVICE array; a count outside its layer's capacity returns AEJ_ERR_ARG).
 * aej_deflate_histogram: match search + parse of every stream.  The tokens stay in `workspace`; hist = device [3][AEJ_DEFLATE_HIST_BINS]
 *   int32: per layer (summed over the batch) the occurrences of the 286 literal / length symbols, then of the 30 distance symbols.
 * aej_deflate_batch: tables = device [3][AEJ_DEFLATE_TABLE_WORDS] uint32 or NULL (fixed code everywhere).  reuse_parse != 0: the tokens
 *   aej_deflate_histogram left in this workspace for these very coeffs / counts are used (the usual sequence: histogram -> build tables ->
 *   batch); 0: the streams are parsed again first.  The stream of (image b, layer l) is written at streams + (3 b + l) * stream_stride
 *   (4-byte aligned, stride a multiple of 4), its length to sizes[3 b + l] (device int64).  aej_deflate_stream_bound(raw bytes) is always
 *   enough for stream_stride; a stream that does not fit returns AEJ_ERR_CAPACITY.  Both calls synchronise the context's stream. */
#define AEJ_DEFLATE_HIST_BINS 320
#define AEJ_DEFLATE_TABLE_WORDS 448
AEJ_API uint64_t aej_deflate_stream_bound(uint64_t raw_bytes);
AEJ_API uint64_t aej_deflate_workspace_bytes(aej_ctx *ctx, int batch, int H, int W);
AEJ_API int aej_deflate_histogram(aej_ctx *ctx, const int32_t *coeffs, const int64_t *counts, int batch, int H, int W, int32_t *hist, void *workspace,
                                  uint64_t workspace_bytes);
/* HOST-only helper (no context, no device work): the three layers' dynamic codes from the histograms above, copied to the host --
 * hist_host [3][AEJ_DEFLATE_HIST_BINS], tables_host [3][AEJ_DEFLATE_TABLE_WORDS].  Layout of a table: [0..285] literal / length symbols as
 * `bit-reversed code | nbits << 16`, [286..315] the distance symbols likewise, [316] number of block-header bits, [317..] those bits, LSB
 * first (BFINAL = 1, BTYPE = 10, HLIT / HDIST / HCLEN, the run-length coded code lengths).  tests/deflate_reference.py is the readable
 * restatement (tests compare the two word for word).  cover_all [3] or NULL (= all 1): 1 = every symbol gets a code (a table valid for
 * any data), 0 = only the symbols counted (shorter block headers; a stream that needs a missing code is written with the fixed code). */
AEJ_API int aej_deflate_build_tables(const int32_t *hist_host, const int32_t *cover_all, uint32_t *tables_host);
AEJ_API int aej_deflate_batch(aej_ctx *ctx, const int32_t *coeffs, const int64_t *counts, int batch, int H, int W, const uint32_t *tables, int reuse_parse,
                              uint8_t *streams, uint64_t stream_stride, int64_t *sizes, void *workspace, uint64_t workspace_bytes);

/* HOST-only helper (no context, no device work) for the 8-bit ingest of HOST images (src/image/image.py:80 makes every
 * loaded image `uint8.astype(float32) / 255.0`): if every one of the n float32 values at rgb_host is bit for bit float32(k) / 255.0f for
 * some k in 0..255, the levels k are written to u8_host and 1 is returned -- the caller then uploads 3 B per pixel instead of 12 and
 * calls aej_encode_batch_u8, whose outputs are identical; otherwise 0 (u8_host is then undefined; the pass stops at the first block that
 * holds another value).  `threads` host threads share the pass (clamped to 1..64). */
AEJ_API int aej_pack_u8_levels_host(const float *rgb_host, int64_t n, uint8_t *u8_host, int threads);

/* ---- decode path: Jpeg.decompress after the host-side entropy decode (jpeg.py:285-296) ------
 * coeffs / leaves / counts use the layout of aej_encode_batch's outputs (so an encoded batch can be decoded in
 * place); counts is a DEVICE array [batch][3][4], only n_leaves is read.  rgb_out: [batch][H][W][3] float32 in [0,1]
 * (Jpeg._dequantize, _apply_inverse_dct, _block_merge, _upsample, _convert_color_space_inverse).
 * The leaf tables must tile every layer (aej_encode_batch's output does; Jpeg.decompress checks a parsed stream on the host
 * with aej_leaf_positions_host): a table that does not fit the plan -- n_leaves beyond the layer's capacity, a size outside
 * the settings' block range, an origin outside the layer, a coefficient offset outside the layer's span, more leaves of one size
 * than can exist -- returns AEJ_ERR_ARG instead of touching memory it
 * does not own; pixels no leaf covers are left undefined (the reference starts from np.zeros, jpeg.py:421). */
AEJ_API uint64_t aej_decode_workspace_bytes(aej_ctx *ctx, int batch, int H, int W);
AEJ_API int aej_decode_batch(aej_ctx *ctx, const int32_t *coeffs, const int32_t *leaves, const int64_t *counts, int batch, int H,
                             int W, float *rgb_out, void *workspace, uint64_t workspace_bytes);
/* color.convert(space, "sRGB", x) (conversion.py:122-124): in [n][3] -> sRGB [n][3], float32 */
AEJ_API int aej_color_convert_inverse(aej_ctx *ctx, int space, const float *in, float *out_rgb, int64_t n);
/* HOST helper (no device): leaf positions from leaf sizes, the walk of Jpeg._block_merge (jpeg.py:424-448).
 * sizes_host [n] -> xy_host [n][2]; returns the number of leaves placed (== n for a well-formed stream), or -2 when the n
 * leaves leave a part of the layer uncovered. */
AEJ_API int64_t aej_leaf_positions_host(const int32_t *sizes_host, int64_t n, int root, int H, int W, int32_t *xy_host);

#ifdef __cplusplus
}
#endif
#endif /* AEJ_H */
