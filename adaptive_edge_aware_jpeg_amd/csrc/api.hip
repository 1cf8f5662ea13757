// api.hip -- the C ABI of libaejpeg_hip.so (include/aej.h): context, settings/tables, workspace carving and
// the kernel sequence of the encode hot path.  Host code only; gfx950 kernels live in the other .hip files.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/aej.h"
#include "../../include/aej_testing.h"
#include "aej_common.h"
#include "aej_launch.h"

using namespace aej;

struct aej_pending;
constexpr int kFlagWords = 16;       // [0] quadtree overflow flag, [1] entries that went through the hysteresis work queue (diagnostic)
struct aej_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool has_settings = false;
    int space = 0, bmin = 0, bmax = 0, nsizes = 0;
    void *tables = nullptr;            // one device allocation holding every table below
    const float *d_D[kMaxSizes] = {};
    const int *d_zzinv[kMaxSizes] = {};
    const int *d_zz[kMaxSizes] = {};
    const int *d_qm[3][kMaxSizes] = {};
    const float *d_space_w = nullptr, *d_color_w = nullptr;
    float *d_bilateral = nullptr;      // [16 + 256] space / colour weights of the bilateral filter (own allocation: aej_set_canny_params rebuilds it)
    aej_canny_params canny = { 0.10, 0.30, 0.75, 75.0, 75.0, 1 };     // edge_detection.py:31-40 defaults
    int *h_flag = nullptr;             // pinned host words for counter read-backs (kFlagWords)
    long long last_hyst_queued = 0;    // tiles that went through the hysteresis work queue in the last whole-path call (diagnostic)
    bool capturing = false;            // the stream is being captured into a hipGraph: kernel nodes only (zero-fill by kernel, no copies)
    long long n_encode_calls = 0;      // aej_get_hysteresis_stats
    // launch-latency path (aej_set_graph_mode): the whole launch sequence of one encode call captured in a hipGraph,
    // keyed by everything its kernel arguments depend on, and replayed on a private stream
    int graph_mode = 0;                // 0 off (default: measured slower than eager launches), 1 automatic (small batches only), 2 always when possible
    hipStream_t gstream = nullptr;
    hipEvent_t gevent = nullptr;
    struct GraphEntry {
        const void *rgb; void *coeffs, *leaves, *states, *counts, *dct, *ws;
        int batch, H, W, in_u8;
        hipGraphExec_t exec;
        unsigned long long last_use;
    };
    std::vector<GraphEntry> graphs;
    unsigned long long graph_clock = 0, n_graph_launches = 0, n_graph_captures = 0;
    // sub-batch pipelining (aej_set_sub_batches): a large call is cut into sub-batches that run the whole chain on private streams,
    // each one stage behind the previous, so that HBM-bound stages (colour planes, DCT) of one run beside the issue-bound stages
    // (blur, Sobel / NMS, quadtree) of another
    int sub_mode = 0;                  // 0 automatic, 1 never split, n > 1 split into n (when the batch allows)
    int hw_queues = 4;                 // hardware queues the runtime maps streams onto, as the host states it (aej_set_hw_queues; HIP's default 4): streams beyond it share queues
    int fail_after = -1;               // aej_test_fail_after_stage (test instrumentation)
    static constexpr int kMaxSub = 8;
    int dct_crowded = 0;               // this call runs as sub-batches or beside other calls: DCT kernels that share CUs (aej_launch.h DctArgs::crowded)
    hipStream_t sub_stream[kMaxSub] = {};
    hipEvent_t sub_color_done[kMaxSub] = {}, sub_in = nullptr;
    int *sub_flag[kMaxSub] = {};       // pinned read-back words per sub-batch (layout of h_flag)
    long long n_split_calls = 0;
    int sub_chain = -1;                // colour stages wait for a stage of the previous part (g_last_color_done): 1 its colour stage, 2 its blur, 3 its
                                       // Sobel / NMS; 0 no staggering; -1 (default) = 1, between the sub-batches of one call and between whole calls on
                                       // rotating contexts alike (round 4, profiles/r04_sched_sweep_final.txt: unsplit 64 x 4K calls on three contexts
                                       // 6.01 / 7.58 / 7.49 / 6.22 ms for 1 / 2 / 3 / 0; 6 x 4K 0.66 / 0.75 / 0.85 / 0.68; round 3's kernels had preferred 2
                                       // between whole calls).  aej_set_option "sub_chain" overrides.
    int chain_hook = 0;                // run_canny_chain publishes chain_event after the blur (2) / Sobel (3) stage of the part being enqueued
    hipEvent_t chain_event = nullptr;
    Tuning tune;                       // aej_set_option: kernel / launch-shape choices (nothing in the library reads the environment)
    struct aej_pending *pending = nullptr;     // the call between aej_encode_batch_begin and aej_encode_batch_end
    // optional stage timing (aej_set_profiling): events on ctx->stream around each stage of aej_encode_batch
    bool profiling = false;
    hipEvent_t ev[24] = {};
    int ev_stage[24] = {};
    int n_ev = 0;
    float stage_ms[AEJ_N_STAGES] = {};
};

namespace aej {
static int fail(aej_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}
int hip_fail(aej_ctx *ctx, hipError_t e, const char *expr, const char *file, int line)
{
    return fail(ctx, AEJ_ERR_HIP, "%s failed: %s (%s:%d)", expr, hipGetErrorString(e), file, line);
}
}  // namespace aej

static void mark(aej_ctx *ctx, int stage)
{
    if (!ctx->profiling || ctx->n_ev >= 24) return;
    if (!ctx->ev[ctx->n_ev] && hipEventCreate(&ctx->ev[ctx->n_ev]) != hipSuccess) return;
    ctx->ev_stage[ctx->n_ev] = stage;   // the stage that ENDS at this event
    (void)hipEventRecord(ctx->ev[ctx->n_ev], ctx->stream);
    ctx->n_ev++;
}

extern "C" const char *aej_stage_name(int i);
// test instrumentation (aej_test_fail_after_stage): one-shot failure right after `stage` has been enqueued
static int injected_failure(aej_ctx *ctx, int stage)
{
    if (ctx->fail_after != stage) return 0;
    ctx->fail_after = -1;
    ctx->err = std::string("injected failure after stage ") + aej_stage_name(stage);
    return AEJ_ERR_STATE;
}

static void collect_marks(aej_ctx *ctx)
{
    for (int i = 0; i < AEJ_N_STAGES; i++) ctx->stage_ms[i] = 0.f;
    for (int i = 1; i < ctx->n_ev; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->ev[i - 1], ctx->ev[i]) == hipSuccess && ctx->ev_stage[i] >= 0) ctx->stage_ms[ctx->ev_stage[i]] += ms;
    }
}

// The colour stage of every encode part (a whole call or a sub-batch) waits for the colour stage of the part enqueued before it on
// the same device -- by any context -- and publishes its own completion here.  Within one call this staggers the sub-batches;
// across contexts it keeps two calls in flight out of phase (begun together they would run their HBM-bound stages side by side
// and their issue-bound stages side by side, which gains nothing; one stage apart, colour planes / DCT of one run beside blur /
// Sobel of the other).  Waiting on an event that has long completed costs nothing.
#include <mutex>
static std::mutex g_chain_mutex;
constexpr int kMaxDevices = 64;
static hipEvent_t g_last_color_done[kMaxDevices] = {};      // per device (aej_create refuses device >= kMaxDevices); owned by the context that recorded it
static int g_calls_in_flight[kMaxDevices] = {};             // per device: calls between aej_encode_batch_begin and _end (guarded by g_chain_mutex)

static void free_pending(aej_ctx *ctx);      // defined with aej_pending
static bool call_in_flight(const aej_ctx *ctx);

static void drop_graphs(aej_ctx *ctx)
{
    for (auto &e : ctx->graphs) if (e.exec) (void)hipGraphExecDestroy(e.exec);
    ctx->graphs.clear();
}

// ---- constant tables ---------------------------------------------------------------------------------
// down-sampling ratios (rh, rw) per layer: JpegCompressionSettings.COLOR_SPACE_SETTINGS, jpeg.py:62-147
static const int kRatios[7][3][2] = {
    { { 1, 1 }, { 2, 2 }, { 2, 2 } },  // YCbCr
    { { 1, 1 }, { 2, 2 }, { 2, 2 } },  // YCoCg
    { { 1, 1 }, { 2, 2 }, { 2, 2 } },  // YCoCg-R
    { { 1, 1 }, { 2, 2 }, { 2, 2 } },  // OKLAB
    { { 1, 1 }, { 1, 4 }, { 1, 4 } },  // ICtCp
    { { 1, 1 }, { 1, 4 }, { 1, 4 } },  // ICaCb
    { { 1, 1 }, { 2, 2 }, { 2, 2 } },  // JzAzBz
};
// MIDPOINTS / SCALE_FACTORS: float32 of the Python literals (ycbcr.py:41-42, ycocg.py:41-42,62-63, oklab.py:51-52,
// ictcp.py:162-163, icacb.py:162-163, jzazbz.py:211-212)
static const double kMid[7][3] = {
    { 0.5000000037252903, 7.450580596923828e-09, 0.0 }, { 0.5, 0.0, 0.0 }, { 0.5, 0.0, 0.0 },
    { 0.4999999, 0.021152213, -0.056563325 }, { 0.07497266, -0.0008235276, 0.023989676 },
    { 0.07498085, 0.02180194, -0.018250957 }, { 0.0087900255, 0.00048353244, -0.0020741792 },
};
static const double kScale[7][3] = {
    { 253.99999810755253, 254.000003784895, 254.0 }, { 254.0, 254.0, 254.0 }, { 254.0, 127.0, 127.0 },
    { 254.00005, 497.9055, 497.94604 }, { 1693.9674, 1133.9044, 1694.004 },
    { 1693.7823, 1838.5665, 1330.3855 }, { 14448.194, 7590.505, 5552.201 },
};

static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static long long align_up(long long v, long long a) { return (v + a - 1) / a * a; }

// quadtree.py:89-90 + utils.py:36-41: largest_power_of_2(max(H, W)) * 2
static int root_size_of(int h, int w)
{
    int n = h > w ? h : w;
    int lp;
    if (n <= 2) lp = n;
    else { lp = 1; while (lp * 2 < n) lp *= 2; }
    return lp * 2;
}

static void fill_clahe_geom(Geom &g)
{
    long long bp = 0;
    for (int l = 0; l < g.nl; l++) {
        g.wpr[l] = (g.w[l] + 63) / 64;
        g.bpoff[l] = bp;
        bp += bp_words(g.h[l], g.wpr[l]);
    }
    g.bpstride = bp;
    for (int l = 0; l < g.nl; l++) {
        int w = g.w[l], h = g.h[l];
        int wp = w, hp = h;
        if ((w % 4) != 0 || (h % 4) != 0) { wp = w + (4 - w % 4); hp = h + (4 - h % 4); }   // clahe.cpp copyMakeBorder
        g.ctw[l] = wp / 4;
        g.cth[l] = hp / 4;
    }
}

static int make_geom(aej_ctx *ctx, int space, int B, int H, int W, Geom &g)
{
    memset(&g, 0, sizeof g);
    if (H > 65535 || W > 65535) return fail(ctx, AEJ_ERR_UNSUPPORTED, "image %dx%d: sides above 65535 pixels are not built (leaf origins travel as 16-bit pairs, LeafWork)", H, W);
    g.B = B; g.nl = 3; g.H = H; g.W = W;
    long long off = 0;
    for (int l = 0; l < 3; l++) {
        g.rh[l] = kRatios[space][l][0];
        g.rw[l] = kRatios[space][l][1];
        g.h[l] = H / g.rh[l];
        g.w[l] = W / g.rw[l];
        if (g.h[l] < 1 || g.w[l] < 1) return fail(ctx, AEJ_ERR_ARG, "image %dx%d too small for the down-sampling ratios", H, W);
        g.poff[l] = off;
        off += align_up((long long)g.h[l] * g.w[l], 64);
    }
    g.pstride = off;
    fill_clahe_geom(g);
    return 0;
}

static void make_plane_geom(int H, int W, Geom &g)   // one stand-alone plane as "layer 0"
{
    memset(&g, 0, sizeof g);
    g.B = 1; g.nl = 1; g.H = H; g.W = W;
    g.h[0] = H; g.w[0] = W; g.rh[0] = g.rw[0] = 1;
    g.poff[0] = 0;
    g.pstride = align_up((long long)H * W, 64);
    fill_clahe_geom(g);
}

static int make_qtgeom(aej_ctx *ctx, const Geom &g, int bmin, int bmax, QtGeom &q, bool allow_small_root = false)
{
    memset(&q, 0, sizeof q);
    if (!is_pow2(bmin) || !is_pow2(bmax) || bmin > bmax || bmin < 1)
        return fail(ctx, AEJ_ERR_ARG, "block sizes must be powers of two with min <= max (got %d, %d)", bmin, bmax);
    q.bmin = bmin; q.bmax = bmax; q.cell = bmin;
    long long pyr = 0, chunks = 0, co = 0, lo = 0, so = 0;
    for (int l = 0; l < g.nl; l++) {
        int root = root_size_of(g.h[l], g.w[l]);
        if (root < bmin) {
            // the whole layer is one leaf of size `root` (quadtree.py:116-118); the codec has no tables for that size
            if (!allow_small_root || g.nl != 1) return fail(ctx, AEJ_ERR_UNSUPPORTED, "layer %d (%dx%d): root %d smaller than min block %d", l, g.h[l], g.w[l], root, bmin);
            q.cell = root;
        }
        q.root[l] = root;
        q.ncell[l] = root / q.cell;
        q.ltot[l] = ilog2(q.ncell[l]);
        if (q.ltot[l] > 14) return fail(ctx, AEJ_ERR_UNSUPPORTED, "quadtree deeper than 14 levels");
        q.pyr_off[l] = pyr;
        long long states = 0;
        for (int j = 0; j <= q.ltot[l]; j++) { long long s = q.ncell[l] >> j; states += s * s; }
        pyr += align_up(states, 256);
        long long nc2 = (long long)q.ncell[l] * q.ncell[l];
        q.nchunk[l] = (int)(nc2 / 256 > 0 ? nc2 / 256 : 1);
        q.chunk_off[l] = chunks;
        chunks += q.nchunk[l];
        int top = bmax < root ? bmax : root;
        long long wc = align_up(g.w[l], top), hc = align_up(g.h[l], top);
        if (wc > root) wc = root;
        if (hc > root) hc = root;
        q.coeff_cap[l] = wc * hc;
        q.leaf_cap[l] = (wc / q.cell) * (hc / q.cell);
        q.state_cap[l] = states;
        q.coeff_off[l] = co; co += align_up(q.coeff_cap[l], 64);
        q.leaf_off[l] = lo;  lo += align_up(q.leaf_cap[l], 16);
        q.state_off[l] = so; so += align_up(q.state_cap[l], 64);
    }
    q.pyr_stride = pyr; q.chunk_stride = chunks;
    q.coeff_stride = co; q.leaf_stride = lo; q.state_stride = so;
    // per-size work lists: worst-case leaves of size s per layer
    int k = 0;
    for (int s = bmin; s <= bmax && k < kMaxSizes; s *= 2, k++) {
        long long off = 0;
        for (int l = 0; l < g.nl; l++) {
            q.work_off[l][k] = off;
            if (s > q.root[l]) continue;
            int top = bmax < q.root[l] ? bmax : q.root[l];
            long long wc = align_up(g.w[l], top), hc = align_up(g.h[l], top);
            if (wc > q.root[l]) wc = q.root[l];
            if (hc > q.root[l]) hc = q.root[l];
            off += ((wc + s - 1) / s) * ((hc + s - 1) / s);
        }
        q.work_stride[k] = off;
    }
    q.nsizes = k;
    return 0;
}

// ---- workspace carving ----------------------------------------------------------------------------------
struct Carver {
    char *base;
    unsigned long long off = 0;
    explicit Carver(void *p) : base(static_cast<char *>(p)) {}
    template <typename T> T *take(long long n)
    {
        off = (off + 255) & ~255ull;
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += (unsigned long long)n * sizeof(T);
        return p;
    }
};

struct CannyWs {
    CannyBuffers cb;
    char *zero_begin, *zero_end;     // region cleared at the start of every call
};

static void carve_canny(Carver &c, const Geom &g, CannyWs &w)
{
    memset(&w, 0, sizeof w);
    long long planes = (long long)g.B * g.pstride;
    long long tiles = hyst_tiles_per_image(g) * g.B;
    w.cb.u8a = c.take<unsigned char>(planes);
    w.cb.u8b = c.take<unsigned char>(planes);
    w.cb.weak = c.take<unsigned long long>((long long)g.B * g.bpstride);
    w.cb.strong = c.take<unsigned long long>((long long)g.B * g.bpstride);
    w.cb.lut = c.take<unsigned char>((long long)g.B * 3 * 16 * 256);
    w.cb.thr = c.take<int>((long long)g.B * 3 * 2);
    w.zero_begin = reinterpret_cast<char *>(c.take<int>(0));
    w.cb.hlist = c.take<int>(hyst_ring_slots(g));
    w.cb.tile_hist = c.take<int>((long long)g.B * 3 * 16 * 256);
    w.cb.blur_hist = c.take<int>((long long)g.B * 3 * 256);
    w.cb.pass_count = c.take<int>(kHystCounters);
    w.cb.hflags = c.take<int>(2 * tiles);
    c.take<int>(0);
    w.zero_end = c.base ? c.base + c.off : nullptr;
}

struct QtWs {
    QtBuffers qb;
    char *zero_begin, *zero_end;
};

static void carve_qt(Carver &c, const Geom &g, const QtGeom &q, bool with_work, QtWs &w)
{
    memset(&w, 0, sizeof w);
    w.zero_begin = reinterpret_cast<char *>(c.take<int>(0));
    w.qb.pyr = c.take<unsigned char>((long long)g.B * q.pyr_stride);
    w.qb.overflow = c.take<int>(1);
    c.take<int>(0);
    w.zero_end = c.base ? c.base + c.off : nullptr;
    w.qb.chunk_cnt = c.take<int>((long long)g.B * q.chunk_stride * kChunkInts);
    w.qb.lane_code = c.take<unsigned short>((long long)g.B * q.chunk_stride * 64);
    if (with_work) {
        w.qb.work_count = c.take<int>((long long)g.B * 3 * kMaxSizes);
        for (int k = 0; k < q.nsizes; k++) {
            long long cap = q.work_stride[k] * g.B;
            w.qb.work_cap[k] = cap;
            w.qb.work[k] = c.take<LeafWork>(cap > 0 ? cap : 1);
        }
    } else {
        w.qb.work_count = nullptr;
    }
}

// ---- lifetime ---------------------------------------------------------------------------------------------
extern "C" int aej_abi_version(void) { return AEJ_ABI_VERSION; }

extern "C" aej_ctx *aej_create(int device, void *hip_stream)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n || device >= kMaxDevices) return nullptr;      // (the per-device chain state below is indexed by it)
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    aej_ctx *ctx = new aej_ctx();
    ctx->device = device;
    ctx->stream = static_cast<hipStream_t>(hip_stream);
    if (hipHostMalloc(reinterpret_cast<void **>(&ctx->h_flag), kFlagWords * sizeof(int), hipHostMallocDefault) != hipSuccess) { delete ctx; return nullptr; }
    return ctx;
}

extern "C" void aej_destroy(aej_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (call_in_flight(ctx)) {
        (void)hipDeviceSynchronize();
        std::lock_guard<std::mutex> lock(g_chain_mutex);
        if (g_calls_in_flight[ctx->device] > 0) g_calls_in_flight[ctx->device]--;
    }
    free_pending(ctx);
    {
        std::lock_guard<std::mutex> lock(g_chain_mutex);
        for (int i = 0; i < aej_ctx::kMaxSub; i++)
            if (ctx->sub_color_done[i] && g_last_color_done[ctx->device] == ctx->sub_color_done[i]) g_last_color_done[ctx->device] = nullptr;
    }
    drop_graphs(ctx);
    if (ctx->gstream) (void)hipStreamDestroy(ctx->gstream);
    if (ctx->gevent) (void)hipEventDestroy(ctx->gevent);
    if (ctx->tables) (void)hipFree(ctx->tables);
    if (ctx->d_bilateral) (void)hipFree(ctx->d_bilateral);
    for (int i = 0; i < aej_ctx::kMaxSub; i++) {
        if (ctx->sub_stream[i]) (void)hipStreamDestroy(ctx->sub_stream[i]);
        if (ctx->sub_color_done[i]) (void)hipEventDestroy(ctx->sub_color_done[i]);
        if (ctx->sub_flag[i]) (void)hipHostFree(ctx->sub_flag[i]);
    }
    if (ctx->sub_in) (void)hipEventDestroy(ctx->sub_in);
    if (ctx->h_flag) (void)hipHostFree(ctx->h_flag);
    for (int i = 0; i < 24; i++) if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    delete ctx;
}

extern "C" const char *aej_last_error(aej_ctx *ctx)
{
    if (!ctx) return "aej: no context (aej_create failed: no such HIP device?)";
    return ctx->err.c_str();
}

extern "C" int aej_synchronize(aej_ctx *ctx)
{
    if (!ctx) return AEJ_ERR_ARG;
    if (call_in_flight(ctx)) return fail(ctx, AEJ_ERR_STATE, "%s between aej_encode_batch_begin and aej_encode_batch_end", __func__);
    AEJ_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- settings ---------------------------------------------------------------------------------------------
// Jpeg._zigzag_ordering (jpeg.py:726-766) as an anti-diagonal walk: even diagonals run bottom-left -> top-right
static void zigzag_order(int s, std::vector<int> &zz)
{
    zz.resize((size_t)s * s);
    int i = 0;
    for (int d = 0; d < 2 * s - 1; d++) {
        int lo = d - s + 1 > 0 ? d - s + 1 : 0, hi = d < s - 1 ? d : s - 1;
        if (d % 2) for (int r = lo; r <= hi; r++) zz[i++] = r * s + (d - r);
        else for (int r = hi; r >= lo; r--) zz[i++] = r * s + (d - r);
    }
}

extern "C" int aej_set_settings(aej_ctx *ctx, int space, int bmin, int bmax, const int32_t *qmats_host)
{
    if (!ctx) return AEJ_ERR_ARG;
    if (call_in_flight(ctx)) return fail(ctx, AEJ_ERR_STATE, "aej_set_settings between aej_encode_batch_begin and aej_encode_batch_end");
    if (space < 0 || space > 6) return fail(ctx, AEJ_ERR_ARG, "Unsupported color space id: %d", space);
    if (!is_pow2(bmin) || !is_pow2(bmax) || bmin > bmax || bmin < 2)
        return fail(ctx, AEJ_ERR_ARG, "block size range (%d, %d): powers of two with 2 <= min <= max required", bmin, bmax);
    if (bmax > kMaxBlock)
        return fail(ctx, AEJ_ERR_UNSUPPORTED, "block size range (%d, %d): no kernel for blocks above %d", bmin, bmax, kMaxBlock);
    if (ilog2(bmax) - ilog2(bmin) + 1 > kMaxSizes)
        return fail(ctx, AEJ_ERR_UNSUPPORTED, "block size range (%d, %d) spans more than %d sizes", bmin, bmax, kMaxSizes);
    if (!qmats_host) return fail(ctx, AEJ_ERR_ARG, "qmats_host is NULL");
    AEJ_HIP_CHECK(hipSetDevice(ctx->device));
    int nsizes = 0;
    for (int s = bmin; s <= bmax; s *= 2) nsizes++;
    // host image of all tables
    std::vector<char> blob;
    auto put = [&](const void *p, size_t bytes) {
        size_t o = (blob.size() + 255) & ~(size_t)255;
        blob.resize(o + bytes);
        memcpy(blob.data() + o, p, bytes);
        return o;
    };
    size_t oD[kMaxSizes], oZ[kMaxSizes], oZf[kMaxSizes], oQ[3][kMaxSizes];
    int k = 0;
    size_t qpos = 0;
    std::vector<size_t> qoff_layer_size;
    for (int s = bmin; s <= bmax; s *= 2, k++) {
        std::vector<float> D((size_t)s * s);
        for (int u = 0; u < s; u++) {
            double alpha = u == 0 ? sqrt(1.0 / (double)s) : sqrt(2.0 / (double)s);
            for (int n = 0; n < s; n++) D[(size_t)u * s + n] = (float)(alpha * cos(3.14159265358979323846 * (double)(2 * n + 1) * (double)u / (2.0 * (double)s)));
        }
        oD[k] = put(D.data(), D.size() * 4);
        std::vector<int> zz, inv((size_t)s * s);
        zigzag_order(s, zz);
        for (int i = 0; i < s * s; i++) inv[zz[i]] = i;
        oZ[k] = put(inv.data(), inv.size() * 4);
        oZf[k] = put(zz.data(), zz.size() * 4);
    }
    for (int l = 0; l < 3; l++) {
        k = 0;
        for (int s = bmin; s <= bmax; s *= 2, k++) {
            for (int i = 0; i < s * s; i++)
                if (qmats_host[qpos + i] < 1) return fail(ctx, AEJ_ERR_ARG, "quantisation matrix entries must be >= 1");
            oQ[l][k] = put(qmats_host + qpos, (size_t)s * s * 4);
            qpos += (size_t)s * s;
        }
    }
    drop_graphs(ctx);                  // captured kernel arguments point into the old tables
    if (ctx->tables) { AEJ_HIP_CHECK(hipStreamSynchronize(ctx->stream)); AEJ_HIP_CHECK(hipFree(ctx->tables)); ctx->tables = nullptr; }
    AEJ_HIP_CHECK(hipMalloc(&ctx->tables, blob.size()));
    AEJ_HIP_CHECK(hipMemcpy(ctx->tables, blob.data(), blob.size(), hipMemcpyHostToDevice));
    char *base = static_cast<char *>(ctx->tables);
    for (int i = 0; i < kMaxSizes; i++) {
        ctx->d_D[i] = nullptr; ctx->d_zzinv[i] = nullptr; ctx->d_zz[i] = nullptr;
        for (int l = 0; l < 3; l++) ctx->d_qm[l][i] = nullptr;
    }
    for (int i = 0; i < nsizes; i++) {
        ctx->d_D[i] = reinterpret_cast<const float *>(base + oD[i]);
        ctx->d_zzinv[i] = reinterpret_cast<const int *>(base + oZ[i]);
        ctx->d_zz[i] = reinterpret_cast<const int *>(base + oZf[i]);
        for (int l = 0; l < 3; l++) ctx->d_qm[l][i] = reinterpret_cast<const int *>(base + oQ[l][i]);
    }
    ctx->space = space; ctx->bmin = bmin; ctx->bmax = bmax; ctx->nsizes = nsizes;
    ctx->has_settings = true;
    return 0;
}

// bilateralFilter(d = 5, sigmaColor, sigmaSpace) weights (edge_detection.py:37-39,78; OpenCV bilateral_filter): (float)exp(double)
// tables built on the host, once per parameter set
static int ensure_canny_tables(aej_ctx *ctx)
{
    if (ctx->d_color_w) return 0;
    float tab[16 + 256] = { 0 };
    double sigc = ctx->canny.bilateral_sigma_color, sigs = ctx->canny.bilateral_sigma_space;
    if (sigc <= 0) sigc = 1;
    if (sigs <= 0) sigs = 1;
    const double cc = -0.5 / (sigc * sigc), sc = -0.5 / (sigs * sigs);
    for (int i = 0; i < 256; i++) tab[16 + i] = (float)exp((double)i * (double)i * cc);
    int t = 0;
    for (int i = -2; i <= 2; i++)
        for (int j = -2; j <= 2; j++) {
            double r = sqrt((double)i * i + (double)j * j);
            if (r > 2.0) continue;
            tab[t++] = (float)exp(r * r * sc);
        }
    if (!ctx->d_bilateral) AEJ_HIP_CHECK(hipMalloc(&ctx->d_bilateral, sizeof tab));
    AEJ_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    AEJ_HIP_CHECK(hipMemcpy(ctx->d_bilateral, tab, sizeof tab, hipMemcpyHostToDevice));
    ctx->d_space_w = ctx->d_bilateral;
    ctx->d_color_w = ctx->d_bilateral + 16;
    return 0;
}

static void apply_canny_params(const aej_ctx *ctx, CannyBuffers &cb)
{
    cb.space_w = ctx->d_space_w;
    cb.color_w = ctx->d_color_w;
    cb.low_q = ctx->canny.canny_low_ratio * 100;       // `canny_low_ratio * 100`, edge_detection.py:81-82
    cb.high_q = ctx->canny.canny_high_ratio * 100;
    cb.clip_limit = ctx->canny.clahe_clip_limit;
    cb.l2 = ctx->canny.use_l2_gradient ? 1 : 0;
}

extern "C" int aej_set_canny_params(aej_ctx *ctx, const aej_canny_params *p)
{
    if (!ctx) return AEJ_ERR_ARG;
    if (call_in_flight(ctx)) return fail(ctx, AEJ_ERR_STATE, "aej_set_canny_params between aej_encode_batch_begin and aej_encode_batch_end");
    const aej_canny_params def = { 0.10, 0.30, 0.75, 75.0, 75.0, 1 };
    const aej_canny_params v = p ? *p : def;
    if (!(v.canny_low_ratio >= 0.0 && v.canny_low_ratio <= 1.0 && v.canny_high_ratio >= 0.0 && v.canny_high_ratio <= 1.0))
        return fail(ctx, AEJ_ERR_ARG, "Canny threshold ratios must lie in [0, 1] (np.percentile takes 0..100)");
    if (!(v.clahe_clip_limit == v.clahe_clip_limit) || !(v.bilateral_sigma_color == v.bilateral_sigma_color) || !(v.bilateral_sigma_space == v.bilateral_sigma_space))
        return fail(ctx, AEJ_ERR_ARG, "NaN Canny hyper-parameter");
    const bool tables_change = v.bilateral_sigma_color != ctx->canny.bilateral_sigma_color || v.bilateral_sigma_space != ctx->canny.bilateral_sigma_space;
    ctx->canny = v;
    drop_graphs(ctx);                      // captured kernel arguments carry the old values
    if (tables_change) { ctx->d_color_w = nullptr; ctx->d_space_w = nullptr; }
    return 0;
}

// ---- Canny chain on a prepared uint8 buffer (cb.u8a) ------------------------------------------------------
// The hysteresis is two launches whatever the image holds: a pass over every tile, then the queue of dirtied tiles drained to the
// fix-point on the device (canny.hip k_hyst_drain) -- no pass count for the host to guess, nothing to read back, nothing to repair.
static int run_canny_chain(aej_ctx *ctx, const Geom &g, CannyWs &w)
{
    hipStream_t st = ctx->stream;
    launch_clahe_pad_hist(st, g, w.cb);
    launch_clahe_lut(st, g, w.cb);
    mark(ctx, AEJ_STAGE_CLAHE_LUT);
    auto publish = [&]() {
        std::lock_guard<std::mutex> lock(g_chain_mutex);
        if (hipEventRecord(ctx->chain_event, st) == hipSuccess) g_last_color_done[ctx->device] = ctx->chain_event;
    };
    launch_clahe_blur(st, g, w.cb);
    mark(ctx, AEJ_STAGE_CLAHE_BLUR);
    if (ctx->chain_hook == 2) publish();
    launch_thresholds(st, g, w.cb);
    mark(ctx, AEJ_STAGE_THRESHOLDS);
    launch_sobel_nms(st, g, w.cb, ctx->tune);
    mark(ctx, AEJ_STAGE_SOBEL_NMS);
    if (ctx->chain_hook == 3) publish();
    launch_hysteresis(st, g, w.cb);
    mark(ctx, AEJ_STAGE_HYSTERESIS);
    AEJ_HIP_CHECK(hipGetLastError());
    return 0;
}

// The zero-fills are kernels while a hipGraph is being captured: the shipped graph holds kernel nodes only (a graph that also held the
// runtime's memset / memcpy nodes faulted on its second replay inside a PyTorch process in round 2; the record of that is
// profiles/r03_graph_memcpy_nodes_fault.txt, the stand-alone replay of the same node types tools/ubench/graph_memcpy_replay.hip).
static int clear_canny_ws(aej_ctx *ctx, const CannyWs &w)
{
    if (ctx->capturing) launch_zero(ctx->stream, w.zero_begin, (size_t)(w.zero_end - w.zero_begin));       // both ends are 256-byte aligned (Carver)
    else AEJ_HIP_CHECK(hipMemsetAsync(w.zero_begin, 0, (size_t)(w.zero_end - w.zero_begin), ctx->stream));
    return 0;
}

static int run_quadtree(aej_ctx *ctx, const Geom &g, const QtGeom &q, QtWs &w, const unsigned long long *edge_bits)
{
    hipStream_t st = ctx->stream;
    if (ctx->capturing) launch_zero(st, w.zero_begin, (size_t)(w.zero_end - w.zero_begin));
    else AEJ_HIP_CHECK(hipMemsetAsync(w.zero_begin, 0, (size_t)(w.zero_end - w.zero_begin), st));
    w.qb.edge_bits = edge_bits;
    launch_qt_cells(st, g, q, edge_bits, w.qb);
    launch_qt_count(st, g, q, w.qb);
    launch_qt_scan(st, g, q, w.qb);
    launch_qt_emit(st, g, q, w.qb);
    AEJ_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- colour planes: fast 4x2-patch kernel when the shape allows it, generic kernel otherwise ---------------------------
// OpenCV resize.cpp computeResizeAreaTab: taps of destination index d are entries off[d]..off[d+1]
static void area_tab(int ssize, int dsize, double scale, std::vector<int> &off, std::vector<int> &si, std::vector<float> &alpha)
{
    off.assign((size_t)dsize + 1, 0); si.clear(); alpha.clear();
    for (int dx = 0; dx < dsize; dx++) {
        double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        if (sx2 > ssize - 1) sx2 = ssize - 1;
        if (sx1 > sx2) sx1 = sx2;
        off[dx] = (int)si.size();
        if (sx1 - fsx1 > 1e-3) { si.push_back(sx1 - 1); alpha.push_back((float)((sx1 - fsx1) / cell)); }
        for (int sx = sx1; sx < sx2; sx++) { si.push_back(sx); alpha.push_back((float)(1.0 / cell)); }
        if (fsx2 - sx2 > 1e-3) {
            double a = fsx2 - sx2;
            if (a > 1.) a = 1.;
            if (a > cell) a = cell;
            si.push_back(sx2); alpha.push_back((float)(a / cell));
        }
    }
    off[dsize] = (int)si.size();
}

static long long area_tab_ints(const Geom &g)      // workspace ints reserved for the tables (upper bound)
{
    return 2LL * (g.w[1] + 1 + g.h[1] + 1) + 2LL * (g.W + 2 * g.w[1] + 2) + 2LL * (g.H + 2 * g.h[1] + 2) + 64;
}

static bool planes_fast_ok(const Geom &g)
{
    bool ok = (g.W % 4) == 0 && (g.H % 2) == 0;
    for (int l = 1; l < 3; l++) ok = ok && g.h[l] * g.rh[l] == g.H && g.w[l] * g.rw[l] == g.W;
    return ok;
}

static int run_color_planes(aej_ctx *ctx, const void *rgb, bool in_u8, const Geom &g, float *raw, float *norm, unsigned char *u8, int *hist,
                            int *tab_ws)
{
    float mid[3], scale[3];
    for (int i = 0; i < 3; i++) { mid[i] = (float)kMid[ctx->space][i]; scale[i] = (float)kScale[ctx->space][i]; }
    if (planes_fast_ok(g)) {
        // the persistent colour streamer beside other parts' kernels (sub-batches, calls in flight): 224 instead of 256 workgroups -- a few CUs
        // without a colour workgroup let the foreground kernels' largest workgroups in sooner (interleaved 3 x: 5.87-5.91 against 5.92-5.97 ms per
        // 64 x 4K step; 192: 5.89-5.91; 160: 5.95-6.06; alone the kernel wants all 256: blocking calls 6.54-6.75 against 6.56-6.68)
        Tuning t = ctx->tune;
        if (t.color_workgroups == 0 && ctx->dct_crowded && ctx->space < 3) t.color_workgroups = 224;
        if (launch_color_planes(ctx->stream, ctx->space, rgb, in_u8, g, mid, scale, raw, norm, u8, hist, t)) return fail(ctx, AEJ_ERR_ARG, "bad colour space");
        return 0;
    }
    AreaTabs t;
    memset(&t, 0, sizeof t);
    // resize(): scale = 1 / (dsize / ssize) in double; the fast integer paths need BOTH scales integral
    double sx = 1.0 / ((double)g.w[1] / (double)g.W), sy = 1.0 / ((double)g.h[1] / (double)g.H);
    int isx = (int)lrint(sx), isy = (int)lrint(sy);
    bool fast = fabs(sx - isx) < 2.220446049250313e-16 && fabs(sy - isy) < 2.220446049250313e-16;
    t.isx = isx; t.isy = isy;
    if (fast) t.mode = (isx == 2 && isy == 2) ? 0 : 1;
    else {
        t.mode = 2;
        if (!tab_ws) return fail(ctx, AEJ_ERR_STATE, "no workspace for the INTER_AREA tables");
        std::vector<int> xoff, xsi, yoff, ysi;
        std::vector<float> xal, yal;
        area_tab(g.W, g.w[1], sx, xoff, xsi, xal);
        area_tab(g.H, g.h[1], sy, yoff, ysi, yal);
        std::vector<int> blob;
        auto put_i = [&](const std::vector<int> &v) { size_t o = blob.size(); blob.insert(blob.end(), v.begin(), v.end()); return o; };
        auto put_f = [&](const std::vector<float> &v) { size_t o = blob.size(); blob.resize(o + v.size()); memcpy(blob.data() + o, v.data(), v.size() * 4); return o; };
        size_t o1 = put_i(xoff), o2 = put_i(xsi), o3 = put_f(xal), o4 = put_i(yoff), o5 = put_i(ysi), o6 = put_f(yal);
        if ((long long)blob.size() > area_tab_ints(g)) return fail(ctx, AEJ_ERR_CAPACITY, "INTER_AREA tables larger than reserved");
        AEJ_HIP_CHECK(hipMemcpyAsync(tab_ws, blob.data(), blob.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        AEJ_HIP_CHECK(hipStreamSynchronize(ctx->stream));     // blob is a host temporary
        t.xoff = tab_ws + o1; t.xsi = tab_ws + o2; t.xal = reinterpret_cast<const float *>(tab_ws + o3);
        t.yoff = tab_ws + o4; t.ysi = tab_ws + o5; t.yal = reinterpret_cast<const float *>(tab_ws + o6);
    }
    if (launch_color_planes_generic(ctx->stream, ctx->space, rgb, in_u8, g, mid, scale, t, raw, norm, u8, hist)) return fail(ctx, AEJ_ERR_ARG, "bad colour space");
    return 0;
}

// ---- whole path ---------------------------------------------------------------------------------------------
static long long big_scratch_floats(int bmax)      // the launches of different sizes run one after the other: one scratch, sized for the largest
{
    long long m = 0;
    for (int s = 256; s <= bmax; s *= 2) m = std::max(m, big_scratch_floats_for(s));
    return m;
}

struct EncodeWs {
    float *big;              // scratch of the 256 x 256 DCT kernel (null unless the settings allow that size)
    float *norm;
    int *area_tabs;
    CannyWs canny;
    QtWs qt;
    unsigned long long bytes;
};

static void carve_encode(void *base, const Geom &g, const QtGeom &q, EncodeWs &w)
{
    Carver c(base);
    w.norm = c.take<float>((long long)g.B * g.pstride);
    w.area_tabs = c.take<int>(area_tab_ints(g));
    carve_canny(c, g, w.canny);
    carve_qt(c, g, q, true, w.qt);
    w.big = big_scratch_floats(q.bmax) ? c.take<float>(big_scratch_floats(q.bmax)) : nullptr;
    w.bytes = (c.off + 255) & ~255ull;
}

static unsigned long long sub_ws_bytes(Geom g, const QtGeom &q, int nsub);      // sub-batch pipelining, below

static int check_encode_args(aej_ctx *ctx, int batch, int H, int W)
{
    if (!ctx) return AEJ_ERR_ARG;
    if (!ctx->has_settings) return fail(ctx, AEJ_ERR_STATE, "aej_set_settings has not been called");
    if (batch < 1 || H < 1 || W < 1) return fail(ctx, AEJ_ERR_ARG, "batch, H, W must be positive");
    if (batch * 3 > kMaxPlanes) return fail(ctx, AEJ_ERR_UNSUPPORTED, "batch %d too large for one call (max %d images)", batch, kMaxPlanes / 3);
    return 0;
}

extern "C" int aej_encode_plan(aej_ctx *ctx, int batch, int H, int W, aej_plan *plan)
{
    int rc = check_encode_args(ctx, batch, H, W);
    if (rc) return rc;
    if (!plan) return fail(ctx, AEJ_ERR_ARG, "plan is NULL");
    Geom g;
    QtGeom q;
    if ((rc = make_geom(ctx, ctx->space, batch, H, W, g))) return rc;
    if ((rc = make_qtgeom(ctx, g, ctx->bmin, ctx->bmax, q))) return rc;
    EncodeWs w;
    carve_encode(nullptr, g, q, w);
    memset(plan, 0, sizeof *plan);
    plan->batch = batch; plan->H = H; plan->W = W;
    for (int l = 0; l < 3; l++) {
        plan->layer_h[l] = g.h[l]; plan->layer_w[l] = g.w[l]; plan->root_size[l] = q.root[l];
        plan->coeff_off[l] = q.coeff_off[l]; plan->leaf_off[l] = q.leaf_off[l]; plan->state_off[l] = q.state_off[l];
    }
    plan->coeff_stride = q.coeff_stride; plan->leaf_stride = q.leaf_stride; plan->state_stride = q.state_stride;
    plan->workspace_bytes = w.bytes;
    // a call that is cut into sub-batches uses one slice per sub-batch (their fixed parts make the sum slightly larger); with the
    // automatic mode the decision can change with later settings, so the plan covers every split the context could choose
    for (int n = 2; n <= aej_ctx::kMaxSub && n <= batch; n++)
        plan->workspace_bytes = std::max<uint64_t>(plan->workspace_bytes, sub_ws_bytes(g, q, n) * (unsigned long long)n);
    return 0;
}

// everything behind the hysteresis: quadtree, then one DCT launch per block size
constexpr long long kGraphAutoPixels = 8LL << 20;      // latency-sized calls: at most 8 Mpx (automatic graph mode, the one-launch DCT)

static int enqueue_back(aej_ctx *ctx, const Geom &g, const QtGeom &q, EncodeWs &w, int32_t *coeffs, float *dct_f32)
{
    hipStream_t st = ctx->stream;
    int rc;
    if ((rc = run_quadtree(ctx, g, q, w.qt, w.canny.cb.strong))) return rc;
    mark(ctx, AEJ_STAGE_QUADTREE);
    DctArgs args[kMaxSizes];
    int k = 0;
    for (int s = q.bmin; s <= q.bmax; s *= 2, k++) {
        DctArgs &a = args[k];
        a.norm = w.norm; a.coeffs = coeffs; a.dct_f32 = dct_f32;
        a.work = w.qt.qb.work[k]; a.work_count = w.qt.qb.work_count; a.k = k; a.nplanes = g.B * 3;
        a.scratch = w.big;
        a.D = ctx->d_D[k]; a.zzinv = ctx->d_zzinv[k];
        a.crowded = ctx->dct_crowded;
        for (int l = 0; l < 3; l++) a.qm[l] = ctx->d_qm[l][k];
    }
    // latency-sized, unprofiled calls: every size in one launch (per-size stage times need per-size launches)
    const bool one_launch = ctx->tune.dct_multi && !ctx->profiling && (long long)g.B * g.H * g.W <= kGraphAutoPixels;
    if (!(one_launch && launch_dct_multi(st, g, q, args, w.qt.qb.work_cap) == 0)) {
        k = 0;
        for (int s = q.bmin; s <= q.bmax; s *= 2, k++) {
            if (launch_dct(st, s, g, q, args[k], w.qt.qb.work_cap[k], ctx->tune)) return fail(ctx, AEJ_ERR_UNSUPPORTED, "no DCT kernel for block size %d with %d planes", s, args[k].nplanes);
            mark(ctx, AEJ_STAGE_DCT_2 + ilog2(s) - 1);
        }
    }
    AEJ_HIP_CHECK(hipGetLastError());
    return 0;
}

// the counters one call reads back: the quadtree's overflow flag and (a diagnostic) how many tiles went through the hysteresis queue
static int enqueue_readback(aej_ctx *ctx, EncodeWs &w)
{
    AEJ_HIP_CHECK(hipMemcpyAsync(ctx->h_flag, w.qt.qb.overflow, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    AEJ_HIP_CHECK(hipMemcpyAsync(ctx->h_flag + 1, w.canny.cb.pass_count + 32 /* kQTail */, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    return 0;
}

constexpr size_t kMaxGraphs = 8;

// Launch-latency path: the whole sequence (about 20 launches for 4-64 blocks) as ONE hipGraphLaunch.  The graph is captured
// on a private stream (the caller's may be the legacy null stream, which cannot be captured) ordered behind the caller's stream
// by an event, and cached under every pointer / shape its kernel arguments contain.
static int encode_graph(aej_ctx *ctx, const void *rgb, bool in_u8, const Geom &g, const QtGeom &q, EncodeWs &w, int32_t *coeffs, int32_t *leaves,
                        uint8_t *states, int64_t *counts, float *dct_f32, void *workspace, bool &used)
{
    used = false;
    if (!ctx->gstream) {
        AEJ_HIP_CHECK(hipStreamCreateWithFlags(&ctx->gstream, hipStreamNonBlocking));
        AEJ_HIP_CHECK(hipEventCreateWithFlags(&ctx->gevent, hipEventDisableTiming));
    }
    aej_ctx::GraphEntry *hit = nullptr;
    for (auto &e : ctx->graphs)
        if (e.rgb == rgb && e.coeffs == coeffs && e.leaves == leaves && e.states == states && e.counts == counts && e.dct == dct_f32 && e.ws == workspace &&
            e.batch == g.B && e.H == g.H && e.W == g.W && e.in_u8 == (int)in_u8) { hit = &e; break; }
    hipStream_t user = ctx->stream;
    if (!hit) {
        // first sight of this combination of buffers: only remember it and let the caller run the ordinary path -- a caller that
        // allocates fresh outputs for every call would otherwise pay a capture per call; the second sight captures
        if (ctx->graphs.size() >= kMaxGraphs) {           // evict the least recently used
            size_t lru = 0;
            for (size_t i = 1; i < ctx->graphs.size(); i++) if (ctx->graphs[i].last_use < ctx->graphs[lru].last_use) lru = i;
            if (ctx->graphs[lru].exec) (void)hipGraphExecDestroy(ctx->graphs[lru].exec);
            ctx->graphs.erase(ctx->graphs.begin() + (long)lru);
        }
        ctx->graphs.push_back({ rgb, coeffs, leaves, states, counts, dct_f32, workspace, g.B, g.H, g.W, (int)in_u8, nullptr, ++ctx->graph_clock });
        return 0;
    }
    if (!hit->exec) {
        hipGraph_t graph = nullptr;
        ctx->stream = ctx->gstream;                       // every enqueue below goes to the capturing stream
        ctx->capturing = true;
        hipError_t e = hipStreamBeginCapture(ctx->gstream, hipStreamCaptureModeThreadLocal);
        int rc = e == hipSuccess ? 0 : hip_fail(ctx, e, "hipStreamBeginCapture", __FILE__, __LINE__);
        if (!rc) rc = clear_canny_ws(ctx, w.canny);
        if (!rc) rc = run_color_planes(ctx, rgb, in_u8, g, nullptr, w.norm, w.canny.cb.u8a, w.canny.cb.tile_hist, w.area_tabs);
        if (!rc) {
            hipStream_t st = ctx->stream;
            launch_clahe_pad_hist(st, g, w.canny.cb);
            launch_clahe_lut(st, g, w.canny.cb);
            launch_clahe_blur(st, g, w.canny.cb);
            launch_thresholds(st, g, w.canny.cb);
            launch_sobel_nms(st, g, w.canny.cb, ctx->tune);
            launch_hysteresis(st, g, w.canny.cb);
            rc = enqueue_back(ctx, g, q, w, coeffs, dct_f32);
        }
        hipError_t e2 = e == hipSuccess ? hipStreamEndCapture(ctx->gstream, &graph) : hipSuccess;
        ctx->stream = user;
        ctx->capturing = false;
        if (rc || e2 != hipSuccess || !graph) {
            if (graph) (void)hipGraphDestroy(graph);
            (void)hipGetLastError();
            return rc ? rc : 0;                           // not captured: the caller runs the ordinary path
        }
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess || !exec) { (void)hipGetLastError(); return 0; }
        hit->exec = exec;
        ctx->n_graph_captures++;
    }
    hit->last_use = ++ctx->graph_clock;
    AEJ_HIP_CHECK(hipEventRecord(ctx->gevent, user));     // inputs produced on the caller's stream are complete before the graph reads them
    AEJ_HIP_CHECK(hipStreamWaitEvent(ctx->gstream, ctx->gevent, 0));
    AEJ_HIP_CHECK(hipGraphLaunch(hit->exec, ctx->gstream));
    // the counter read-back stays outside the graph (ordinary copies behind it on the same stream): kernel nodes only, see clear_canny_ws
    ctx->stream = ctx->gstream;
    const int rb = enqueue_readback(ctx, w);
    ctx->stream = user;
    if (rb) return rb;
    AEJ_HIP_CHECK(hipStreamSynchronize(ctx->gstream));
    ctx->n_graph_launches++;
    used = true;
    return 0;
}

// ---- sub-batch pipelining ------------------------------------------------------------------------------------
// How many sub-batches a call is cut into (automatic mode: by call size and by how many hardware queues the process has, below;
// smaller calls have too few workgroups per kernel to share the chip).  Never for profiled calls (the
// stage timings describe the serial chain), graph replay, or shapes that need the host-built INTER_AREA tables.
static int sub_batches(const aej_ctx *ctx, const Geom &g, int hw_queues, bool as_if_unprofiled = false)
{
    if (ctx->sub_mode == 1 || (ctx->profiling && !as_if_unprofiled) || ctx->graph_mode == 2 || !planes_fast_ok(g)) return 1;
    int n = ctx->sub_mode;
    if (n == 0) {
        const long long px = (long long)g.B * g.H * g.W;
        if (hw_queues >= 8) {
            // every stream has a hardware queue of its own: four chains for a 64 x 4K call, two for a 64 x 1080p or 8 x 8K one, also
            // beside a call in flight on another context (64 x 4K, two contexts: 7.45 ms with 4 sub-batches each, 7.5 with 2, 7.75 with
            // none, 8.2 with 8; 64 x 1080p: 2.11 ms with 2, 2.26 with 4)
            n = (px >= (384LL << 20) && g.B >= 16) ? 4 : (px >= (64LL << 20) && g.B >= 8) ? 2 : 1;
        } else {
            // HIP's default of 4 hardware queues: streams start to share queues (two streams on one queue run one after the other), so
            // two sub-batches, and only for a call that has the device to itself (with 4 queues: 4 sub-batches 8.4 ms, 2: 8.1 ms)
            bool alone;
            { std::lock_guard<std::mutex> lock(g_chain_mutex); alone = g_calls_in_flight[ctx->device] == 0; }
            n = (alone && px >= (64LL << 20) && g.B >= 8) ? 2 : 1;
        }
    }
    if (n > aej_ctx::kMaxSub) n = aej_ctx::kMaxSub;
    if (n > g.B) n = g.B;
    return n;
}

static unsigned long long encode_ws_bytes(const Geom &g, const QtGeom &q)
{
    EncodeWs w;
    carve_encode(nullptr, g, q, w);
    return w.bytes;
}

// bytes of the workspace slice of one sub-batch (sized for the largest of them)
static unsigned long long sub_ws_bytes(Geom g, const QtGeom &q, int nsub)
{
    g.B = (g.B + nsub - 1) / nsub;
    return encode_ws_bytes(g, q);
}

// One call in flight: everything aej_encode_batch_end needs to complete and check what aej_encode_batch_begin enqueued.  Part 0 is the
// whole batch on the context's stream, or parts 0..n-1 are the sub-batches.
struct EncodePart { Geom g; EncodeWs w; bool whole_call = false; int32_t *coeffs = nullptr; float *dct = nullptr; hipStream_t stream = nullptr; int *flag = nullptr; bool used = false; };
struct aej_pending {
    bool active = false, complete = false;     // complete: already synchronised and verified (graph replay)
    QtGeom q;
    std::vector<EncodePart> parts;
};

static void free_pending(aej_ctx *ctx) { delete ctx->pending; ctx->pending = nullptr; }

static bool call_in_flight(const aej_ctx *ctx) { return ctx->pending && ctx->pending->active; }

static aej_pending &pending_of(aej_ctx *ctx)
{
    if (!ctx->pending) ctx->pending = new aej_pending;
    return *ctx->pending;
}

// the launch sequence of one part on ctx->stream / ctx->h_flag (both set by the caller): clear, colour planes, Canny chain, quadtree,
// DCT, counter read-back.  `after` / `done`: sub-batch staggering (null for the unsplit call).
static int enqueue_part(aej_ctx *ctx, EncodePart &p, const QtGeom &q, const void *rgb, bool in_u8, hipEvent_t done)
{
    int rc;
    mark(ctx, -1);
    if ((rc = clear_canny_ws(ctx, p.w.canny))) return rc;
    mark(ctx, AEJ_STAGE_CLEAR);
    // one stage behind the part enqueued before this one (g_last_color_done): its colour stage (HBM-bound) has finished, its blur
    // (issue-bound) is starting
    const bool chain = ctx->sub_chain && done && !ctx->profiling;
    const int chain_mode = ctx->sub_chain > 0 ? ctx->sub_chain : 1;
    if (chain) {
        std::lock_guard<std::mutex> lock(g_chain_mutex);
        if (hipEvent_t after = g_last_color_done[ctx->device]) AEJ_HIP_CHECK(hipStreamWaitEvent(ctx->stream, after, 0));
    }
    if ((rc = run_color_planes(ctx, rgb, in_u8, p.g, nullptr, p.w.norm, p.w.canny.cb.u8a, p.w.canny.cb.tile_hist, p.w.area_tabs))) return rc;
    mark(ctx, AEJ_STAGE_COLOR_PLANES);
    if ((rc = injected_failure(ctx, AEJ_STAGE_COLOR_PLANES))) return rc;
    auto publish = [&]() -> int {
        std::lock_guard<std::mutex> lock(g_chain_mutex);
        AEJ_HIP_CHECK(hipEventRecord(done, ctx->stream));
        g_last_color_done[ctx->device] = done;
        return 0;
    };
    if (chain && chain_mode == 1 && (rc = publish())) return rc;
    // (the hook is cleared on every exit: a later stand-alone aej_canny on this context must not re-record the shared chain event)
    struct HookGuard { aej_ctx *c; ~HookGuard() { c->chain_hook = 0; } } hook_guard{ ctx };
    ctx->chain_hook = (chain && chain_mode > 1) ? chain_mode : 0;
    ctx->chain_event = done;
    if ((rc = run_canny_chain(ctx, p.g, p.w.canny))) return rc;
    ctx->chain_hook = 0;
    if ((rc = injected_failure(ctx, AEJ_STAGE_HYSTERESIS))) return rc;
    if ((rc = enqueue_back(ctx, p.g, q, p.w, p.coeffs, p.dct))) return rc;
    if ((rc = injected_failure(ctx, AEJ_STAGE_DCT_64))) return rc;
    return enqueue_readback(ctx, p.w);      // one read-back for the whole part
}

static int encode_begin_impl(aej_ctx *ctx, const void *rgb, bool in_u8, int batch, int H, int W, int32_t *coeffs, int32_t *leaves,
                             uint8_t *states, int64_t *counts, float *dct_f32, void *workspace, uint64_t workspace_bytes, bool &started)
{
    started = false;            // true once this call has put something in flight (then aej_encode_batch_end has to follow, also after an error)
    int rc = check_encode_args(ctx, batch, H, W);
    if (rc) return rc;
    if (!rgb || !coeffs || !leaves || !states || !counts || !workspace) return fail(ctx, AEJ_ERR_ARG, "NULL buffer");
    aej_pending &pd = pending_of(ctx);
    if (pd.active) return fail(ctx, AEJ_ERR_STATE, "aej_encode_batch_begin: the previous call has not been ended (aej_encode_batch_end)");
    AEJ_HIP_CHECK(hipSetDevice(ctx->device));
    Geom g;
    if ((rc = make_geom(ctx, ctx->space, batch, H, W, g))) return rc;
    if ((rc = make_qtgeom(ctx, g, ctx->bmin, ctx->bmax, pd.q))) return rc;
    const QtGeom &q = pd.q;
    if ((rc = ensure_canny_tables(ctx))) return rc;
    ctx->n_ev = 0;
    ctx->n_encode_calls++;
    const int nsub = sub_batches(ctx, g, ctx->hw_queues);
    {
        std::lock_guard<std::mutex> lock(g_chain_mutex);
        ctx->dct_crowded = nsub > 1 || g_calls_in_flight[ctx->device] > 0;
    }
    // (a profiled call runs unsplit so that its stage times describe the serial chain, but with the kernels the same call uses unprofiled)
    if (ctx->profiling && sub_batches(ctx, g, ctx->hw_queues, true) > 1) ctx->dct_crowded = 1;
    pd.parts.assign((size_t)nsub, EncodePart());
    pd.complete = false;
    hipStream_t user = ctx->stream;
    int *user_flag = ctx->h_flag;

    if (nsub == 1) {
        EncodePart &p = pd.parts[0];
        p.g = g; p.coeffs = coeffs; p.dct = dct_f32; p.stream = user; p.flag = user_flag; p.used = true; p.whole_call = true;
        p.g.tiled = planes_fast_ok(p.g) && color_planes_can_tile(p.g, ctx->space, in_u8, ctx->tune);
        carve_encode(workspace, g, q, p.w);
        if (p.w.bytes > workspace_bytes) return fail(ctx, AEJ_ERR_CAPACITY, "workspace too small: need %llu bytes, got %llu", p.w.bytes, (unsigned long long)workspace_bytes);
        apply_canny_params(ctx, p.w.canny.cb);
        p.w.qt.qb.leaves = leaves;
        p.w.qt.qb.states = states;
        p.w.qt.qb.counts = reinterpret_cast<long long *>(counts);
        bool graphed = false;
        const bool want_graph = ctx->graph_mode != 0 && !ctx->profiling && planes_fast_ok(g) &&
                                (ctx->graph_mode == 2 || (long long)batch * H * W <= kGraphAutoPixels);
        // (p.g, not g: capture, replay and a miss repair in encode_end_impl must share one plane layout -- Geom::tiled)
        if (want_graph && (rc = encode_graph(ctx, rgb, in_u8, p.g, q, p.w, coeffs, leaves, states, counts, dct_f32, workspace, graphed))) {
            if (ctx->gstream) (void)hipStreamSynchronize(ctx->gstream);       // a replay whose read-back failed may still be running
            return rc;
        }
        if (graphed) pd.complete = true;      // the replay path has synchronised its own stream
        else {
            if (!ctx->sub_color_done[0]) AEJ_HIP_CHECK(hipEventCreateWithFlags(&ctx->sub_color_done[0], hipEventDisableTiming));
            rc = enqueue_part(ctx, p, q, rgb, in_u8, hyst_tiles_per_image(g) * g.B <= 4096 ? nullptr : ctx->sub_color_done[0]);      // (latency-sized calls stay out of the chain)
        }
        // also after an error: whatever enqueue_part had already put on the stream is drained by the caller (encode_end_impl), exactly as
        // on the sub-batch path below
        pd.active = started = true;
        { std::lock_guard<std::mutex> lock(g_chain_mutex); g_calls_in_flight[ctx->device]++; }
        return rc;
    }

    // ---- sub-batches on private streams
    const unsigned long long slice = sub_ws_bytes(g, q, nsub);
    if (slice * (unsigned long long)nsub > workspace_bytes)
        return fail(ctx, AEJ_ERR_CAPACITY, "workspace too small for %d sub-batches: need %llu bytes, got %llu", nsub, slice * (unsigned long long)nsub,
                    (unsigned long long)workspace_bytes);
    for (int i = 0; i < nsub; i++) {
        if (!ctx->sub_color_done[i]) AEJ_HIP_CHECK(hipEventCreateWithFlags(&ctx->sub_color_done[i], hipEventDisableTiming));
        if (!ctx->sub_stream[i]) {
            AEJ_HIP_CHECK(hipStreamCreateWithFlags(&ctx->sub_stream[i], hipStreamNonBlocking));
            AEJ_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&ctx->sub_flag[i]), kFlagWords * sizeof(int), hipHostMallocDefault));
        }
    }
    if (!ctx->sub_in) AEJ_HIP_CHECK(hipEventCreateWithFlags(&ctx->sub_in, hipEventDisableTiming));
    AEJ_HIP_CHECK(hipEventRecord(ctx->sub_in, user));          // inputs produced on the caller's stream are complete before any sub-batch reads them
    ctx->n_split_calls++;
    const size_t px_bytes = in_u8 ? 1 : sizeof(float);
    for (int i = 0; i < nsub && !rc; i++) {
        EncodePart &p = pd.parts[(size_t)i];
        const int b0 = (int)((long long)g.B * i / nsub), b1 = (int)((long long)g.B * (i + 1) / nsub);
        p.g = g;
        p.g.B = b1 - b0;
        p.g.tiled = planes_fast_ok(p.g) && color_planes_can_tile(p.g, ctx->space, in_u8, ctx->tune);      // (decided per part: the strip height depends on the part's batch)
        carve_encode(static_cast<char *>(workspace) + (size_t)i * slice, p.g, q, p.w);
        apply_canny_params(ctx, p.w.canny.cb);
        p.w.qt.qb.leaves = leaves + (long long)b0 * q.leaf_stride * 4;
        p.w.qt.qb.states = states + (long long)b0 * q.state_stride;
        p.w.qt.qb.counts = reinterpret_cast<long long *>(counts) + (long long)b0 * 12;
        p.coeffs = coeffs + (long long)b0 * q.coeff_stride;
        p.dct = dct_f32 ? dct_f32 + (long long)b0 * q.coeff_stride : nullptr;
        p.stream = ctx->sub_stream[i];
        p.flag = ctx->sub_flag[i];
        p.used = true;
        const void *in = static_cast<const char *>(rgb) + (size_t)b0 * g.H * g.W * 3 * px_bytes;
        ctx->stream = p.stream;
        ctx->h_flag = p.flag;
        hipError_t e = hipStreamWaitEvent(ctx->stream, ctx->sub_in, 0);
        if (e != hipSuccess) rc = hip_fail(ctx, e, "hipStreamWaitEvent", __FILE__, __LINE__);
        else rc = enqueue_part(ctx, p, q, in, in_u8, ctx->sub_color_done[i]);
    }
    ctx->stream = user;
    ctx->h_flag = user_flag;
    pd.active = started = true;  // also after an error: the caller drains whatever was enqueued
    { std::lock_guard<std::mutex> lock(g_chain_mutex); g_calls_in_flight[ctx->device]++; }
    return rc;
}

// completion of the call in flight: every stream it used is drained (also after an error: nothing may still be running when the
// caller sees the result) and the device-side counters are checked
static int encode_end_impl(aej_ctx *ctx, int rc_begin)
{
    if (!ctx) return AEJ_ERR_ARG;
    aej_pending &pd = pending_of(ctx);
    if (!pd.active) return rc_begin ? rc_begin : fail(ctx, AEJ_ERR_STATE, "aej_encode_batch_end without a call in flight");
    pd.active = false;
    { std::lock_guard<std::mutex> lock(g_chain_mutex); if (g_calls_in_flight[ctx->device] > 0) g_calls_in_flight[ctx->device]--; }
    (void)hipSetDevice(ctx->device);
    int rc = rc_begin;
    long long queued = 0;
    for (EncodePart &p : pd.parts) {
        if (!p.used) continue;                    // not reached by a failed begin
        hipError_t e = pd.complete ? hipSuccess : hipStreamSynchronize(p.stream);
        if (e != hipSuccess && !rc) rc = hip_fail(ctx, e, "hipStreamSynchronize", __FILE__, __LINE__);
        if (rc) continue;
        if (p.flag[0]) { rc = fail(ctx, AEJ_ERR_CAPACITY, "internal capacity exceeded in the quadtree emit pass"); continue; }
        // (bit 30 of the queue's tail counter: a wave of the hysteresis work queue waited longer than any correct run can make it -- canny.hip kQPoison)
        if (p.flag[1] & 0x40000000) { rc = fail(ctx, AEJ_ERR_STATE, "the hysteresis work queue did not drain (internal error): the edge maps of this call are not trustworthy"); continue; }
        queued += p.flag[1];
    }
    if (rc) return rc;
    ctx->last_hyst_queued = queued;
    if (ctx->profiling) collect_marks(ctx);
    return 0;
}

static int encode_batch_impl(aej_ctx *ctx, const void *rgb, bool in_u8, int batch, int H, int W, int32_t *coeffs, int32_t *leaves,
                             uint8_t *states, int64_t *counts, float *dct_f32, void *workspace, uint64_t workspace_bytes)
{
    bool started;
    const int rc = encode_begin_impl(ctx, rgb, in_u8, batch, H, W, coeffs, leaves, states, counts, dct_f32, workspace, workspace_bytes, started);
    return started ? encode_end_impl(ctx, rc) : rc;
}

extern "C" int aej_encode_batch_begin(aej_ctx *ctx, const void *rgb, int rgb_is_u8, int batch, int H, int W, int32_t *coeffs, int32_t *leaves,
                                      uint8_t *states, int64_t *counts, float *dct_f32, void *workspace, uint64_t workspace_bytes)
{
    bool started;
    const int rc = encode_begin_impl(ctx, rgb, rgb_is_u8 != 0, batch, H, W, coeffs, leaves, states, counts, dct_f32, workspace, workspace_bytes, started);
    return rc && started ? encode_end_impl(ctx, rc) : rc;      // a failed begin leaves nothing of its own in flight
}

extern "C" int aej_encode_batch_end(aej_ctx *ctx) { return encode_end_impl(ctx, 0); }

extern "C" int aej_encode_batch(aej_ctx *ctx, const float *rgb, int batch, int H, int W, int32_t *coeffs, int32_t *leaves,
                                uint8_t *states, int64_t *counts, float *dct_f32, void *workspace, uint64_t workspace_bytes)
{
    return encode_batch_impl(ctx, rgb, false, batch, H, W, coeffs, leaves, states, counts, dct_f32, workspace, workspace_bytes);
}

extern "C" int aej_encode_batch_u8(aej_ctx *ctx, const uint8_t *rgb_u8, int batch, int H, int W, int32_t *coeffs, int32_t *leaves,
                                   uint8_t *states, int64_t *counts, float *dct_f32, void *workspace, uint64_t workspace_bytes)
{
    return encode_batch_impl(ctx, rgb_u8, true, batch, H, W, coeffs, leaves, states, counts, dct_f32, workspace, workspace_bytes);
}

// ---- stage entry points ---------------------------------------------------------------------------------------
extern "C" int aej_color_convert(aej_ctx *ctx, int space, const float *rgb, float *out, int64_t n)
{
    if (!ctx) return AEJ_ERR_ARG;
    if (call_in_flight(ctx)) return fail(ctx, AEJ_ERR_STATE, "%s between aej_encode_batch_begin and aej_encode_batch_end", __func__);
    if (n < 0 || (n > 0 && (!rgb || !out))) return fail(ctx, AEJ_ERR_ARG, "bad buffer");
    if (n == 0) return 0;
    AEJ_HIP_CHECK(hipSetDevice(ctx->device));
    if (launch_color_convert(ctx->stream, space, rgb, out, n)) return fail(ctx, AEJ_ERR_ARG, "Invalid color space id %d", space);
    AEJ_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int aej_color_planes(aej_ctx *ctx, const float *rgb, int batch, int H, int W, float *planes_raw, float *planes_norm,
                                uint8_t *planes_u8)
{
    int rc = check_encode_args(ctx, batch, H, W);
    if (rc) return rc;
    if (call_in_flight(ctx)) return fail(ctx, AEJ_ERR_STATE, "%s between aej_encode_batch_begin and aej_encode_batch_end", __func__);
    AEJ_HIP_CHECK(hipSetDevice(ctx->device));
    Geom g;
    if ((rc = make_geom(ctx, ctx->space, batch, H, W, g))) return rc;
    int *tabs = nullptr;
    if (!planes_fast_ok(g)) AEJ_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&tabs), (size_t)area_tab_ints(g) * 4));   // stage entry only
    rc = run_color_planes(ctx, rgb, false, g, planes_raw, planes_norm, planes_u8, nullptr, tabs);
    if (tabs) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(tabs); }
    if (rc) return rc;
    AEJ_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" uint64_t aej_canny_workspace_bytes(int H, int W)
{
    if (H < 1 || W < 1) return 0;
    Geom g;
    make_plane_geom(H, W, g);
    Carver c(nullptr);
    CannyWs w;
    carve_canny(c, g, w);
    return (c.off + 255) & ~255ull;
}

extern "C" int aej_canny(aej_ctx *ctx, const float *plane, int H, int W, uint8_t *edge, uint8_t *stages, int32_t *thresholds,
                         void *workspace, uint64_t workspace_bytes)
{
    if (!ctx) return AEJ_ERR_ARG;
    if (call_in_flight(ctx)) return fail(ctx, AEJ_ERR_STATE, "%s between aej_encode_batch_begin and aej_encode_batch_end", __func__);
    if (H < 1 || W < 1) return fail(ctx, AEJ_ERR_ARG, "Input array must be a 2D.");
    if (!plane || !edge || !workspace) return fail(ctx, AEJ_ERR_ARG, "NULL buffer");
    if (workspace_bytes < aej_canny_workspace_bytes(H, W)) return fail(ctx, AEJ_ERR_CAPACITY, "workspace too small");
    AEJ_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = ensure_canny_tables(ctx);
    if (rc) return rc;
    Geom g;
    make_plane_geom(H, W, g);
    Carver c(workspace);
    CannyWs w;
    carve_canny(c, g, w);
    apply_canny_params(ctx, w.cb);
    long long n = (long long)H * W;
    if (stages) { w.cb.dump_clahe = stages + n; w.cb.dump_gauss = stages + 2 * n; }
    if ((rc = clear_canny_ws(ctx, w))) return rc;
    launch_plane_u8(ctx->stream, plane, g, w.cb.u8a, w.cb.tile_hist);
    // stage dumps are H*W bytes each; the plane buffers are padded to 64, so copy exactly n bytes
    hipStream_t st = ctx->stream;
    if (stages) AEJ_HIP_CHECK(hipMemcpyAsync(stages, w.cb.u8a, n, hipMemcpyDeviceToDevice, st));
    launch_clahe_pad_hist(st, g, w.cb);
    launch_clahe_lut(st, g, w.cb);
    launch_clahe_blur(st, g, w.cb);
    if (stages) AEJ_HIP_CHECK(hipMemcpyAsync(stages + 3 * n, w.cb.u8b, n, hipMemcpyDeviceToDevice, st));
    launch_thresholds(st, g, w.cb);
    if (thresholds) AEJ_HIP_CHECK(hipMemcpyAsync(thresholds, w.cb.thr, 2 * sizeof(int), hipMemcpyDeviceToDevice, st));
    launch_sobel_nms(st, g, w.cb, ctx->tune);
    Geom ge = g;
    ge.pstride = n;   // uint8 outputs are exactly H*W
    if (stages) launch_bits_to_map(st, ge, w.cb.weak, w.cb.strong, stages + 4 * n);
    launch_hysteresis(st, g, w.cb);
    launch_bits_to_edge(st, ge, w.cb.strong, edge);
    AEJ_HIP_CHECK(hipGetLastError());
    AEJ_HIP_CHECK(hipStreamSynchronize(st));
    return 0;
}

extern "C" int aej_quadtree_capacity(int H, int W, int min_size, int max_size, int64_t *leaf_cap, int64_t *state_cap, int64_t *coeff_cap)
{
    if (H < 1 || W < 1) return AEJ_ERR_ARG;
    Geom g;
    make_plane_geom(H, W, g);
    QtGeom q;
    int rc = make_qtgeom(nullptr, g, min_size, max_size, q, true);
    if (rc) return rc;
    if (leaf_cap) *leaf_cap = q.leaf_cap[0];
    if (state_cap) *state_cap = q.state_cap[0];
    if (coeff_cap) *coeff_cap = q.coeff_cap[0];
    return 0;
}

extern "C" uint64_t aej_quadtree_workspace_bytes(int H, int W, int min_size, int max_size)
{
    if (H < 1 || W < 1) return 0;
    Geom g;
    make_plane_geom(H, W, g);
    QtGeom q;
    if (make_qtgeom(nullptr, g, min_size, max_size, q, true)) return 0;
    Carver c(nullptr);
    QtWs w;
    carve_qt(c, g, q, false, w);
    c.take<unsigned long long>(g.bpstride);
    return (c.off + 255) & ~255ull;
}

extern "C" int aej_quadtree(aej_ctx *ctx, const uint8_t *edge, int H, int W, int min_size, int max_size, int32_t *leaves,
                            uint8_t *states, int64_t *counts, void *workspace, uint64_t workspace_bytes)
{
    if (!ctx) return AEJ_ERR_ARG;
    if (call_in_flight(ctx)) return fail(ctx, AEJ_ERR_STATE, "%s between aej_encode_batch_begin and aej_encode_batch_end", __func__);
    if (H < 1 || W < 1) return fail(ctx, AEJ_ERR_ARG, "Input array must be a 2D with a single channel.");
    if (!edge || !leaves || !states || !counts || !workspace) return fail(ctx, AEJ_ERR_ARG, "NULL buffer");
    AEJ_HIP_CHECK(hipSetDevice(ctx->device));
    Geom g;
    make_plane_geom(H, W, g);
    g.pstride = (long long)H * W;
    QtGeom q;
    int rc = make_qtgeom(ctx, g, min_size, max_size, q, true);
    if (rc) return rc;
    Carver c(workspace);
    QtWs w;
    carve_qt(c, g, q, false, w);
    unsigned long long *bits = c.take<unsigned long long>(g.bpstride);
    if (((c.off + 255) & ~255ull) > workspace_bytes) return fail(ctx, AEJ_ERR_CAPACITY, "workspace too small");
    w.qb.leaves = leaves; w.qb.states = states; w.qb.counts = reinterpret_cast<long long *>(counts);
    launch_pack_edge_bits(ctx->stream, g, edge, bits);
    if ((rc = run_quadtree(ctx, g, q, w, bits))) return rc;
    AEJ_HIP_CHECK(hipMemcpyAsync(ctx->h_flag, w.qb.overflow, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    AEJ_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (*ctx->h_flag) return fail(ctx, AEJ_ERR_CAPACITY, "leaf/state capacity exceeded");
    return 0;
}

extern "C" int aej_dct_quant_zigzag(aej_ctx *ctx, const float *norm, int H, int W, int layer, const int32_t *leaves, int64_t n_leaves,
                                    int32_t *coeffs, float *dct_f32)
{
    if (!ctx) return AEJ_ERR_ARG;
    if (call_in_flight(ctx)) return fail(ctx, AEJ_ERR_STATE, "%s between aej_encode_batch_begin and aej_encode_batch_end", __func__);
    if (!ctx->has_settings) return fail(ctx, AEJ_ERR_STATE, "aej_set_settings has not been called");
    if (H < 1 || W < 1 || layer < 0 || layer > 2 || n_leaves < 0) return fail(ctx, AEJ_ERR_ARG, "bad argument");
    if (H > 65535 || W > 65535) return fail(ctx, AEJ_ERR_UNSUPPORTED, "plane %dx%d: sides above 65535 pixels are not built", H, W);
    if (n_leaves == 0) return 0;
    if (!norm || !leaves || !coeffs) return fail(ctx, AEJ_ERR_ARG, "NULL buffer");
    AEJ_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // geometry: a single image whose `layer` is the given plane
    Geom g;
    memset(&g, 0, sizeof g);
    g.B = 1; g.nl = 3; g.H = H; g.W = W;
    for (int l = 0; l < 3; l++) { g.h[l] = H; g.w[l] = W; g.rh[l] = g.rw[l] = 1; g.poff[l] = 0; }
    g.pstride = (long long)H * W;
    QtGeom q;
    memset(&q, 0, sizeof q);
    q.bmin = ctx->bmin; q.bmax = ctx->bmax; q.cell = ctx->bmin; q.nsizes = ctx->nsizes;   // work_off / work_stride stay 0
    // stage-only scratch (not on the hot path): per-size work lists
    char *scratch = nullptr;
    size_t list_bytes = (size_t)n_leaves * sizeof(LeafWork);
    size_t total = 256 + (size_t)ctx->nsizes * ((list_bytes + 255) & ~(size_t)255);   // 256 B = [3 planes][kMaxSizes] counters
    const size_t big_off = total;
    total += (size_t)big_scratch_floats(ctx->bmax) * sizeof(float);
    AEJ_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&scratch), total));
    int *work_count = reinterpret_cast<int *>(scratch);
    LeafWork *work[kMaxSizes] = {};
    for (int k = 0; k < ctx->nsizes; k++) work[k] = reinterpret_cast<LeafWork *>(scratch + 256 + (size_t)k * ((list_bytes + 255) & ~(size_t)255));
    hipError_t e = hipMemsetAsync(scratch, 0, 256, st);
    if (e == hipSuccess) {
        launch_work_from_leaves(st, leaves, n_leaves, ctx->bmin, layer, work, work_count);
        int k = 0;
        for (int s = ctx->bmin; s <= ctx->bmax; s *= 2, k++) {
            DctArgs a;
            a.norm = norm; a.coeffs = coeffs; a.dct_f32 = dct_f32;
            a.work = work[k]; a.work_count = work_count; a.k = k; a.nplanes = 3;
            a.scratch = big_scratch_floats(ctx->bmax) ? reinterpret_cast<float *>(scratch + big_off) : nullptr;
            a.D = ctx->d_D[k]; a.zzinv = ctx->d_zzinv[k];
            for (int l = 0; l < 3; l++) a.qm[l] = ctx->d_qm[l][k];
            if (launch_dct(st, s, g, q, a, n_leaves, ctx->tune)) { e = hipErrorInvalidValue; break; }
        }
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    (void)hipFree(scratch);
    if (e != hipSuccess) return fail(ctx, AEJ_ERR_HIP, "aej_dct_quant_zigzag: %s", hipGetErrorString(e));
    return 0;
}

// ---- decode path (next-scope row: jpeg.py:274-297) -------------------------------------------------------------
// host helper: Jpeg._block_merge's walk (jpeg.py:424-448): leaf positions from the leaf sizes and the layer geometry
extern "C" int64_t aej_leaf_positions_host(const int32_t *sizes_host, int64_t n, int root, int H, int W, int32_t *xy_host)
{
    if (!sizes_host || !xy_host || n < 0 || root < 1) return -1;
    struct It { int x, y, s; };
    std::vector<It> stack;
    stack.push_back({ 0, 0, root });
    int64_t li = 0;
    while (!stack.empty()) {
        It it = stack.back();
        stack.pop_back();
        if (it.x >= W || it.y >= H || it.s == 0) continue;
        if (li >= n) return -2;      // a node inside the layer is left without a leaf: the sizes do not tile it
        if (it.s == sizes_host[li]) { xy_host[2 * li] = it.x; xy_host[2 * li + 1] = it.y; li++; }
        else {
            int h = it.s / 2;
            stack.push_back({ it.x + h, it.y + h, h });
            stack.push_back({ it.x, it.y + h, h });
            stack.push_back({ it.x + h, it.y, h });
            stack.push_back({ it.x, it.y, h });
        }
    }
    return li;
}

extern "C" int aej_color_convert_inverse(aej_ctx *ctx, int space, const float *in, float *out_rgb, int64_t n)
{
    if (!ctx) return AEJ_ERR_ARG;
    if (call_in_flight(ctx)) return fail(ctx, AEJ_ERR_STATE, "%s between aej_encode_batch_begin and aej_encode_batch_end", __func__);
    if (n < 0 || (n > 0 && (!in || !out_rgb))) return fail(ctx, AEJ_ERR_ARG, "bad buffer");
    if (n == 0) return 0;
    AEJ_HIP_CHECK(hipSetDevice(ctx->device));
    if (launch_color_inverse(ctx->stream, space, in, out_rgb, n)) return fail(ctx, AEJ_ERR_ARG, "Invalid color space id %d", space);
    AEJ_HIP_CHECK(hipGetLastError());
    return 0;
}

struct DecodeWs {
    float *big;
    float *planes;
    int *work_count;
    LeafWork *work[kMaxSizes];
    long long work_cap[kMaxSizes];
    unsigned long long bytes;
};

static void carve_decode(void *base, const Geom &g, const QtGeom &q, DecodeWs &w)
{
    Carver c(base);
    w.planes = c.take<float>((long long)g.B * g.pstride);
    w.work_count = c.take<int>((long long)g.B * 3 * kMaxSizes + 1);       // + 1: the "tables do not fit the plan" flag
    for (int k = 0; k < kMaxSizes; k++) { w.work[k] = nullptr; w.work_cap[k] = 0; }
    for (int k = 0; k < q.nsizes; k++) {
        w.work_cap[k] = q.work_stride[k] * g.B;
        w.work[k] = c.take<LeafWork>(w.work_cap[k] > 0 ? w.work_cap[k] : 1);
    }
    w.big = big_scratch_floats(q.bmax) ? c.take<float>(big_scratch_floats(q.bmax)) : nullptr;
    w.bytes = (c.off + 255) & ~255ull;
}

extern "C" uint64_t aej_decode_workspace_bytes(aej_ctx *ctx, int batch, int H, int W)
{
    if (check_encode_args(ctx, batch, H, W)) return 0;
    Geom g;
    QtGeom q;
    if (make_geom(ctx, ctx->space, batch, H, W, g) || make_qtgeom(ctx, g, ctx->bmin, ctx->bmax, q)) return 0;
    DecodeWs w;
    carve_decode(nullptr, g, q, w);
    return w.bytes;
}

extern "C" int aej_decode_batch(aej_ctx *ctx, const int32_t *coeffs, const int32_t *leaves, const int64_t *counts, int batch, int H, int W,
                                float *rgb_out, void *workspace, uint64_t workspace_bytes)
{
    int rc = check_encode_args(ctx, batch, H, W);
    if (rc) return rc;
    if (call_in_flight(ctx)) return fail(ctx, AEJ_ERR_STATE, "%s between aej_encode_batch_begin and aej_encode_batch_end", __func__);
    if (!coeffs || !leaves || !counts || !rgb_out || !workspace) return fail(ctx, AEJ_ERR_ARG, "NULL buffer");
    AEJ_HIP_CHECK(hipSetDevice(ctx->device));
    Geom g;
    QtGeom q;
    if ((rc = make_geom(ctx, ctx->space, batch, H, W, g))) return rc;
    if ((rc = make_qtgeom(ctx, g, ctx->bmin, ctx->bmax, q))) return rc;
    DecodeWs w;
    carve_decode(workspace, g, q, w);
    if (w.bytes > workspace_bytes) return fail(ctx, AEJ_ERR_CAPACITY, "workspace too small: need %llu bytes", w.bytes);
    hipStream_t st = ctx->stream;
    int *bad = w.work_count + (size_t)batch * 3 * kMaxSizes;
    AEJ_HIP_CHECK(hipMemsetAsync(w.work_count, 0, ((size_t)batch * 3 * kMaxSizes + 1) * sizeof(int), st));
    launch_work_from_tables(st, g, q, leaves, reinterpret_cast<const long long *>(counts), w.work, w.work_count, bad);
    int k = 0;
    for (int s = q.bmin; s <= q.bmax; s *= 2, k++) {
        IdctArgs a;
        a.coeffs = coeffs; a.planes = w.planes; a.work = w.work[k]; a.work_count = w.work_count; a.k = k; a.nplanes = batch * 3;
        a.scratch = w.big;
        a.D = ctx->d_D[k]; a.zz = ctx->d_zz[k]; a.zzinv = ctx->d_zzinv[k];
        for (int l = 0; l < 3; l++) { a.qm[l] = ctx->d_qm[l][k]; a.mid[l] = (float)kMid[ctx->space][l]; a.scale[l] = (float)kScale[ctx->space][l]; }
        if (launch_idct(st, s, g, q, a, w.work_cap[k])) return fail(ctx, AEJ_ERR_UNSUPPORTED, "no IDCT kernel for block size %d with %d planes", s, a.nplanes);
    }
    if (launch_upsample_color(st, ctx->space, g, w.planes, rgb_out)) return fail(ctx, AEJ_ERR_ARG, "bad colour space");
    AEJ_HIP_CHECK(hipGetLastError());
    AEJ_HIP_CHECK(hipMemcpyAsync(ctx->h_flag, bad, sizeof(int), hipMemcpyDeviceToHost, st));
    AEJ_HIP_CHECK(hipStreamSynchronize(st));
    if (*ctx->h_flag) return fail(ctx, AEJ_ERR_ARG, "corrupt stream: the leaf tables do not fit the plan (leaf count, block size outside %d-%d, or too many leaves of one size)", q.bmin, q.bmax);
    return 0;
}

// ---- opt-in GPU entropy stage (deflate.hip) -----------------------------------------------------------------------
static int deflate_geometry(aej_ctx *ctx, int batch, int H, int W, QtGeom &q)
{
    int rc = check_encode_args(ctx, batch, H, W);
    if (rc) return rc;
    Geom g;
    if ((rc = make_geom(ctx, ctx->space, batch, H, W, g))) return rc;
    return make_qtgeom(ctx, g, ctx->bmin, ctx->bmax, q);
}

extern "C" uint64_t aej_deflate_stream_bound(uint64_t raw_bytes) { return deflate_stream_bound(raw_bytes); }

// host only (no context, no device): the per-layer dynamic codes from the histograms aej_deflate_histogram counted
namespace aej { int deflate_build_table_host(const int *hist /* [320] */, int cover_all, unsigned *table /* [448] */); }      // deflate.hip
extern "C" int aej_deflate_build_tables(const int32_t *hist_host, const int32_t *cover_all, uint32_t *tables_host)
{
    if (!hist_host || !tables_host) return AEJ_ERR_ARG;
    for (int l = 0; l < 3; l++)
        if (deflate_build_table_host(hist_host + l * AEJ_DEFLATE_HIST_BINS, cover_all ? cover_all[l] : 1, tables_host + l * AEJ_DEFLATE_TABLE_WORDS)) return AEJ_ERR_CAPACITY;
    return 0;
}

// host only: 8-bit ingest of a host float32 batch (include/aej.h)
extern "C" int aej_pack_u8_levels_host(const float *rgb_host, int64_t n, uint8_t *u8_host, int threads)
{
    if (!rgb_host || !u8_host || n < 0) return AEJ_ERR_ARG;
    float lut[256];
    for (int k = 0; k < 256; k++) lut[k] = (float)k / 255.0f;        // the quotients Image.load forms (image.py:80) and the ingest kernel's table
    const int64_t kBlock = 1 << 16;
    const int64_t nblocks = (n + kBlock - 1) / kBlock;
    int nt = threads < 1 ? 1 : threads > 64 ? 64 : threads;
    if ((int64_t)nt > nblocks) nt = nblocks > 0 ? (int)nblocks : 1;
    std::atomic<int64_t> next(0);
    std::atomic<int> exact(1);
    auto work = [&]() {
        for (;;) {
            const int64_t blk = next.fetch_add(1);
            if (blk >= nblocks || !exact.load(std::memory_order_relaxed)) return;      // (another thread met a value that is no level: stop early)
            const int64_t lo = blk * kBlock, hi = std::min(n, lo + kBlock);
            unsigned bad = 0;
            for (int64_t i = lo; i < hi; i++) {
                const float x = rgb_host[i];
                // x * 255 is within half a unit of k for x = float32(k) / 255; anything outside [0, 255] (or NaN) maps to an entry that cannot compare equal
                float y = x * 255.0f + 0.5f;
                y = y >= 0.0f ? y : 0.0f;                                              // (NaN -> 0: lut[0] == NaN is false)
                const int k = y < 255.5f ? (int)y : 255;
                uint32_t a, b;
                memcpy(&a, &x, 4);
                memcpy(&b, &lut[k], 4);
                bad |= a ^ b;                                                          // bit-exact: -0.0f is not a level either
                u8_host[i] = (uint8_t)k;
            }
            if (bad) { exact.store(0, std::memory_order_relaxed); return; }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
    return exact.load();
}

extern "C" uint64_t aej_deflate_workspace_bytes(aej_ctx *ctx, int batch, int H, int W)
{
    QtGeom q;
    if (deflate_geometry(ctx, batch, H, W, q)) return 0;
    return deflate_workspace_bytes(batch, q.coeff_cap);
}

static int deflate_check(aej_ctx *ctx, const char *who, const void *a, const void *b, const void *ws, uint64_t ws_bytes, int batch, const QtGeom &q)
{
    if (call_in_flight(ctx)) return fail(ctx, AEJ_ERR_STATE, "%s between aej_encode_batch_begin and aej_encode_batch_end", who);
    if (!a || !b || !ws) return fail(ctx, AEJ_ERR_ARG, "NULL buffer");
    if (ws_bytes < deflate_workspace_bytes(batch, q.coeff_cap)) return fail(ctx, AEJ_ERR_CAPACITY, "workspace too small");
    return 0;
}

// the error word is the workspace's first: 1 = a stream does not fit, 2 = a count beyond its layer's capacity
static int deflate_finish(aej_ctx *ctx, void *workspace, uint64_t stream_stride)
{
    AEJ_HIP_CHECK(hipGetLastError());
    AEJ_HIP_CHECK(hipMemcpyAsync(ctx->h_flag, workspace, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    AEJ_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (*ctx->h_flag == 2) return fail(ctx, AEJ_ERR_ARG, "counts: a layer's coefficient count is negative or exceeds the capacity the plan gives it (corrupt or stale counts buffer)");
    if (*ctx->h_flag) return fail(ctx, AEJ_ERR_CAPACITY, "a deflate stream does not fit stream_stride = %llu bytes (aej_deflate_stream_bound gives a safe size)", (unsigned long long)stream_stride);
    return 0;
}

extern "C" int aej_deflate_histogram(aej_ctx *ctx, const int32_t *coeffs, const int64_t *counts, int batch, int H, int W, int32_t *hist, void *workspace,
                                     uint64_t workspace_bytes)
{
    QtGeom q;
    int rc = deflate_geometry(ctx, batch, H, W, q);
    if (rc) return rc;
    if (!hist) return fail(ctx, AEJ_ERR_ARG, "NULL buffer");
    if ((rc = deflate_check(ctx, __func__, coeffs, counts, workspace, workspace_bytes, batch, q))) return rc;
    AEJ_HIP_CHECK(hipSetDevice(ctx->device));
    launch_deflate_parse(ctx->stream, coeffs, reinterpret_cast<const long long *>(counts), batch, q.coeff_stride, q.coeff_off, q.coeff_cap, hist, workspace);
    return deflate_finish(ctx, workspace, 0);
}

extern "C" int aej_deflate_batch(aej_ctx *ctx, const int32_t *coeffs, const int64_t *counts, int batch, int H, int W, const uint32_t *tables, int reuse_parse,
                                 uint8_t *streams, uint64_t stream_stride, int64_t *sizes, void *workspace, uint64_t workspace_bytes)
{
    QtGeom q;
    int rc = deflate_geometry(ctx, batch, H, W, q);
    if (rc) return rc;
    if (!streams || !sizes) return fail(ctx, AEJ_ERR_ARG, "NULL buffer");
    if ((rc = deflate_check(ctx, __func__, coeffs, counts, workspace, workspace_bytes, batch, q))) return rc;
    if (stream_stride < 16 || (stream_stride & 3)) return fail(ctx, AEJ_ERR_CAPACITY, "stream_stride must be a multiple of 4 and at least 16");
    if ((reinterpret_cast<uintptr_t>(streams) & 3) != 0) return fail(ctx, AEJ_ERR_ARG, "streams must be 4-byte aligned");
    AEJ_HIP_CHECK(hipSetDevice(ctx->device));
    launch_deflate(ctx->stream, coeffs, reinterpret_cast<const long long *>(counts), batch, q.coeff_stride, q.coeff_off, q.coeff_cap, tables, reuse_parse ? 1 : 0,
                   streams, stream_stride, reinterpret_cast<long long *>(sizes), workspace);
    return deflate_finish(ctx, workspace, stream_stride);
}

// ---- evaluation metrics (evaluation_metrics.py:50-89) ------------------------------------------------------------
struct MetricsWs {
    double *acc;
    unsigned char *ga, *gb;
    float *xa, *xb;
    float *pyr[2][4];            // MS-SSIM scales 1..4 of both images, planar [B][3][h][w]
    int f, hp, wp;
    int lh[5], lw[5], lp[5];     // scale dimensions; lp[l] = padding applied when going from scale l-1 to l
    unsigned long long bytes;
};

static void carve_metrics(void *base, int B, int H, int W, MetricsWs &w)
{
    unsigned long long off = 0;
    auto take = [&](unsigned long long n) { void *p = base ? (char *)base + off : nullptr; off += (unsigned long long)align_up((long long)n, 256); return p; };
    w.acc = (double *)take((unsigned long long)B * kMetricSlots * sizeof(double));
    w.ga = (unsigned char *)take((unsigned long long)B * H * W);
    w.gb = (unsigned char *)take((unsigned long long)B * H * W);
    // piq.ssim: f = max(1, round(min(H, W) / 256)) -- Python round(): ties to even
    w.f = (int)nearbyint((double)(H < W ? H : W) / 256.0);
    if (w.f < 1) w.f = 1;
    w.hp = H / w.f; w.wp = W / w.f;
    w.xa = (float *)take((unsigned long long)B * w.hp * w.wp * 4);
    w.xb = (float *)take((unsigned long long)B * w.hp * w.wp * 4);
    w.lh[0] = H; w.lw[0] = W; w.lp[0] = 0;
    for (int l = 1; l < 5; l++) {
        const int p = (w.lh[l - 1] % 2) > (w.lw[l - 1] % 2) ? (w.lh[l - 1] % 2) : (w.lw[l - 1] % 2);
        w.lp[l] = p;
        w.lh[l] = (w.lh[l - 1] + p) / 2;
        w.lw[l] = (w.lw[l - 1] + p) / 2;
        for (int i = 0; i < 2; i++) w.pyr[i][l - 1] = (float *)take((unsigned long long)B * 3 * w.lh[l] * w.lw[l] * 4);
    }
    w.bytes = off;
}

extern "C" uint64_t aej_metrics_workspace_bytes(int batch, int H, int W)
{
    if (batch < 1 || H < 1 || W < 1) return 0;
    MetricsWs w;
    carve_metrics(nullptr, batch, H, W, w);
    return w.bytes;
}

extern "C" int aej_metrics_batch(aej_ctx *ctx, const float *img_a, const float *img_b, int batch, int H, int W, int which, double *out,
                                 void *workspace, uint64_t workspace_bytes)
{
    if (!ctx) return AEJ_ERR_ARG;
    if (call_in_flight(ctx)) return fail(ctx, AEJ_ERR_STATE, "%s between aej_encode_batch_begin and aej_encode_batch_end", __func__);
    if (!img_a || !img_b || !out || !workspace) return fail(ctx, AEJ_ERR_ARG, "NULL buffer");
    if (batch < 1 || H < 1 || W < 1) return fail(ctx, AEJ_ERR_ARG, "bad shape %d x %d x %d", batch, H, W);
    if ((which & ~7) || !(which & 7)) return fail(ctx, AEJ_ERR_ARG, "which must be a combination of AEJ_METRIC_PSNR | AEJ_METRIC_SSIM | AEJ_METRIC_MS_SSIM");
    AEJ_HIP_CHECK(hipSetDevice(ctx->device));
    MetricsWs w;
    carve_metrics(workspace, batch, H, W, w);
    if (w.bytes > workspace_bytes) return fail(ctx, AEJ_ERR_CAPACITY, "workspace too small: need %llu bytes, got %llu", w.bytes, (unsigned long long)workspace_bytes);
    const bool want_ssim = which & AEJ_METRIC_SSIM, want_ms = which & AEJ_METRIC_MS_SSIM;
    // piq/ssim.py _ssim_per_channel / piq/ms_ssim.py _multi_scale_ssim raise ValueError for these
    if (want_ssim && (w.hp < 11 || w.wp < 11)) return fail(ctx, AEJ_ERR_ARG, "Kernel size can't be greater than actual input size. Input size: %dx%d. Kernel size: 11x11", w.hp, w.wp);
    if (want_ms && (H < 161 || W < 161)) return fail(ctx, AEJ_ERR_ARG, "Invalid size of the input images, expected at least 161x161.");
    hipStream_t st = ctx->stream;
    float g11[11];
    {
        double e[11], sum = 0.0;
        for (int i = 0; i < 11; i++) { double c = (double)i - 5.0; e[i] = exp(-(c * c) / (2.0 * 1.5 * 1.5)); sum += e[i]; }
        for (int i = 0; i < 11; i++) g11[i] = (float)(e[i] / sum);
    }
    AEJ_HIP_CHECK(hipMemsetAsync(w.acc, 0, (size_t)batch * kMetricSlots * sizeof(double), st));
    launch_metric_prep(st, img_a, img_b, batch, (long long)H * W, w.acc, want_ssim ? w.ga : nullptr, want_ssim ? w.gb : nullptr);
    long long n_ssim = 0, n_level[5] = { 0, 0, 0, 0, 0 };
    if (want_ssim) {
        launch_metric_pool_grey(st, w.ga, w.gb, batch, H, W, w.f, w.hp, w.wp, w.xa, w.xb);
        launch_ssim_level(st, false, w.xa, w.xb, batch, 1, w.hp, w.wp, g11, w.acc, kMetricSlotGrey, true);
        n_ssim = (long long)(w.hp - 10) * (w.wp - 10);
    }
    if (want_ms) {
        for (int l = 0; l < 5; l++) {
            const float *xa = l == 0 ? img_a : w.pyr[0][l - 1], *xb = l == 0 ? img_b : w.pyr[1][l - 1];
            if (l > 0) {
                const float *pa = l == 1 ? img_a : w.pyr[0][l - 2], *pb = l == 1 ? img_b : w.pyr[1][l - 2];
                if (l == 1) {
                    // even sizes: scale 0's SSIM kernel has written scale 1 on its way (no padding to replicate)
                    if (w.lp[1] != 0) launch_pool2_rgb(st, pa, pb, batch, w.lh[0], w.lw[0], w.lp[1], w.lh[1], w.lw[1], w.pyr[0][0], w.pyr[1][0]);
                } else {
                    launch_pool2(st, false, pa, batch, 3, w.lh[l - 1], w.lw[l - 1], w.lp[l], w.lh[l], w.lw[l], w.pyr[0][l - 1]);
                    launch_pool2(st, false, pb, batch, 3, w.lh[l - 1], w.lw[l - 1], w.lp[l], w.lh[l], w.lw[l], w.pyr[1][l - 1]);
                }
            }
            const bool fused_pool = l == 0 && w.lp[1] == 0;
            launch_ssim_level(st, l == 0, xa, xb, batch, 3, w.lh[l], w.lw[l], g11, w.acc, kMetricSlotScales + l * 6, l == 4, fused_pool ? w.pyr[0][0] : nullptr,
                              fused_pool ? w.pyr[1][0] : nullptr);
            n_level[l] = (long long)(w.lh[l] - 10) * (w.lw[l] - 10);
        }
    }
    launch_metric_final(st, w.acc, batch, (long long)H * W, n_ssim, n_level, out);
    AEJ_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int aej_get_hysteresis_stats(aej_ctx *ctx, int64_t *out_host)
{
    if (!ctx || !out_host) return AEJ_ERR_ARG;
    out_host[0] = ctx->n_encode_calls;
    out_host[1] = ctx->last_hyst_queued;
    return 0;
}

extern "C" int aej_set_graph_mode(aej_ctx *ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 2) return AEJ_ERR_ARG;
    if (call_in_flight(ctx)) return fail(ctx, AEJ_ERR_STATE, "%s between aej_encode_batch_begin and aej_encode_batch_end", __func__);
    ctx->graph_mode = mode;
    if (mode == 0) drop_graphs(ctx);
    return 0;
}

extern "C" int aej_set_sub_batches(aej_ctx *ctx, int n)
{
    if (!ctx || n < 0 || n > aej_ctx::kMaxSub) return AEJ_ERR_ARG;
    ctx->sub_mode = n;
    return 0;
}

extern "C" int64_t aej_get_split_calls(aej_ctx *ctx) { return ctx ? (int64_t)ctx->n_split_calls : -1; }

extern "C" int aej_set_hw_queues(aej_ctx *ctx, int n)
{
    if (!ctx || n < 1) return AEJ_ERR_ARG;
    ctx->hw_queues = n;
    return 0;
}

extern "C" int aej_get_schedule_host(aej_ctx *ctx, int batch, int H, int W, int32_t *out_host)
{
    int rc = check_encode_args(ctx, batch, H, W);
    if (rc) return rc;
    if (!out_host) return fail(ctx, AEJ_ERR_ARG, "out_host is NULL");
    Geom g;
    if ((rc = make_geom(ctx, ctx->space, batch, H, W, g))) return rc;
    const int n = sub_batches(ctx, g, ctx->hw_queues), n_wide = sub_batches(ctx, g, 16);
    out_host[0] = ctx->hw_queues;
    out_host[1] = n;
    out_host[2] = (ctx->hw_queues < 8 && n_wide > n) ? 1 : 0;
    out_host[3] = 0;
    return 0;
}

// ---- options: every tuning / A-B choice of the library is a per-context value set through this entry (include/aej.h has the table) ----
namespace {
struct OptionDef { const char *name; int aej::Tuning::*field; int aej_ctx::*ctx_field; long long lo, hi; bool drops_graphs; };
const OptionDef kOptions[] = {
    { "color_strip", &aej::Tuning::color_strip, nullptr, 0, 1, true },
    { "color_strip_rows", &aej::Tuning::color_strip_rows, nullptr, 0, 64, true },
    { "color_workgroups", &aej::Tuning::color_workgroups, nullptr, 0, 1 << 16, true },
    { "planes_row_major", &aej::Tuning::planes_row_major, nullptr, 0, 1, true },
    { "dct64_kernel", &aej::Tuning::dct64_kernel, nullptr, 0, 4, true },
    { "dct_small_workgroups", &aej::Tuning::dct_small_workgroups, nullptr, 0, 1 << 16, true },
    { "sobel_lds", &aej::Tuning::sobel_lds, nullptr, 0, 1, true },
    { "sobel_xcd", &aej::Tuning::sobel_xcd, nullptr, 0, 1, true },
    { "dct_multi", &aej::Tuning::dct_multi, nullptr, 0, 1, true },
    { "sub_chain", nullptr, &aej_ctx::sub_chain, -1, 3, false },
};
const OptionDef *find_option(const char *name)
{
    if (!name) return nullptr;
    for (const OptionDef &o : kOptions) if (!strcmp(o.name, name)) return &o;
    return nullptr;
}
}  // namespace

extern "C" int aej_set_option(aej_ctx *ctx, const char *name, int64_t value)
{
    if (!ctx) return AEJ_ERR_ARG;
    if (call_in_flight(ctx)) return fail(ctx, AEJ_ERR_STATE, "%s between aej_encode_batch_begin and aej_encode_batch_end", __func__);
    const OptionDef *o = find_option(name);
    if (!o) return fail(ctx, AEJ_ERR_ARG, "aej_set_option: unknown option '%s'", name ? name : "(null)");
    if (value < o->lo || value > o->hi || (o->field == &aej::Tuning::dct64_kernel && value != 0 && value != 1 && value != 4))
        return fail(ctx, AEJ_ERR_ARG, "aej_set_option: %s = %lld outside its range [%lld, %lld]", name, (long long)value, o->lo, o->hi);
    if (o->field) ctx->tune.*(o->field) = (int)value;
    else ctx->*(o->ctx_field) = (int)value;
    if (o->drops_graphs) drop_graphs(ctx);      // a captured graph holds the launches the old value chose
    return 0;
}

extern "C" int aej_get_option(aej_ctx *ctx, const char *name, int64_t *value_host)
{
    if (!ctx || !value_host) return AEJ_ERR_ARG;
    const OptionDef *o = find_option(name);
    if (!o) return fail(ctx, AEJ_ERR_ARG, "aej_get_option: unknown option '%s'", name ? name : "(null)");
    *value_host = o->field ? ctx->tune.*(o->field) : ctx->*(o->ctx_field);
    return 0;
}

extern "C" int aej_test_fail_after_stage(aej_ctx *ctx, int stage)
{
    if (!ctx || stage < -1 || stage >= AEJ_N_STAGES) return AEJ_ERR_ARG;
    ctx->fail_after = stage;
    return 0;
}

extern "C" int aej_get_graph_stats(aej_ctx *ctx, int64_t *out_host)
{
    if (!ctx || !out_host) return AEJ_ERR_ARG;
    out_host[0] = (int64_t)ctx->n_graph_launches;
    out_host[1] = (int64_t)ctx->n_graph_captures;
    out_host[2] = (int64_t)ctx->graphs.size();
    return 0;
}

extern "C" int aej_set_stream(aej_ctx *ctx, void *hip_stream)
{
    if (!ctx) return AEJ_ERR_ARG;
    if (call_in_flight(ctx)) return fail(ctx, AEJ_ERR_STATE, "%s between aej_encode_batch_begin and aej_encode_batch_end", __func__);
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if (s == ctx->stream) return 0;
    AEJ_HIP_CHECK(hipSetDevice(ctx->device));
    AEJ_HIP_CHECK(hipStreamSynchronize(ctx->stream));     // nothing of ours is left running on the stream we leave
    ctx->stream = s;
    return 0;
}

extern "C" int aej_set_profiling(aej_ctx *ctx, int enable)
{
    if (!ctx) return AEJ_ERR_ARG;
    if (call_in_flight(ctx)) return fail(ctx, AEJ_ERR_STATE, "%s between aej_encode_batch_begin and aej_encode_batch_end", __func__);
    ctx->profiling = enable != 0;
    return 0;
}

extern "C" int aej_get_stage_ms(aej_ctx *ctx, float *ms_host)
{
    if (!ctx || !ms_host) return AEJ_ERR_ARG;
    for (int i = 0; i < AEJ_N_STAGES; i++) ms_host[i] = ctx->stage_ms[i];
    return 0;
}

extern "C" const char *aej_stage_name(int i)
{
    static const char *names[AEJ_N_STAGES] = { "clear", "color_planes", "clahe_lut", "clahe_blur", "thresholds", "sobel_nms",
                                               "hysteresis", "quadtree", "dct2", "dct4", "dct8", "dct16", "dct32", "dct64", "dct128", "dct256", "dct512", "dct1024" };
    return i >= 0 && i < AEJ_N_STAGES ? names[i] : "";
}
