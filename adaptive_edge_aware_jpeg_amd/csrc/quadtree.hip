// quadtree.hip -- QuadTree._build_tree + get_leaves_and_states (src/jpeg/quadtree.py:93-165) without a
// tree walk.  gfx950 only.
//
// The reference splits top-down with an explicit stack; a node (x, y, s) splits iff
//     s > max_size  or  (s > min_size and any(edge[y:y+s, x:x+s])).
// Order-free restatement used here (equivalence is checked against the reference-generated golden cases):
//   * "any(edge)" over aligned squares is an OR-pyramid over min-size cells;
//   * the pre-order DFS sequence of (internal '01' / leaf '00' / out-of-bounds child '10') symbols is the
//     sequence of all existing nodes sorted by (Morton code of origin, descending size), so a cell in Morton
//     order emits the nodes that *originate* at it, largest first;
//   * leaves in DFS order are therefore in ascending Morton order of their origin cell.
// Three passes over 256-cell (16x16, Morton-aligned) chunks: OR-pyramid, per-chunk counts, prefix sums, emit.
#include "aej_common.h"
#include "aej_launch.h"

namespace aej {

__device__ __forceinline__ unsigned compact1by1(unsigned v)
{
    v &= 0x55555555u;
    v = (v | (v >> 1)) & 0x33333333u;
    v = (v | (v >> 2)) & 0x0F0F0F0Fu;
    v = (v | (v >> 4)) & 0x00FF00FFu;
    v = (v | (v >> 8)) & 0x0000FFFFu;
    return v;
}
// Morton code with x in the even bits, y in the odd bits (children order TL, TR, BL, BR: quadtree.py:123-131)
__device__ __forceinline__ void morton_decode(unsigned m, int &x, int &y) { x = (int)compact1by1(m); y = (int)compact1by1(m >> 1); }

__device__ __forceinline__ long long lvl_off(int ncell, int lv)
{
    long long o = 0;
    for (int j = 0; j < lv; j++) { long long s = ncell >> j; o += s * s; }
    return o;
}

// ------------------------------------------------------------------------------------------------
// pass 1: OR-pyramid.  One block per chunk of 16x16 cells; levels 0..4 reduced in LDS, higher levels
// (only those that can still matter, cell<<lv <= bmax) are set with idempotent stores of 1.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_qt_cells(Geom g, QtGeom q, const unsigned long long *__restrict__ edge_bits,
                                                  unsigned char *__restrict__ pyr_all)
{
    __shared__ unsigned char lv[5][256];
    const int tid = threadIdx.x, l = blockIdx.y, b = blockIdx.z;
    const int ncell = q.ncell[l], ltot = q.ltot[l], cell = q.cell;
    const long long ncell2 = (long long)ncell * ncell;
    if ((long long)blockIdx.x * 256 >= ncell2) return;
    const int w = g.w[l], h = g.h[l];
    int ccx, ccy;
    morton_decode(blockIdx.x, ccx, ccy);
    ccx *= 16; ccy *= 16;
    if (ccx * cell >= w || ccy * cell >= h) return;   // chunk origin out of bounds => whole chunk is
    int lx, ly;
    morton_decode(tid, lx, ly);
    const int cx = ccx + lx, cy = ccy + ly;
    const unsigned long long *src = edge_bits + (long long)b * g.bpstride + g.bpoff[l];
    const int wpr = g.wpr[l];
    unsigned char *pyr = pyr_all + (long long)b * q.pyr_stride + q.pyr_off[l];
    const bool valid = (long long)tid < ncell2;
    unsigned char e = 0;
    if (valid) {
        int x0 = cx * cell, y0 = cy * cell;
        int x1 = min(x0 + cell, w), y1 = min(y0 + cell, h);
        for (int y = y0; y < y1; y++)
            for (int xw = x0 >> 6; xw <= (x1 - 1) >> 6; xw++) {
                int lo = max(x0, xw * 64) - xw * 64, hi = min(x1, xw * 64 + 64) - xw * 64;
                unsigned long long mask = (hi - lo == 64) ? ~0ull : (((1ull << (hi - lo)) - 1ull) << lo);
                e |= (src[(long long)y * wpr + xw] & mask) ? 1 : 0;
            }
        pyr[(long long)cy * ncell + cx] = e;
    }
    lv[0][tid] = e;
    __syncthreads();
    const int lmax_in = ltot < 4 ? ltot : 4;
    for (int k = 1; k <= lmax_in; k++) {
        int n = 256 >> (2 * k);
        if (tid < n) {
            unsigned char v = lv[k - 1][4 * tid] | lv[k - 1][4 * tid + 1] | lv[k - 1][4 * tid + 2] | lv[k - 1][4 * tid + 3];
            lv[k][tid] = v;
            int mx, my;
            morton_decode(tid, mx, my);
            int side = ncell >> k;
            int px = (ccx >> k) + mx, py = (ccy >> k) + my;
            if (px < side && py < side) pyr[lvl_off(ncell, k) + (long long)py * side + px] = v;
        }
        __syncthreads();
    }
    if (tid == 0 && ltot > 4 && lv[4][0]) {
        for (int k = 5; k <= ltot && (cell << k) <= q.bmax; k++) {
            int side = ncell >> k;
            pyr[lvl_off(ncell, k) + (long long)(ccy >> k) * side + (ccx >> k)] = 1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// per-cell node evaluation shared by the count and emit passes
// ------------------------------------------------------------------------------------------------
struct CellNodes {
    int nsym;        // number of symbols originating at this cell
    unsigned syms;   // 2 bits per symbol, first emitted in the low bits
    int leaf_lvl;    // level of the leaf originating here, or -1
};

__device__ __forceinline__ CellNodes eval_cell(const QtGeom &q, int l, int w, int h, const unsigned char *__restrict__ pyr, unsigned gidx)
{
    CellNodes r;
    r.nsym = 0; r.syms = 0; r.leaf_lvl = -1;
    const int ncell = q.ncell[l], ltot = q.ltot[l], cell = q.cell;
    int cx, cy;
    morton_decode(gidx, cx, cy);
    int a = gidx == 0 ? ltot : min((int)(__ffs(gidx) - 1) >> 1, ltot);
    if (a < ltot) {
        // the level-a node exists only if its parent is in bounds and splits
        int mask = ~((2 << a) - 1);
        int pcx = cx & mask, pcy = cy & mask;
        if (pcx * cell >= w || pcy * cell >= h) return r;
        int psize = cell << (a + 1);
        bool psplit = psize > q.bmax;
        if (!psplit && psize > q.bmin) {
            int side = ncell >> (a + 1);
            psplit = pyr[lvl_off(ncell, a + 1) + (long long)(cy >> (a + 1)) * side + (cx >> (a + 1))] != 0;
        }
        if (!psplit) return r;
    }
    if (cx * cell >= w || cy * cell >= h) {   // quadtree.py:109-110, 153-155: absent child
        r.nsym = 1; r.syms = 2u;
        return r;
    }
    for (int lvn = a; lvn >= 0; lvn--) {
        int size = cell << lvn;
        bool split = size > q.bmax;
        if (!split && size > q.bmin) {
            int side = ncell >> lvn;
            split = pyr[lvl_off(ncell, lvn) + (long long)(cy >> lvn) * side + (cx >> lvn)] != 0;
        }
        if (split) {
            r.syms |= 1u << (2 * r.nsym);
            r.nsym++;
        } else {
            r.nsym++;          // '00'
            r.leaf_lvl = lvn;
            break;
        }
    }
    return r;
}

__device__ __forceinline__ int wave_incl_scan(int v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

// exclusive scan of three per-thread values over a 256-thread block: wave shuffles + one barrier
__device__ __forceinline__ void block_excl_scan3(int v0, int v1, int v2, int (*s_w)[4], int tid, int &e0, int &e1, int &e2)
{
    const int lane = tid & 63, wv = tid >> 6;
    int i0 = wave_incl_scan(v0, lane), i1 = wave_incl_scan(v1, lane), i2 = wave_incl_scan(v2, lane);
    if (lane == 63) { s_w[0][wv] = i0; s_w[1][wv] = i1; s_w[2][wv] = i2; }
    __syncthreads();
    int p0 = 0, p1 = 0, p2 = 0;
    for (int k = 0; k < wv; k++) { p0 += s_w[0][k]; p1 += s_w[1][k]; p2 += s_w[2][k]; }
    e0 = p0 + i0 - v0; e1 = p1 + i1 - v1; e2 = p2 + i2 - v2;
}

// pass 2: per-chunk totals (symbols, leaves, coefficients, leaves per block size)
__global__ __launch_bounds__(256) void k_qt_count(Geom g, QtGeom q, const unsigned char *__restrict__ pyr_all, int *__restrict__ chunk_cnt)
{
    __shared__ int s_red[3 + kMaxSizes][4];
    const int tid = threadIdx.x, l = blockIdx.y, b = blockIdx.z;
    const long long ncell2 = (long long)q.ncell[l] * q.ncell[l];
    if ((long long)blockIdx.x >= q.nchunk[l]) return;
    const unsigned gidx = blockIdx.x * 256u + tid;
    const unsigned char *pyr = pyr_all + (long long)b * q.pyr_stride + q.pyr_off[l];
    int nsym = 0, nleaf = 0, ncoef = 0, lvl = -1;
    if ((long long)gidx < ncell2) {
        CellNodes c = eval_cell(q, l, g.w[l], g.h[l], pyr, gidx);
        nsym = c.nsym;
        lvl = c.leaf_lvl;
        if (lvl >= 0) { nleaf = 1; int s = q.cell << lvl; ncoef = s * s; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        nsym += __shfl_down(nsym, o);
        nleaf += __shfl_down(nleaf, o);
        ncoef += __shfl_down(ncoef, o);
    }
    const int wv = tid >> 6;
    if ((tid & 63) == 0) { s_red[0][wv] = nsym; s_red[1][wv] = nleaf; s_red[2][wv] = ncoef; }
    for (int k = 0; k < q.nsizes; k++) {
        unsigned long long m = __ballot(lvl == k);
        if ((tid & 63) == 0) s_red[3 + k][wv] = __popcll(m);
    }
    __syncthreads();
    if (tid < 3 + kMaxSizes) {
        int *o = chunk_cnt + ((long long)b * q.chunk_stride + q.chunk_off[l] + blockIdx.x) * kChunkInts;
        int v = (tid < 3 || tid - 3 < q.nsizes) ? s_red[tid][0] + s_red[tid][1] + s_red[tid][2] + s_red[tid][3] : 0;
        o[tid < 3 ? tid : tid + 1] = v;
    }
}

// pass 3: exclusive scan of the chunk records per (image, layer); totals -> counts and per-plane work counts
__global__ __launch_bounds__(1024) void k_qt_scan(Geom g, QtGeom q, int *__restrict__ chunk_cnt, long long *__restrict__ counts,
                                                  int *__restrict__ work_count)
{
    constexpr int NQ = 3 + kMaxSizes;
    __shared__ int s_w[NQ][16];
    __shared__ int carry[NQ];
    const int tid = threadIdx.x, l = blockIdx.x, b = blockIdx.y;
    const int lane = tid & 63, wv = tid >> 6;
    int *base = chunk_cnt + ((long long)b * q.chunk_stride + q.chunk_off[l]) * kChunkInts;
    const int n = q.nchunk[l];
    const int nq = 3 + q.nsizes;
    if (tid < NQ) carry[tid] = 0;
    __syncthreads();
    for (int start = 0; start < n; start += 1024) {
        const int i = start + tid;
        int v[NQ], inc[NQ];
#pragma unroll
        for (int c = 0; c < NQ; c++) {
            v[c] = (i < n && c < nq) ? base[i * kChunkInts + (c < 3 ? c : c + 1)] : 0;
            inc[c] = wave_incl_scan(v[c], lane);
            if (lane == 63) s_w[c][wv] = inc[c];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NQ; c++) {
            int p = carry[c];
            for (int k = 0; k < wv; k++) p += s_w[c][k];
            if (i < n && c < nq) base[i * kChunkInts + (c < 3 ? c : c + 1)] = p + inc[c] - v[c];
        }
        __syncthreads();
        if (tid < NQ) {
            int t = 0;
            for (int k = 0; k < 16; k++) t += s_w[tid][k];
            carry[tid] += t;
        }
        __syncthreads();
    }
    if (tid == 0) {
        long long *o = counts + ((long long)b * 3 + l) * 4;
        o[0] = carry[2];      // n_coeffs
        o[1] = carry[1];      // n_leaves
        o[2] = carry[0];      // n_states
        o[3] = q.root[l];
    }
    if (work_count && tid < kMaxSizes) work_count[((long long)b * 3 + l) * kMaxSizes + tid] = tid < q.nsizes ? carry[3 + tid] : 0;
}

// pass 4: emit symbols, leaf table and the per-size DCT work lists (positions come from the scans: deterministic,
// Morton-ordered lists, no global atomics)
__global__ __launch_bounds__(256) void k_qt_emit(Geom g, QtGeom q, QtBuffers qb)
{
    __shared__ int s_w[3][4];
    __shared__ int s_wc[4][kMaxSizes];
    const int tid = threadIdx.x, l = blockIdx.y, b = blockIdx.z;
    const int lane = tid & 63, wv = tid >> 6;
    const long long ncell2 = (long long)q.ncell[l] * q.ncell[l];
    if ((long long)blockIdx.x >= q.nchunk[l]) return;
    const unsigned gidx = blockIdx.x * 256u + tid;
    const unsigned char *pyr = qb.pyr + (long long)b * q.pyr_stride + q.pyr_off[l];
    const int *coff = qb.chunk_cnt + ((long long)b * q.chunk_stride + q.chunk_off[l] + blockIdx.x) * kChunkInts;
    const int sym_base = coff[0], leaf_base = coff[1], coef_base = coff[2];

    CellNodes c;
    c.nsym = 0; c.syms = 0; c.leaf_lvl = -1;
    if ((long long)gidx < ncell2) c = eval_cell(q, l, g.w[l], g.h[l], pyr, gidx);
    const int size = c.leaf_lvl >= 0 ? (q.cell << c.leaf_lvl) : 0;
    int e0, e1, e2;
    block_excl_scan3(c.nsym, c.leaf_lvl >= 0 ? 1 : 0, size * size, s_w, tid, e0, e1, e2);
    const int sym_pos = sym_base + e0, leaf_pos = leaf_base + e1, coef_pos = coef_base + e2;

    unsigned char *st = qb.states + (long long)b * q.state_stride + q.state_off[l];
    for (int k = 0; k < c.nsym; k++) {
        if (sym_pos + k < q.state_cap[l]) st[sym_pos + k] = (unsigned char)((c.syms >> (2 * k)) & 3u);
        else *qb.overflow = 1;
    }
    int cx = 0, cy = 0;
    bool leaf_ok = false;
    if (c.leaf_lvl >= 0) {
        morton_decode(gidx, cx, cy);
        if (leaf_pos < q.leaf_cap[l] && (long long)coef_pos + (long long)size * size <= q.coeff_cap[l]) {
            int *lf = qb.leaves + ((long long)b * q.leaf_stride + q.leaf_off[l] + leaf_pos) * 4;
            reinterpret_cast<int4 *>(lf)[0] = make_int4(cx * q.cell, cy * q.cell, size, coef_pos);
            leaf_ok = true;
        } else {
            *qb.overflow = 1;
        }
    }
    if (!qb.work_count) return;
    // rank of this leaf among the leaves of its size inside the chunk
    int rank = 0;
    for (int k = 0; k < q.nsizes; k++) {
        unsigned long long m = __ballot(c.leaf_lvl == k);
        if (c.leaf_lvl == k) rank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wc[wv][k] = __popcll(m);
    }
    __syncthreads();
    if (leaf_ok) {
        const int k = c.leaf_lvl;      // size == bmin << k (the codec path always has cell == bmin)
        for (int w2 = 0; w2 < wv; w2++) rank += s_wc[w2][k];
        long long pos = (long long)coff[4 + k] + rank;
        long long seg = (long long)b * q.work_stride[k] + q.work_off[l][k];
        if (seg + pos < qb.work_cap[k])
            reinterpret_cast<int4 *>(qb.work[k])[seg + pos] = make_int4(b * 3 + l, cx * q.cell, cy * q.cell, coef_pos);
        else
            *qb.overflow = 1;
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static int max_chunks(const Geom &g, const QtGeom &q)
{
    int m = 1;
    for (int l = 0; l < g.nl; l++) if (q.nchunk[l] > m) m = q.nchunk[l];
    return m;
}

void launch_qt_cells(hipStream_t st, const Geom &g, const QtGeom &q, const unsigned long long *edge_bits, const QtBuffers &qb)
{
    hipLaunchKernelGGL(k_qt_cells, dim3(max_chunks(g, q), g.nl, g.B), dim3(256), 0, st, g, q, edge_bits, qb.pyr);
}
void launch_qt_count(hipStream_t st, const Geom &g, const QtGeom &q, const QtBuffers &qb)
{
    hipLaunchKernelGGL(k_qt_count, dim3(max_chunks(g, q), g.nl, g.B), dim3(256), 0, st, g, q, qb.pyr, qb.chunk_cnt);
}
void launch_qt_scan(hipStream_t st, const Geom &g, const QtGeom &q, const QtBuffers &qb)
{
    hipLaunchKernelGGL(k_qt_scan, dim3(g.nl, g.B), dim3(1024), 0, st, g, q, qb.chunk_cnt, qb.counts, qb.work_count);
}
void launch_qt_emit(hipStream_t st, const Geom &g, const QtGeom &q, const QtBuffers &qb)
{
    hipLaunchKernelGGL(k_qt_emit, dim3(max_chunks(g, q), g.nl, g.B), dim3(256), 0, st, g, q, qb);
}

}  // namespace aej
