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
// OR-pyramid and node evaluation, one WAVE per chunk.
// A chunk is a Morton-aligned square of 16x16 min-size cells (256 cells); lane t owns the four sibling cells
// 4t..4t+3 (= one level-1 node), so levels 0..1 of the pyramid are lane-local and levels 2..4 are nibble / 16-bit /
// whole-word tests on one wave ballot.  Levels >= 5 (only needed when bmax/bmin >= 32) are a tiny global array
// filled by k_qt_upper with idempotent stores of 1.  No LDS, no workgroup barriers.
// ------------------------------------------------------------------------------------------------
struct ChunkEdges {
    unsigned e0;             // bit i = OR over cell 4*lane + i
    bool e1, e2, e3, e4;     // OR over the lane's level-1 / 2 / 3 / 4 ancestors
};

__device__ __forceinline__ ChunkEdges chunk_edges(const Geom &g, const QtGeom &q, int l, int b, const unsigned long long *__restrict__ edge_bits,
                                                  unsigned chunk, int lane)
{
    const int ncell = q.ncell[l], cell = q.cell;
    const int w = g.w[l], h = g.h[l], wpr = g.wpr[l];
    int ccx, ccy, lx, ly;
    morton_decode(chunk, ccx, ccy);
    morton_decode((unsigned)lane, lx, ly);
    const unsigned long long *src = edge_bits + (long long)b * g.bpstride + g.bpoff[l];
    ChunkEdges E;
    E.e0 = 0;
    if (cell == 4 && ncell >= 16) {
        // fast path (min block 4): the chunk is one 64x64 bit-plane tile = 64 contiguous words; this lane's four cells are the
        // 8x8-pixel square at rows 8*ly.., bits 8*lx..: eight word loads issued together, then nibble tests
        const int X0 = ccx * 64, Y0 = ccy * 64;
        if (X0 < w && Y0 < h) {
            const unsigned long long *tile = src + bp_index(Y0, X0 >> 6, wpr);
            unsigned rows[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int y = Y0 + 8 * ly + r;
                rows[r] = y < h ? (unsigned)((tile[8 * ly + r] >> (8 * lx)) & 0xFFull) : 0u;
            }
            const unsigned top = rows[0] | rows[1] | rows[2] | rows[3], bot = rows[4] | rows[5] | rows[6] | rows[7];
            E.e0 = ((top & 0x0Fu) ? 1u : 0u) | ((top & 0xF0u) ? 2u : 0u) | ((bot & 0x0Fu) ? 4u : 0u) | ((bot & 0xF0u) ? 8u : 0u);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int cx = ccx * 16 + lx * 2 + (i & 1), cy = ccy * 16 + ly * 2 + (i >> 1);
            if (cx < ncell && cy < ncell && cx * cell < w && cy * cell < h) {
                int x0 = cx * cell, y0 = cy * cell;
                int x1 = min(x0 + cell, w), y1 = min(y0 + cell, h);
                bool e = false;
                for (int y = y0; y < y1; y++)
                    for (int xw = x0 >> 6; xw <= (x1 - 1) >> 6; xw++) {
                        int lo = max(x0, xw * 64) - xw * 64, hi = min(x1, xw * 64 + 64) - xw * 64;
                        unsigned long long mask = (hi - lo == 64) ? ~0ull : (((1ull << (hi - lo)) - 1ull) << lo);
                        e |= (src[bp_index(y, xw, wpr)] & mask) != 0;
                    }
                E.e0 |= (e ? 1u : 0u) << i;
            }
        }
    }
    E.e1 = E.e0 != 0;
    const unsigned long long m = __ballot(E.e1);
    E.e2 = ((m >> (lane & ~3)) & 0xFull) != 0;
    E.e3 = ((m >> (lane & ~15)) & 0xFFFFull) != 0;
    E.e4 = m != 0;
    return E;
}

// The chunk kernels run on a grid of (sum over layers of ceil(nchunk / 4), batch): the layers have different chunk counts (a 4:2:0
// chroma plane has a quarter of the luma's), and a grid sized for the largest layer launched twice as many waves as it needed.
__device__ __forceinline__ bool locate_chunk_block(const QtGeom &q, int nl, int bx, int &l, unsigned &chunk0)
{
    for (l = 0; l < nl; l++) {
        const int nb = (q.nchunk[l] + 3) >> 2;
        if (bx < nb) { chunk0 = (unsigned)bx * 4u; return true; }
        bx -= nb;
    }
    return false;
}

__global__ __launch_bounds__(256) void k_qt_upper(Geom g, QtGeom q, const unsigned long long *__restrict__ edge_bits, unsigned char *__restrict__ pyr_all)
{
    const int b = blockIdx.y;
    int l;
    unsigned chunk;
    if (!locate_chunk_block(q, g.nl, (int)blockIdx.x, l, chunk)) return;
    const int lane = threadIdx.x & 63;
    chunk += threadIdx.x >> 6;
    const int ncell = q.ncell[l], ltot = q.ltot[l], cell = q.cell;
    if ((long long)chunk >= q.nchunk[l] || ltot <= 4) return;
    int ccx, ccy;
    morton_decode(chunk, ccx, ccy);
    if (ccx * 16 * cell >= g.w[l] || ccy * 16 * cell >= g.h[l]) return;
    ChunkEdges E = chunk_edges(g, q, l, b, edge_bits, chunk, lane);
    if (lane == 0 && E.e4) {
        unsigned char *pyr = pyr_all + (long long)b * q.pyr_stride + q.pyr_off[l];
        for (int k = 5; k <= ltot && (cell << k) <= q.bmax; k++) {
            int side = ncell >> k;
            pyr[lvl_off(ncell, k) + (long long)((ccy * 16) >> k) * side + ((ccx * 16) >> k)] = 1;
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

// edge(level lvn ancestor of cell gidx = 256*chunk + 4*lane + i)
__device__ __forceinline__ bool node_edge(const QtGeom &q, int l, const ChunkEdges &E, const unsigned char *__restrict__ pyr, int i, int cx, int cy, int lvn)
{
    switch (lvn) {
    case 0: return (E.e0 >> i) & 1u;
    case 1: return E.e1;
    case 2: return E.e2;
    case 3: return E.e3;
    case 4: return E.e4;
    default: {
        const int ncell = q.ncell[l], side = ncell >> lvn;
        return pyr[lvl_off(ncell, lvn) + (long long)(cy >> lvn) * side + (cx >> lvn)] != 0;
    }
    }
}

__device__ __forceinline__ CellNodes eval_cell(const QtGeom &q, int l, int w, int h, const ChunkEdges &E, const unsigned char *__restrict__ pyr, unsigned gidx,
                                               int cx, int cy)
{
    CellNodes r;
    r.nsym = 0; r.syms = 0; r.leaf_lvl = -1;
    const int ltot = q.ltot[l], cell = q.cell, i = (int)(gidx & 3u);
    int a = gidx == 0 ? ltot : min((int)(__ffs((int)gidx) - 1) >> 1, ltot);
    if (a < ltot) {
        // the level-a node exists only if its parent is in bounds and splits
        int mask = ~((2 << a) - 1);
        int pcx = cx & mask, pcy = cy & mask;
        if (pcx * cell >= w || pcy * cell >= h) return r;
        int psize = cell << (a + 1);
        bool psplit = psize > q.bmax;
        if (!psplit && psize > q.bmin) psplit = node_edge(q, l, E, pyr, i, cx, cy, a + 1);
        if (!psplit) return r;
    }
    if (cx * cell >= w || cy * cell >= h) {   // quadtree.py:109-110, 153-155: absent child
        r.nsym = 1; r.syms = 2u;
        return r;
    }
    for (int lvn = a; lvn >= 0; lvn--) {
        int size = cell << lvn;
        bool split = size > q.bmax;
        if (!split && size > q.bmin) split = node_edge(q, l, E, pyr, i, cx, cy, lvn);
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

// The four sibling cells of a lane.  Cell 0 may be the origin of nodes of any level (general walk); cells 1..3 can only
// originate their own level-0 node, which exists iff the lane's level-1 node is in bounds and splits -- the same test
// eval_cell makes, evaluated once for the three of them.
__device__ __forceinline__ void eval_lane(const QtGeom &q, int l, int w, int h, const ChunkEdges &E, const unsigned char *__restrict__ pyr, unsigned chunk,
                                          int lane, long long ncell2, CellNodes (&c)[4], int &cx0, int &cy0)
{
    int ccx, ccy, lx, ly;
    morton_decode(chunk, ccx, ccy);
    morton_decode((unsigned)lane, lx, ly);
    cx0 = ccx * 16 + lx * 2; cy0 = ccy * 16 + ly * 2;
    const unsigned g0 = chunk * 256u + (unsigned)lane * 4u;
#pragma unroll
    for (int i = 0; i < 4; i++) { c[i].nsym = 0; c[i].syms = 0; c[i].leaf_lvl = -1; }
    if ((long long)g0 >= ncell2) return;
    c[0] = eval_cell(q, l, w, h, E, pyr, g0, cx0, cy0);
    const int cell = q.cell;
    bool split1 = 2 * cell > q.bmax;
    if (!split1 && 2 * cell > q.bmin) split1 = E.e1;
    if (!(split1 && cx0 * cell < w && cy0 * cell < h)) return;
#pragma unroll
    for (int i = 1; i < 4; i++) {
        if ((long long)(g0 + i) >= ncell2) continue;
        const int cx = cx0 + (i & 1), cy = cy0 + (i >> 1);
        c[i].nsym = 1;
        if (cx * cell >= w || cy * cell >= h) c[i].syms = 2u;                                           // absent child '10'
        else if (cell > q.bmax || (cell > q.bmin && ((E.e0 >> i) & 1u))) c[i].syms = 1u;                // '01'
        else c[i].leaf_lvl = 0;                                                                         // '00'
    }
}

// The count pass leaves what it found for the emit pass in 12 bits per lane, so the emit pass neither re-reads the edge
// bit-plane nor walks the levels again.  Cell 0 originates k >= 0 split symbols '01' followed by a leaf '00' (type 1), nothing
// (type 2), or it is a single absent child '10' (type 3); cells 1..3 originate at most one symbol (0 none, 1 '00', 2 '01', 3 '10').
__device__ __forceinline__ unsigned pack_lane(const CellNodes (&c)[4])
{
    unsigned code = 0;
    if (c[0].nsym > 0) {
        if (c[0].nsym == 1 && c[0].syms == 2u) code = 3u;
        else {
            const int k = c[0].nsym - (c[0].leaf_lvl >= 0 ? 1 : 0);
            code = (c[0].leaf_lvl >= 0 ? 1u : 2u) | ((unsigned)k << 2);
        }
    }
#pragma unroll
    for (int i = 1; i < 4; i++) {
        const unsigned ci = c[i].nsym == 0 ? 0u : (c[i].syms == 2u ? 3u : c[i].syms == 1u ? 2u : 1u);
        code |= ci << (4 + 2 * i);
    }
    return code;
}

__device__ __forceinline__ void unpack_lane(unsigned code, unsigned g0, int ltot, CellNodes (&c)[4])
{
#pragma unroll
    for (int i = 0; i < 4; i++) { c[i].nsym = 0; c[i].syms = 0; c[i].leaf_lvl = -1; }
    const unsigned t = code & 3u;
    const int k = (int)((code >> 2) & 15u);
    if (t == 3u) { c[0].nsym = 1; c[0].syms = 2u; }
    else if (t != 0u) {
        const int a = g0 == 0 ? ltot : min((int)(__ffs((int)g0) - 1) >> 1, ltot);      // level of the largest node originating here
        c[0].syms = 0x55555555u & ((1u << (2 * k)) - 1u);
        c[0].nsym = k + (t == 1u ? 1 : 0);
        if (t == 1u) c[0].leaf_lvl = a - k;
    }
#pragma unroll
    for (int i = 1; i < 4; i++) {
        const unsigned ci = (code >> (4 + 2 * i)) & 3u;
        if (ci) { c[i].nsym = 1; c[i].syms = ci == 3u ? 2u : ci == 2u ? 1u : 0u; c[i].leaf_lvl = ci == 1u ? 0 : -1; }
    }
}

__device__ __forceinline__ int wave_incl_scan(int v, int /*lane*/) { return wave_scan_incl(v); }      // aej_common.h: DPP, all lanes active
__device__ __forceinline__ int wave_sum(int v) { return wave_total(v); }

// pass 2: per-chunk totals (symbols, leaves, coefficients, leaves per block size); one wave per chunk
__global__ __launch_bounds__(256) void k_qt_count(Geom g, QtGeom q, const unsigned long long *__restrict__ edge_bits,
                                                  const unsigned char *__restrict__ pyr_all, int *__restrict__ chunk_cnt,
                                                  unsigned short *__restrict__ lane_code)
{
    const int b = blockIdx.y;
    int l;
    unsigned chunk;
    if (!locate_chunk_block(q, g.nl, (int)blockIdx.x, l, chunk)) return;
    const int lane = threadIdx.x & 63;
    chunk += threadIdx.x >> 6;
    if ((long long)chunk >= q.nchunk[l]) return;
    const long long ncell2 = (long long)q.ncell[l] * q.ncell[l];
    const unsigned char *pyr = pyr_all + (long long)b * q.pyr_stride + q.pyr_off[l];
    {
        // A chunk that lies wholly outside the plane (about half of the root square) can originate one thing only: the
        // absent-child symbol '10' of a node whose origin is the chunk's first cell.  Only lane 0 has anything to evaluate.
        int ccx, ccy;
        morton_decode(chunk, ccx, ccy);
        if (ccx * 16 * q.cell >= g.w[l] || ccy * 16 * q.cell >= g.h[l]) {
            ChunkEdges Z;
            Z.e0 = 0; Z.e1 = Z.e2 = Z.e3 = Z.e4 = false;
            int ns = 0;
            if (lane == 0 && (long long)chunk * 256 < ncell2) ns = eval_cell(q, l, g.w[l], g.h[l], Z, pyr, chunk * 256u, ccx * 16, ccy * 16).nsym;
            lane_code[((long long)b * q.chunk_stride + q.chunk_off[l] + chunk) * 64 + lane] = (unsigned short)(ns ? 3u : 0u);
            if (lane == 0) {
                int *o = chunk_cnt + ((long long)b * q.chunk_stride + q.chunk_off[l] + chunk) * kChunkInts;
                o[0] = ns; o[1] = 0; o[2] = 0; o[3] = 0;
#pragma unroll
                for (int k = 0; k < kMaxSizes; k++) o[4 + k] = 0;
            }
            return;
        }
    }
    const ChunkEdges E = chunk_edges(g, q, l, b, edge_bits, chunk, lane);
    if (!E.e4 && (q.cell << 4) <= q.bmax) {
        // A chunk without a single edge pixel whose own size does not exceed the maximum block: no node inside it splits, so
        // the only cell that can originate anything is the chunk's first one (the chunk itself as a leaf, or as the origin of a
        // larger node: the general walk, on lane 0 alone).  No sibling tests, no wave reductions -- this is every chunk of a
        // flat region (41 % of the bench planes' area).
        CellNodes z[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { z[i].nsym = 0; z[i].syms = 0; z[i].leaf_lvl = -1; }
        if (lane == 0 && (long long)chunk * 256 < ncell2) {
            int ccx, ccy;
            morton_decode(chunk, ccx, ccy);
            z[0] = eval_cell(q, l, g.w[l], g.h[l], E, pyr, chunk * 256u, ccx * 16, ccy * 16);
        }
        lane_code[((long long)b * q.chunk_stride + q.chunk_off[l] + chunk) * 64 + lane] = (unsigned short)pack_lane(z);
        if (lane == 0) {
            int *o = chunk_cnt + ((long long)b * q.chunk_stride + q.chunk_off[l] + chunk) * kChunkInts;
            const int lv = z[0].leaf_lvl, sz = lv >= 0 ? q.cell << lv : 0;
            o[0] = z[0].nsym; o[1] = lv >= 0 ? 1 : 0; o[2] = sz * sz; o[3] = 0;
#pragma unroll
            for (int k = 0; k < kMaxSizes; k++) o[4 + k] = (k == lv) ? 1 : 0;
        }
        return;
    }
    int nsym = 0, nleaf = 0, ncoef = 0;
    int nsz[kMaxSizes];
#pragma unroll
    for (int k = 0; k < kMaxSizes; k++) nsz[k] = 0;
    CellNodes cn[4];
    int cx0, cy0;
    eval_lane(q, l, g.w[l], g.h[l], E, pyr, chunk, lane, ncell2, cn, cx0, cy0);
    lane_code[((long long)b * q.chunk_stride + q.chunk_off[l] + chunk) * 64 + lane] = (unsigned short)pack_lane(cn);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const CellNodes &c = cn[i];
        nsym += c.nsym;
        if (c.leaf_lvl >= 0) {
            nleaf++;
            int s = q.cell << c.leaf_lvl;
            ncoef += s * s;
#pragma unroll
            for (int k = 0; k < kMaxSizes; k++) nsz[k] += (c.leaf_lvl == k) ? 1 : 0;
        }
    }
    // wave totals: the symbol count (< 1024 per chunk) and the per-size leaf counts (<= 256) travel three to a word through the
    // butterfly; leaves and coefficients follow from the per-size counts
    static_assert(kMaxSizes <= 8, "three packed words hold the symbol count and 8 sizes");
    unsigned pk[3] = { (unsigned)nsym, 0u, 0u };
#pragma unroll
    for (int k = 0; k < kMaxSizes; k++) pk[(k + 1) / 3] |= (unsigned)nsz[k] << (10 * ((k + 1) % 3));
#pragma unroll
    for (int i = 0; i < 3; i++) pk[i] = (unsigned)wave_sum((int)pk[i]);
    nsym = (int)(pk[0] & 1023u);
    nleaf = 0; ncoef = 0;
#pragma unroll
    for (int k = 0; k < kMaxSizes; k++) {
        nsz[k] = k < q.nsizes ? (int)((pk[(k + 1) / 3] >> (10 * ((k + 1) % 3))) & 1023u) : 0;
        const int sz = q.cell << k;
        nleaf += nsz[k];
        ncoef += nsz[k] * sz * sz;
    }
    if (lane == 0) {
        int *o = chunk_cnt + ((long long)b * q.chunk_stride + q.chunk_off[l] + chunk) * kChunkInts;
        o[0] = nsym; o[1] = nleaf; o[2] = ncoef; o[3] = 0;
#pragma unroll
        for (int k = 0; k < kMaxSizes; k++) o[4 + k] = nsz[k];
    }
}

// pass 3: exclusive scan of the chunk records per (image, layer); totals -> counts and per-plane work counts.
// 256 threads (round 3; was 1024): a workgroup of 16 waves needs four free wave slots with 56 registers on EVERY SIMD of one CU, which
// in the pipelined path it waited for behind the other chains' resident workgroups (211 us per launch under overlap against 45 us alone);
// four waves fit into the gaps.  A call of a few images has nothing to wait behind and keeps the 1024-thread shape (fewer serial
// rounds: it is latency there).
template <int kScanThreads>
__global__ __launch_bounds__(kScanThreads) void k_qt_scan(Geom g, QtGeom q, int *__restrict__ chunk_cnt, long long *__restrict__ counts,
                                                  int *__restrict__ work_count)
{
    constexpr int NQ = 3 + kMaxSizes;
    __shared__ int s_w[NQ][kScanThreads / 64];
    __shared__ int carry[NQ];
    const int tid = threadIdx.x, l = blockIdx.x, b = blockIdx.y;
    const int lane = tid & 63, wv = tid >> 6;
    int *base = chunk_cnt + ((long long)b * q.chunk_stride + q.chunk_off[l]) * kChunkInts;
    const int n = q.nchunk[l];
    const int nq = 3 + q.nsizes;
    if (tid < NQ) carry[tid] = 0;
    __syncthreads();
    static_assert(kChunkInts == 12 && NQ == 11, "a chunk record is three int4: (nsym, nleaf, ncoef, pad), leaves per size 0..3, 4..7");
    int4 *rec = reinterpret_cast<int4 *>(base);          // (chunk records start on 16-byte boundaries: 48 bytes each, workspace carved at 256)
    for (int start = 0; start < n; start += kScanThreads) {
        const int i = start + tid;
        // the record as three 16-byte loads (a wave reads 3 KiB contiguously) instead of eleven strided dword loads
        int4 r0 = make_int4(0, 0, 0, 0), r1 = r0, r2 = r0;
        if (i < n) { r0 = rec[3 * i]; r1 = rec[3 * i + 1]; r2 = rec[3 * i + 2]; }
        int v[NQ] = { r0.x, r0.y, r0.z, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w }, inc[NQ];
#pragma unroll
        for (int c = 0; c < NQ; c++) {
            if (c >= nq) v[c] = 0;
            inc[c] = wave_incl_scan(v[c], lane);
            if (lane == 63) s_w[c][wv] = inc[c];
        }
        __syncthreads();
        int o[NQ];
#pragma unroll
        for (int c = 0; c < NQ; c++) {
            int p = carry[c];
            for (int k = 0; k < wv; k++) p += s_w[c][k];
            o[c] = p + inc[c] - v[c];
        }
        if (i < n) {
            rec[3 * i] = make_int4(o[0], o[1], o[2], r0.w);
            rec[3 * i + 1] = make_int4(o[3], o[4], o[5], o[6]);
            rec[3 * i + 2] = make_int4(o[7], o[8], o[9], o[10]);
        }
        __syncthreads();
        if (tid < NQ) {
            int t = 0;
            for (int k = 0; k < kScanThreads / 64; k++) t += s_w[tid][k];
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

// pass 4: emit symbols, leaf table and the per-size DCT work lists; one wave per chunk.  Positions come from the
// scans (deterministic, Morton-ordered lists, no global atomics).
__global__ __launch_bounds__(256) void k_qt_emit(Geom g, QtGeom q, QtBuffers qb)
{
    const int b = blockIdx.y;
    int l;
    unsigned chunk;
    if (!locate_chunk_block(q, g.nl, (int)blockIdx.x, l, chunk)) return;
    const int lane = threadIdx.x & 63;
    chunk += threadIdx.x >> 6;
    if ((long long)chunk >= q.nchunk[l]) return;
    const int *coff = qb.chunk_cnt + ((long long)b * q.chunk_stride + q.chunk_off[l] + chunk) * kChunkInts;
    CellNodes c[4];
    const unsigned code = qb.lane_code[((long long)b * q.chunk_stride + q.chunk_off[l] + chunk) * 64 + lane];
    const unsigned long long originators = __ballot(code != 0);
    if (originators == 0) return;         // nothing originates in this chunk (about half of the root square lies outside the plane)
    unpack_lane(code, chunk * 256u + (unsigned)lane * 4u, q.ltot[l], c);
    int ccx, ccy, lx, ly;
    morton_decode(chunk, ccx, ccy);
    morton_decode((unsigned)lane, lx, ly);
    const int cx0 = ccx * 16 + lx * 2, cy0 = ccy * 16 + ly * 2;
    int nsym = 0, nleaf = 0, ncoef = 0, n0 = 0;   // n0: level-0 leaves of this lane (a lane has at most one larger leaf, at cell 0)
#pragma unroll
    for (int i = 0; i < 4; i++) {
        nsym += c[i].nsym;
        if (c[i].leaf_lvl >= 0) { nleaf++; int s = q.cell << c[i].leaf_lvl; ncoef += s * s; }
        if (c[i].leaf_lvl == 0) n0++;
    }
    int sym_pos = coff[0], leaf_pos = coff[1], rank0 = 0, coef_pos = coff[2], rank_big = 0;
    if (originators != 1ull) {            // (lane 0 alone -- a chunk without edges, k_qt_count's short path -- starts at the chunk's offsets)
        // one scan for the three small counters (prefix sums < 1024 each), one for the coefficient offsets
        const int pk = nsym | nleaf << 10 | n0 << 20;
        const int pks = wave_incl_scan(pk, lane) - pk;
        sym_pos += pks & 1023;
        leaf_pos += (pks >> 10) & 1023;
        rank0 = (pks >> 20) & 1023;
        coef_pos += wave_incl_scan(ncoef, lane) - ncoef;
        const int big = c[0].leaf_lvl;                     // > 0 when this lane holds a leaf larger than a cell
        for (int k = 1; k < q.nsizes; k++) {
            unsigned long long m = __ballot(big == k);
            if (big == k) rank_big = __popcll(m & ((1ull << lane) - 1ull));
        }
    }

    unsigned char *st = qb.states + (long long)b * q.state_stride + q.state_off[l];
    int *leaves = qb.leaves + ((long long)b * q.leaf_stride + q.leaf_off[l]) * 4;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        for (int k = 0; k < c[i].nsym; k++) {
            if (sym_pos + k < q.state_cap[l]) st[sym_pos + k] = (unsigned char)((c[i].syms >> (2 * k)) & 3u);
            else *qb.overflow = 1;
        }
        sym_pos += c[i].nsym;
        if (c[i].leaf_lvl >= 0) {
            const int cx = cx0 + (i & 1), cy = cy0 + (i >> 1);
            const int size = q.cell << c[i].leaf_lvl;
            if (leaf_pos < q.leaf_cap[l] && (long long)coef_pos + (long long)size * size <= q.coeff_cap[l]) {
                reinterpret_cast<int4 *>(leaves)[leaf_pos] = make_int4(cx * q.cell, cy * q.cell, size, coef_pos);
                if (qb.work_count) {
                    const int k = c[i].leaf_lvl;       // size == bmin << k (the codec path always has cell == bmin)
                    long long pos = (long long)coff[4 + k] + (k == 0 ? rank0 : rank_big);
                    long long seg = (long long)b * q.work_stride[k] + q.work_off[l][k];
                    if (seg + pos < qb.work_cap[k])
                        qb.work[k][seg + pos] = pack_work(cx * q.cell, cy * q.cell, coef_pos);
                    else
                        *qb.overflow = 1;
                    if (k == 0) rank0++;
                }
            } else {
                *qb.overflow = 1;
            }
            leaf_pos++;
            coef_pos += size * size;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static int chunk_blocks(const Geom &g, const QtGeom &q)      // workgroups of 4 chunks (one wave each), all layers of one image
{
    int n = 0;
    for (int l = 0; l < g.nl; l++) n += (q.nchunk[l] + 3) / 4;
    return n;
}

void launch_qt_cells(hipStream_t st, const Geom &g, const QtGeom &q, const unsigned long long *edge_bits, const QtBuffers &qb)
{
    // only the levels above a chunk (>= 5) live in global memory, and only when a node of that size can still be a leaf
    bool need = false;
    for (int l = 0; l < g.nl; l++) if (q.ltot[l] > 4 && (q.cell << 5) <= q.bmax) need = true;
    if (need) hipLaunchKernelGGL(k_qt_upper, dim3(chunk_blocks(g, q), g.B), dim3(256), 0, st, g, q, edge_bits, qb.pyr);
}
void launch_qt_count(hipStream_t st, const Geom &g, const QtGeom &q, const QtBuffers &qb)
{
    hipLaunchKernelGGL(k_qt_count, dim3(chunk_blocks(g, q), g.B), dim3(256), 0, st, g, q, qb.edge_bits, qb.pyr, qb.chunk_cnt, qb.lane_code);
}
void launch_qt_scan(hipStream_t st, const Geom &g, const QtGeom &q, const QtBuffers &qb)
{
    if (g.B <= 4) hipLaunchKernelGGL(k_qt_scan<1024>, dim3(g.nl, g.B), dim3(1024), 0, st, g, q, qb.chunk_cnt, qb.counts, qb.work_count);
    else hipLaunchKernelGGL(k_qt_scan<256>, dim3(g.nl, g.B), dim3(256), 0, st, g, q, qb.chunk_cnt, qb.counts, qb.work_count);
}
void launch_qt_emit(hipStream_t st, const Geom &g, const QtGeom &q, const QtBuffers &qb)
{
    hipLaunchKernelGGL(k_qt_emit, dim3(chunk_blocks(g, q), g.B), dim3(256), 0, st, g, q, qb);
}

}  // namespace aej
