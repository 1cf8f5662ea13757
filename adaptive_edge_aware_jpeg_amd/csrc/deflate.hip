// deflate.hip -- OPT-IN GPU entropy stage for the .ajpg container: every layer's int32 coefficient array as a zlib stream
// (RFC 1950 / 1951) that the reference's decoder reads with zlib.decompress (src/jpeg/jpeg.py:659).  The reference writes these streams
// with zlib.compress(level=9) on the host (jpeg.py:588-590); with the hot path on the GPU that call IS Jpeg.compress end to end
// (98 % of compress_many on natural 4K images, 1.4 MP/s per host core: profiles/r04_bench_extra_natural.json).  Never the default: the
// default container stays byte-identical to the reference's.
//
// Round 5: a real LZ77 matcher.  Round 4 only looked one byte and one coefficient back (distances 1 and 4) and came out 1.34 x zlib
// level 9's size; zlib's edge on these streams is long matches at arbitrary distances (level 9 covers 97 % of the bytes of a natural
// luma layer with matches of 24 bytes on average, two thirds of them more than 1 KiB back).
//
// Format.  ONE deflate block per stream (BFINAL = 1): dynamic Huffman (BTYPE = 10) with a code the HOST builds once per layer of a batch
// from the token histogram (aej_deflate_build_tables), or RFC 1951's fixed code (BTYPE = 01) when no table is given, when the table lacks
// a code the stream needs, or when the fixed code is smaller.  The stream is produced in chunks of 32 KiB of input = 8192 coefficients:
//   k_lz_parse   one workgroup per chunk, the chunk staged in LDS.  Positions are coefficient (dword) positions -- every useful match of
//                an int32 array starts on one.  (1) All positions are sorted by (hash of [upper three bytes of coefficient i,
//                coefficient i + 1], position) with a bitonic sort: the predecessors of a position in the sorted order that share its
//                hash ARE its hash chain, exact and newest first, with no atomics and no dependence on timing.  (2) Every position walks
//                up to kLzDepth chain entries and the kLzNear nearest positions and keeps two candidates: the best match that starts AT
//                the coefficient ("A": 4 L bytes) and the best that starts one byte in, after a literal low byte ("B": 3 + 4 L bytes --
//                a small coefficient followed by the pattern that followed another small coefficient; zero runs behind a non-zero value).
//                (3) A thread per 64-coefficient sub-block chooses the tokens by dynamic programming over a static bit-cost model (four
//                literals / literal + B match / A match; matches are clipped to the sub-block, so sub-blocks are independent), writes one
//                token word per coefficient to the workspace and counts the symbols.  Sources never leave the chunk (window <= 32 KiB).
//   k_lz_sizes   bits of every sub-block and chunk under both codes, Adler-32 partial sums.
//   k_lz_scan    per stream: which code, the bit offset of every chunk, zlib header, Adler-32, stream size.
//   k_lz_emit    the chunks' bit strings, assembled in LDS and written at their BIT offset (boundary words by atomicOr).
// Everything is deterministic: the same input gives the same bytes.
#include "aej_common.h"
#include "aej_launch.h"
#include <string.h>

#include <algorithm>
#include <vector>

namespace aej {

constexpr int kLzChunkDw = 8192;                    // coefficients per chunk
constexpr int kLzChunkBytes = 4 * kLzChunkDw;       // 32 KiB of input
constexpr int kLzThreads = 1024;                    // parse kernel: one workgroup per CU (its LDS), sixteen waves to hide the LDS latency of the chain walks
constexpr int kLzSubDw = 64;                        // coefficients per sub-block (tokens never cross one)
constexpr int kLzSubs = kLzChunkDw / kLzSubDw;      // 128 sub-blocks per chunk
constexpr int kLzNear = 8;                          // distances (in coefficients) always tried
// Search effort, measured on eight natural 4K images (profiles/r05_deflate_gpu.txt; zlib level 9 = 1.000, level 6 = 1.093):
//   chain entries tried per position (a multiple of 64: a wave's lanes take 64 at a time)     128: 1.085 of zlib-9's bytes, 41 ms    64: 1.096, 32 ms
//   a near match of kLzNice coefficients is good enough, the position's chain is not walked   64 (= never): 32 ms   8: 28 ms   4: 26 ms, 1.100   2: 1.104, 18 ms
constexpr int kLzDepth = 64;
constexpr int kLzNice = 2;
static_assert(4 * kLzSubDw <= 258, "a match stays inside its sub-block, so 256 bytes is the longest (deflate allows 258)");
constexpr int kLzHashBits = 13;
constexpr unsigned kAdlerMod = 65521u;
constexpr int kDefLitLen = 286, kDefDist = 30;
constexpr int kDefHistBins = 320;                   // AEJ_DEFLATE_HIST_BINS: 286 literal / length symbols, then 30 distance symbols (4 unused)
constexpr int kDefHdrWords = 131;
constexpr int kDefTableWords = 448;                 // AEJ_DEFLATE_TABLE_WORDS: [0..285] literal / length codes, [286..315] distance codes,
                                                    // [316] header bits, [317..447] the header's bits (BFINAL = 1, BTYPE = 10, the code lengths)
constexpr int kDefTabHdrBits = kDefLitLen + kDefDist, kDefTabHdr = kDefTabHdrBits + 1;
static_assert(kDefTabHdr + kDefHdrWords == kDefTableWords, "table layout");

// token word of one coefficient position: kind | L << 2 | d << 9   (L in coefficients, d = distance in coefficients, 1 .. 8191)
//   kind 0: four literals            kind 1: literal (low byte) + match of 3 + 4 L bytes at distance 4 d
//   kind 2: match of 4 L bytes at distance 4 d (L >= 1)      kind 3: covered by the match of an earlier position
__device__ __forceinline__ unsigned lz_token(int kind, int L, int d) { return (unsigned)kind | ((unsigned)L << 2) | ((unsigned)d << 9); }

struct LzStreams {
    const int *coeffs;            // [B][coeff_stride]
    const long long *counts;      // [B][3][4]: n_coeffs first
    long long coeff_stride, coeff_off[3], coeff_cap[3];
    int chunk_off[4];             // first chunk slot of layer l inside one image's slots; [3] = slots per image
    unsigned *tokens;             // [B * slots][kLzChunkDw]
    unsigned *chunk_bits;         // [B * slots][2] bits of the chunk's tokens under the fixed / the table's code
    unsigned *chunk_adler;        // [B * slots][2] sum of bytes, sum of (len - i) * byte, both mod 65521
    unsigned char *chunk_missing; // [B * slots] a token of the chunk has no code in the table
    unsigned short *sub_bits;     // [B * slots][kLzSubs][2]
    unsigned long long *chunk_pos;// [B * slots] bit offset of the chunk's tokens inside its stream
    unsigned char *stream_fixed;  // [B * 3] 1 = the stream's block uses the fixed code
    const unsigned *tables;       // [3][kDefTableWords] per-layer dynamic codes, or null: fixed code everywhere
    unsigned char *out;           // [B * 3][stream_stride]
    unsigned long long stream_stride;
    long long *sizes;             // [B * 3] bytes of each finished stream
    int *hist;                    // [3][kDefHistBins] (parse only; may be null)
    int *error;                   // [1] 1: a stream does not fit its slot; 2: a count exceeds its layer's capacity
};

// length symbol and extra bits of a match of L bytes (3 .. 258); distance symbol and extra bits of a distance of D bytes (1 .. 32768)
__device__ __forceinline__ int lz_len_sym(int L, int &ebits, unsigned &extra)
{
    const int l = L - 3;
    if (l < 8) { ebits = 0; extra = 0; return 257 + l; }
    if (L == 258) { ebits = 0; extra = 0; return 285; }
    const int e = 29 - __clz(l);
    ebits = e; extra = (unsigned)l & ((1u << e) - 1u);
    return 257 + 4 * (e + 1) + ((l >> e) & 3);
}
__device__ __forceinline__ int lz_dist_sym(int D, int &ebits, unsigned &extra)
{
    const int x = D - 1;
    if (x < 4) { ebits = 0; extra = 0; return x; }
    const int e = 30 - __clz(x);
    ebits = e; extra = (unsigned)x & ((1u << e) - 1u);
    return 2 * (e + 1) + ((x >> e) & 1);
}

// ---- static bit-cost model of the parser, in eighths of a bit (what a typical coefficient stream's dynamic code charges) ----
__device__ __forceinline__ int lz_lit_cost(unsigned b) { return b == 0u ? 14 : b == 255u ? 24 : (b < 4u || b > 252u) ? 36 : (b < 16u || b > 240u) ? 52 : 72; }
__device__ __forceinline__ int lz_lit4_cost(unsigned w) { return lz_lit_cost(w & 255u) + lz_lit_cost((w >> 8) & 255u) + lz_lit_cost((w >> 16) & 255u) + lz_lit_cost(w >> 24); }
__device__ __forceinline__ int lz_len_cost(int L) { const int l = L - 3; return 32 + (l < 8 ? 0 : 8 * (29 - __clz(l))); }
__device__ __forceinline__ int lz_dist_cost(int D) { const int x = D - 1; if (x < 2) return 16; if (x < 4) return 28; const int e = 30 - __clz(x); return (e <= 1 ? 28 : 40) + 8 * e; }

__device__ __forceinline__ unsigned lz_hash(unsigned w0, unsigned w1) { return (((w0 >> 8) * 2654435761u) ^ (w1 * 2246822519u)) >> (32 - kLzHashBits); }

__device__ __forceinline__ const unsigned *lz_stream_of(const LzStreams &S, int s, long long &n_dw, bool *bad = nullptr)
{
    const int b = s / 3, l = s - 3 * b;
    long long n = S.counts[(long long)s * 4];
    if (n < 0 || n > S.coeff_cap[l]) { if (bad) *bad = true; n = n < 0 ? 0 : S.coeff_cap[l]; }      // a corrupt / stale counts buffer must not send the kernels out of the layer's slot
    n_dw = n;
    return reinterpret_cast<const unsigned *>(S.coeffs + (long long)b * S.coeff_stride + S.coeff_off[l]);
}
// chunk slot -> (layer, chunk of the layer)
__device__ __forceinline__ void lz_locate(const LzStreams &S, int slot, int &l, int &c)
{
    l = slot >= S.chunk_off[2] ? 2 : slot >= S.chunk_off[1] ? 1 : 0;
    c = slot - S.chunk_off[l];
}

// stage a chunk's coefficients in LDS (16-byte loads; zeros beyond the chunk's end)
__device__ __forceinline__ void lz_stage(const unsigned *src, long long c0, int len, unsigned *sIn /* [kLzChunkDw + 4] */)
{
    // coefficient arrays start on 256-byte boundaries and their capacities are multiples of 64 bytes: the loads are aligned and stay inside the layer's slot
    for (int i = threadIdx.x * 4; i < kLzChunkDw; i += kLzThreads * 4) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (i < len) v = *reinterpret_cast<const uint4 *>(src + c0 + i);
        if (i + 1 >= len) v.y = 0u;
        if (i + 2 >= len) v.z = 0u;
        if (i + 3 >= len) v.w = 0u;
        *reinterpret_cast<uint4 *>(sIn + i) = v;
    }
    if (threadIdx.x < 4) sIn[kLzChunkDw + threadIdx.x] = 0u;
}

struct LzLds {
    unsigned in[kLzChunkDw + 4];
    unsigned keys[kLzChunkDw];           // sort keys (hash << 13 | position); afterwards the DP's cost rows
    unsigned resA[kLzChunkDw];           // best A candidate: L | d << 7; bits 30-31: the DP's choice
    unsigned resB[kLzChunkDw];           // best B candidate: L | d << 7 | valid << 20
    unsigned short todo[kLzChunkDw];     // sorted-order indices of the positions whose chain is worth walking
    int hist[kDefHistBins];
    int ntodo, ticket;
};
static_assert(sizeof(LzLds) <= 160 * 1024, "one workgroup per CU");

// Pass 1: match search + parse of one chunk; tokens to the workspace, symbol counts to hist.
__global__ __launch_bounds__(kLzThreads) void k_lz_parse(LzStreams S)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lz_smem[];
    LzLds &L = *reinterpret_cast<LzLds *>(lz_smem);
    const int slot = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    int l, c;
    lz_locate(S, slot, l, c);
    const int s = b * 3 + l;
    long long n_dw;
    bool bad = false;
    const unsigned *src = lz_stream_of(S, s, n_dw, &bad);
    if (bad && tid == 0 && c == 0) atomicMax(S.error, 2);
    const long long c0 = (long long)c * kLzChunkDw;
    if (c0 >= n_dw) return;
    const int len = (int)(n_dw - c0 < kLzChunkDw ? n_dw - c0 : kLzChunkDw);
    lz_stage(src, c0, len, L.in);
    for (int i = tid; i < kDefHistBins; i += kLzThreads) L.hist[i] = 0;
    __syncthreads();
    // ---- (1) sort keys: positions past the end sort last
    for (int i = tid; i < kLzChunkDw; i += kLzThreads)
        L.keys[i] = i < len ? (lz_hash(L.in[i], L.in[i + 1]) << 13) | (unsigned)i : 0xffffffffu;
    __syncthreads();
    // Bitonic sort.  A thread owns eight consecutive keys: the passes of strides 4, 2, 1 of a stage exchange keys inside such a group and run
    // in registers (one LDS round trip and one barrier instead of three of each); the wider strides go through the LDS, a pass each.
    static_assert(kLzChunkDw == 8 * kLzThreads, "eight keys per thread");
    auto local_passes = [&](int k, int jmax) {                             // strides jmax, jmax / 2, ... 1 (jmax <= 4) of stage k
        unsigned v[8];
        const uint4 lo = *reinterpret_cast<const uint4 *>(&L.keys[8 * tid]), hi = *reinterpret_cast<const uint4 *>(&L.keys[8 * tid + 4]);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
#pragma unroll
        for (int j = 4; j > 0; j >>= 1) {
            if (j <= jmax) {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    if ((e & j) == 0) {
                        const bool up = ((8 * tid + e) & k) == 0;
                        const unsigned a = v[e], c2 = v[e | j];
                        const bool sw = (a > c2) == up;
                        v[e] = sw ? c2 : a; v[e | j] = sw ? a : c2;
                    }
                }
            }
        }
        *reinterpret_cast<uint4 *>(&L.keys[8 * tid]) = make_uint4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<uint4 *>(&L.keys[8 * tid + 4]) = make_uint4(v[4], v[5], v[6], v[7]);
    };
    for (int k = 2; k <= kLzChunkDw; k <<= 1) {
        int j = k >> 1;
        for (; j > 4; j >>= 1) {
            for (int t = tid; t < kLzChunkDw / 2; t += kLzThreads) {
                const int i = 2 * t - (t & (j - 1));           // the lower index of pair t for stride j
                const int p = i + j;
                const unsigned a = L.keys[i], bkey = L.keys[p];
                const bool up = (i & k) == 0;
                if ((a > bkey) == up) { L.keys[i] = bkey; L.keys[p] = a; }
            }
            __syncthreads();
        }
        local_passes(k, j);
        __syncthreads();
    }
    // ---- (2a) the kLzNear nearest positions.  A wave per sub-block, a lane per position: bit j of eq[d] (a ballot) says "coefficient j equals
    // the one d before it", so the run of matches behind a position is a shift and a count-trailing-ones -- no walk (inside a run of zeros
    // every walk would be as long as the rest of the sub-block).  Distances ascend, so at equal length the nearer one stays.
    {
        const int lane = tid & 63, wave = tid >> 6;
        for (int sb = wave; sb * kLzSubDw < len; sb += kLzThreads / 64) {
            const int i = sb * kLzSubDw + lane;
            const bool valid = i < len;
            const unsigned w0 = L.in[i];                                   // (zeros beyond the chunk's end)
            int bestA = 0, dA = 0, bestB = -1, dB = 0, sA = 0, sB = -(1 << 30);
#pragma unroll
            for (int d = 1; d <= kLzNear; d++) {
                const unsigned wq = i >= d ? L.in[i - d] : ~w0;
                const unsigned long long eq = __ballot(valid && w0 == wq);
                // coefficients from i + 1 on that equal the one d before them; the sub-block (and the chunk) ends the run: bits beyond are 0
                const unsigned long long behind = lane < 63 ? ~(eq >> (lane + 1)) : ~0ull;
                const int m = behind ? __builtin_ctzll(behind) : 64;
                if (valid && i >= d && ((w0 ^ wq) >> 8) == 0u) {
                    const int dc = lz_dist_cost(4 * d);
                    if (w0 == wq) { const int sc = (4 + 4 * m) * 16 - dc; if (sc > sA) { sA = sc; bestA = 1 + m; dA = d; } }
                    const int sc = (3 + 4 * m) * 16 - dc;
                    if (sc > sB) { sB = sc; bestB = m; dB = d; }
                }
            }
            if (valid) {
                L.resA[i] = (unsigned)bestA | ((unsigned)dA << 7);
                L.resB[i] = bestB >= 0 ? ((unsigned)bestB | ((unsigned)dB << 7) | (1u << 20)) : 0u;
            }
        }
    }
    __syncthreads();
    // ---- (2b) the hash chain of every position (all threads, a position each): up to kLzDepth earlier positions with the same key, nearest first
    // ---- (2b) which positions walk their chain at all: not those whose near match already fills the sub-block (the inside of every run of
    // zeros: two thirds of a natural layer), compacted into a list so that no wave spends a dependent LDS round trip on finding that out
    if (tid == 0) { L.ntodo = 0; L.ticket = kLzThreads / 64; }
    __syncthreads();
    for (int ks = tid; ks < kLzChunkDw; ks += kLzThreads) {
        const unsigned key = L.keys[ks];
        bool want = false;
        if (key != 0xffffffffu) {
            const int i = (int)(key & 8191u);
            int maxl = len - i;
            const int to_sub_end = kLzSubDw - (i & (kLzSubDw - 1));
            maxl = maxl < to_sub_end ? maxl : to_sub_end;
            want = maxl >= 2 && (int)(L.resA[i] & 127u) < (maxl < kLzNice ? maxl : kLzNice);
        }
        const unsigned long long m = __ballot(want);
        int base = 0;
        if ((tid & 63) == 0 && m) base = atomicAdd(&L.ntodo, __builtin_popcountll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (want) L.todo[base + __builtin_popcountll(m & ((1ull << (tid & 63)) - 1ull))] = (unsigned short)ks;
    }
    __syncthreads();
    // A WAVE takes one position at a time (in sorted order) and its 64 lanes take 64 chain entries: the walks of all candidates run side by
    // side and end with the longest one -- which is the length wanted -- instead of one after the other in a lane whose 63 neighbours wait
    // for it (a thread per position spent 82 % of the kernel there: the longest walk among 64 unrelated positions, 96 times over).
    {
        const int lane = tid & 63, wave = tid >> 6;
        // work is handed out by ticket (the cost of a position varies by two orders of magnitude); the next ticket's position is fetched
        // while this one is worked on
        const int ntodo = L.ntodo;
        int n_cur = wave;
        int ks_next = n_cur < ntodo ? (int)L.todo[n_cur] : 0;
        while (n_cur < ntodo) {
            const int ks = ks_next;
            const unsigned mykey = L.keys[ks];                             // (wave-uniform: every lane reads the same word)
            int n_next = 0;
            if (lane == 0) n_next = atomicAdd(&L.ticket, 1);
            const int i = (int)(mykey & 8191u);
            int maxl = len - i;
            const int to_sub_end = kLzSubDw - (i & (kLzSubDw - 1));
            maxl = maxl < to_sub_end ? maxl : to_sub_end;
            const unsigned ra = L.resA[i], rb = L.resB[i];
            int bestA = (int)(ra & 127u), dA = (int)((ra >> 7) & 8191u), bestB = (rb >> 20) ? (int)(rb & 127u) : -1, dB = (int)((rb >> 7) & 8191u);
            n_cur = __builtin_amdgcn_readfirstlane(n_next);
            ks_next = n_cur < ntodo ? (int)L.todo[n_cur] : 0;
            const unsigned w0 = L.in[i], h = mykey >> 13;
            // the coefficients a match would cover, one per lane: what the candidates are compared with comes from here by v_readlane
            // (the LDS is what this phase is bound by: sixteen waves gathering from one array)
            const unsigned tgt = L.in[i + 1 + lane < kLzChunkDw + 4 ? i + 1 + lane : kLzChunkDw + 3];
            int sA = bestA ? 64 * bestA - lz_dist_cost(4 * dA) : 0, sB = bestB >= 0 ? (3 + 4 * bestB) * 16 - lz_dist_cost(4 * dB) : -(1 << 30);
            bool changed = false;
#pragma unroll 1
            for (int base = ks - 1, round = 0; base >= 0 && round < kLzDepth / 64; base -= 64, round++) {
                const int j = base - lane;
                const unsigned key = j >= 0 ? L.keys[j] : 0xffffffffu;
                const bool member = (key >> 13) == h;                      // the chain: the entries before ks with the same hash, nearest first
                const int q = (int)(key & 8191u);
                bool cand = member && i - q > kLzNear;
                const unsigned wq = cand ? L.in[q] : ~w0;
                cand = cand && ((w0 ^ wq) >> 8) == 0u;
                // nearest first: a candidate that does not get FURTHER than the best so far cannot win -- look at the coefficient it would
                // have to match before walking up to it (an A match of the best B's length may still be missing: a_open)
                const bool a_open = w0 == wq && bestA <= bestB;
                if (bestB + 1 >= maxl) cand = cand && a_open;
                else if (bestB >= 0) cand = cand && (a_open || L.in[q + 1 + bestB] == (unsigned)__builtin_amdgcn_readlane((int)tgt, bestB));
                if (__any(cand)) {
                    int m = 0;
                    bool alive = cand;
                    // four coefficients per step (the walk is a chain of dependent LDS reads: fewer, wider steps)
#pragma unroll 1
                    for (int sft = 1; sft < maxl; sft += 4) {
                        if (!__any(alive)) break;
                        // (sft + 2 <= 65: the lane index wraps for the last, masked-off comparisons)
                        const unsigned a0 = (unsigned)__builtin_amdgcn_readlane((int)tgt, sft - 1), a1 = (unsigned)__builtin_amdgcn_readlane((int)tgt, sft & 63),
                                       a2 = (unsigned)__builtin_amdgcn_readlane((int)tgt, (sft + 1) & 63), a3 = (unsigned)__builtin_amdgcn_readlane((int)tgt, (sft + 2) & 63);
                        const int qs = alive ? q + sft : i + sft;          // (dead lanes read one common address: a broadcast)
                        const unsigned b0 = L.in[qs], b1 = L.in[qs + 1], b2 = L.in[qs + 2], b3 = L.in[qs + 3];
                        const bool e0 = alive && a0 == b0;
                        const bool e1 = e0 && a1 == b1 && sft + 1 < maxl;
                        const bool e2 = e1 && a2 == b2 && sft + 2 < maxl;
                        const bool e3 = e2 && a3 == b3 && sft + 3 < maxl;
                        m += (e0 ? 1 : 0) + (e1 ? 1 : 0) + (e2 ? 1 : 0) + (e3 ? 1 : 0);
                        alive = e3;
                    }
                    const int dc = lz_dist_cost(4 * (i - q));
                    const int scA = cand && w0 == wq ? (4 + 4 * m) * 16 - dc : -(1 << 30), scB = cand ? (3 + 4 * m) * 16 - dc : -(1 << 30);
                    // best of the 64 lanes, only when some lane beats what is known: score in the high bits, the nearer entry (the lower lane) wins ties
                    if (__any(scA > sA)) {
                        const unsigned ta = wave_max_u32(scA > sA ? ((unsigned)(scA + 256) << 6) | (unsigned)(63 - lane) : 0u);
                        const int win = 63 - (int)(ta & 63u);
                        sA = (int)(ta >> 6) - 256; bestA = 1 + __builtin_amdgcn_readlane(m, win); dA = i - __builtin_amdgcn_readlane(q, win); changed = true;
                    }
                    if (__any(scB > sB)) {
                        const unsigned tb = wave_max_u32(scB > sB ? ((unsigned)(scB + 256) << 6) | (unsigned)(63 - lane) : 0u);
                        const int win = 63 - (int)(tb & 63u);
                        sB = (int)(tb >> 6) - 256; bestB = __builtin_amdgcn_readlane(m, win); dB = i - __builtin_amdgcn_readlane(q, win); changed = true;
                    }
                    if (bestA >= maxl || bestA >= kLzNice) break;
                }
                if (!__all(member)) break;                                  // the chain ended inside this round
            }
            if (changed && lane == 0) {
                L.resA[i] = (unsigned)bestA | ((unsigned)dA << 7);
                L.resB[i] = bestB >= 0 ? ((unsigned)bestB | ((unsigned)dB << 7) | (1u << 20)) : 0u;
            }
        }
    }
    __syncthreads();
    // ---- (3) parse: one thread per sub-block, backward dynamic programming over the static costs, then the tokens front to back
    unsigned short *cost = reinterpret_cast<unsigned short *>(L.keys);      // [kLzSubs][kLzSubDw + 2] (the keys are no longer needed)
    if (tid < kLzSubs) {
        const int s0 = tid * kLzSubDw;
        const int sn = len - s0 < kLzSubDw ? len - s0 : kLzSubDw;
        if (sn > 0) {
            unsigned short *cs = cost + tid * (kLzSubDw + 2);
            cs[sn] = 0;
            // (the candidates of the next position are fetched before this position's cost look-ups: what an iteration waits for is then
            // one LDS round trip, not two in a row -- only two of the sixteen waves work here)
            unsigned w_n = L.in[s0 + sn - 1], rb_n = L.resB[s0 + sn - 1], ra_n = L.resA[s0 + sn - 1];
            int c_next = 0;                                                  // cs[j + 1], carried in a register
            for (int j = sn - 1; j >= 0; j--) {
                const int i = s0 + j;
                const unsigned w = w_n, rb = rb_n, ra = ra_n;
                if (j > 0) { w_n = L.in[i - 1]; rb_n = L.resB[i - 1]; ra_n = L.resA[i - 1]; }
                int L2 = (int)(rb & 127u), La = (int)(ra & 127u);
                L2 = L2 > sn - j - 1 ? sn - j - 1 : L2;
                La = La > sn - j ? sn - j : La;
                const int cs_b = (rb >> 20) ? cs[j + 1 + L2] : 0, cs_a = La ? cs[j + La] : 0;
                int best = lz_lit4_cost(w) + c_next, ch = 0;
                if (rb >> 20) {
                    const int cb = lz_lit_cost(w & 255u) + lz_len_cost(3 + 4 * L2) + lz_dist_cost(4 * (int)((rb >> 7) & 8191u)) + cs_b;
                    if (cb < best) { best = cb; ch = 1; }
                }
                if (La) {
                    const int ca = lz_len_cost(4 * La) + lz_dist_cost(4 * (int)((ra >> 7) & 8191u)) + cs_a;
                    if (ca < best) { best = ca; ch = 2; }
                }
                cs[j] = (unsigned short)best;
                c_next = best;
                L.resA[i] = (ra & 0x3fffffffu) | ((unsigned)ch << 30);
            }
            unsigned *tok = S.tokens + ((long long)b * S.chunk_off[3] + slot) * kLzChunkDw + s0;
            for (int j = 0; j < sn;) {
                const int i = s0 + j;
                const unsigned ra = L.resA[i], rb = L.resB[i], w = L.in[i];
                const int ch = (int)(ra >> 30);
                int adv = 1;
                if (ch == 0) {
                    tok[j] = lz_token(0, 0, 0);
                    atomicAdd(&L.hist[w & 255u], 1); atomicAdd(&L.hist[(w >> 8) & 255u], 1); atomicAdd(&L.hist[(w >> 16) & 255u], 1); atomicAdd(&L.hist[w >> 24], 1);
                } else {
                    int Lm, d, bytes;
                    if (ch == 1) { Lm = (int)(rb & 127u); Lm = Lm > sn - j - 1 ? sn - j - 1 : Lm; d = (int)((rb >> 7) & 8191u); bytes = 3 + 4 * Lm; adv = 1 + Lm; atomicAdd(&L.hist[w & 255u], 1); }
                    else { Lm = (int)(ra & 127u); Lm = Lm > sn - j ? sn - j : Lm; d = (int)((ra >> 7) & 8191u); bytes = 4 * Lm; adv = Lm; }
                    tok[j] = lz_token(ch, Lm, d);
                    int eb; unsigned ex;
                    atomicAdd(&L.hist[lz_len_sym(bytes, eb, ex)], 1);
                    atomicAdd(&L.hist[kDefLitLen + lz_dist_sym(4 * d, eb, ex)], 1);
                    for (int k = 1; k < adv; k++) tok[j + k] = lz_token(3, 0, 0);
                }
                j += adv;
            }
        }
    }
    __syncthreads();
    if (S.hist) {
        int *gh = S.hist + l * kDefHistBins;
        for (int i = tid; i < kDefHistBins; i += kLzThreads) if (L.hist[i]) atomicAdd(&gh[i], L.hist[i]);
        if (tid == 0 && c0 + len >= n_dw) atomicAdd(&gh[256], 1);         // the stream's end-of-block symbol
    }
}

// fixed-Huffman code of a literal / length symbol, bit-reversed for the LSB-first stream | number of bits << 16
__device__ __forceinline__ unsigned fixed_entry(int sym)
{
    int n;
    unsigned c;
    if (sym < 144) { c = 0x30u + (unsigned)sym; n = 8; }
    else if (sym < 256) { c = 0x190u + (unsigned)(sym - 144); n = 9; }
    else if (sym < 280) { c = (unsigned)(sym - 256); n = 7; }
    else { c = 0xC0u + (unsigned)(sym - 280); n = 8; }
    return (__brev(c) >> (32 - n)) | ((unsigned)n << 16);
}
__device__ __forceinline__ unsigned fixed_dist_entry(int dsym) { return (__brev((unsigned)dsym) >> 27) | (5u << 16); }

// the tokens of one sub-block, front to back:  lit(byte)  /  match(bytes, distance in bytes)
template <typename V>
__device__ __forceinline__ void lz_walk(const unsigned *tok, const unsigned *in, int sn, V &&visit)
{
    for (int j = 0; j < sn;) {
        const unsigned t = tok[j], w = in[j];
        const int kind = (int)(t & 3u), Lm = (int)((t >> 2) & 127u), d = (int)(t >> 9);
        if (kind == 0) { visit.lit(w & 255u); visit.lit((w >> 8) & 255u); visit.lit((w >> 16) & 255u); visit.lit(w >> 24); j++; }
        else if (kind == 1) { visit.lit(w & 255u); visit.match(3 + 4 * Lm, 4 * d); j += 1 + Lm; }
        else if (kind == 2 && Lm > 0) { visit.match(4 * Lm, 4 * d); j += Lm; }
        else j++;                                        // (never reached from a token start; keeps a corrupt workspace from hanging the walk)
    }
}

struct LzCountBits {
    const unsigned *tab;          // LDS copy of the layer's table, or null
    int fixed = 0, dyn = 0;
    bool missing = false;
    __device__ __forceinline__ void lit(unsigned bt) { fixed += bt < 144u ? 8 : 9; if (tab) { const int nb = (int)(tab[bt] >> 16); dyn += nb; missing = missing || nb == 0; } }
    __device__ __forceinline__ void match(int bytes, int dist)
    {
        int e1, e2; unsigned x1, x2;
        const int ls = lz_len_sym(bytes, e1, x1), ds = lz_dist_sym(dist, e2, x2);
        fixed += (int)(fixed_entry(ls) >> 16) + e1 + 5 + e2;
        if (tab) { const int nb = (int)(tab[ls] >> 16), nd = (int)(tab[kDefLitLen + ds] >> 16); dyn += nb + e1 + nd + e2; missing = missing || nb == 0 || nd == 0; }
    }
};

struct BitSink {
    unsigned *words;              // LDS, zeroed
    unsigned long long acc;
    int nacc;
    unsigned w;
    __device__ __forceinline__ void start(unsigned *base, unsigned bit_off) { words = base; w = bit_off >> 5; nacc = (int)(bit_off & 31u); acc = 0; }
    __device__ __forceinline__ void put(unsigned v, int n)
    {
        acc |= (unsigned long long)v << nacc;
        nacc += n;
        if (nacc >= 32) { atomicOr(&words[w], (unsigned)acc); acc >>= 32; nacc -= 32; w++; }
    }
    __device__ __forceinline__ void finish() { if (nacc > 0 && (unsigned)acc) atomicOr(&words[w], (unsigned)acc); }
};
struct LzEmitBits {
    BitSink &sink;
    const unsigned *tab;          // null: fixed code
    __device__ __forceinline__ void lit(unsigned bt) { const unsigned e = tab ? tab[bt] : fixed_entry((int)bt); sink.put(e & 0xffffu, (int)(e >> 16)); }
    __device__ __forceinline__ void match(int bytes, int dist)
    {
        int e1, e2; unsigned x1, x2;
        const int ls = lz_len_sym(bytes, e1, x1), ds = lz_dist_sym(dist, e2, x2);
        const unsigned e = tab ? tab[ls] : fixed_entry(ls);
        const int nb = (int)(e >> 16);
        sink.put((e & 0xffffu) | (x1 << nb), nb + e1);
        const unsigned dd = tab ? tab[kDefLitLen + ds] : fixed_dist_entry(ds);
        const int nd = (int)(dd >> 16);
        sink.put(dd & 0xffffu, nd);                   // (a code of up to 15 bits and up to 13 extra bits: two puts keep a put below 32 bits)
        if (e2) sink.put(x2, e2);
    }
};

__device__ __forceinline__ const unsigned *lz_load_table(const LzStreams &S, int l, unsigned *sTab)
{
    if (!S.tables) return nullptr;
    const unsigned *t = S.tables + l * kDefTableWords;
    for (int i = threadIdx.x; i < kDefTabHdr; i += blockDim.x) sTab[i] = t[i];
    return sTab;
}

// Pass 2: bits per sub-block and chunk under both codes, Adler-32 partial sums of the chunk.
__global__ __launch_bounds__(kLzSubs) void k_lz_sizes(LzStreams S)
{
    __shared__ __attribute__((aligned(16))) unsigned sIn[kLzChunkDw + 4];
    __shared__ unsigned sTab[kDefTabHdr];
    __shared__ unsigned sRed[5][kLzSubs / 64];
    const int slot = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    int l, c;
    lz_locate(S, slot, l, c);
    const int s = b * 3 + l;
    long long n_dw;
    const unsigned *src = lz_stream_of(S, s, n_dw);
    const long long c0 = (long long)c * kLzChunkDw;
    const long long gslot = (long long)b * S.chunk_off[3] + slot;
    if (c0 >= n_dw) { if (tid == 0) { S.chunk_bits[gslot * 2] = 0; S.chunk_bits[gslot * 2 + 1] = 0; S.chunk_missing[gslot] = 0; } return; }
    const int len = (int)(n_dw - c0 < kLzChunkDw ? n_dw - c0 : kLzChunkDw);
    for (int i = tid * 4; i < kLzChunkDw; i += kLzSubs * 4) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (i < len) v = *reinterpret_cast<const uint4 *>(src + c0 + i);
        if (i + 1 >= len) v.y = 0u;
        if (i + 2 >= len) v.z = 0u;
        if (i + 3 >= len) v.w = 0u;
        *reinterpret_cast<uint4 *>(sIn + i) = v;
    }
    const unsigned *tab = lz_load_table(S, l, sTab);
    __syncthreads();
    const int s0 = tid * kLzSubDw;
    const int sn = len - s0 < kLzSubDw ? (len - s0 > 0 ? len - s0 : 0) : kLzSubDw;
    LzCountBits cnt;
    cnt.tab = tab;
    unsigned a = 0, m = 0;
    if (sn > 0) {
        lz_walk(S.tokens + gslot * kLzChunkDw + s0, sIn + s0, sn, cnt);
        // sum of the bytes and sum of j * byte[j] over the sub-block's 4 sn bytes (zeros beyond the end add nothing)
        for (int i = 0; i < sn; i++) {
            const unsigned w = sIn[s0 + i];
            const unsigned sum = __builtin_amdgcn_udot4(w, 0x01010101u, 0u, false);
            a += sum;
            m += (unsigned)(4 * i) * sum + __builtin_amdgcn_udot4(w, 0x03020100u, 0u, false);
        }
    }
    S.sub_bits[(gslot * kLzSubs + tid) * 2] = (unsigned short)cnt.fixed;
    S.sub_bits[(gslot * kLzSubs + tid) * 2 + 1] = (unsigned short)cnt.dyn;
    // chunk totals; sum of (len_bytes - i) * byte = sum over sub-blocks of [(len_bytes - o_t) * a_t - m_t]
    const unsigned long long wsum = sn > 0 ? (unsigned long long)(4 * (len - s0)) * a - m : 0ull;
    unsigned vf = (unsigned)cnt.fixed, vd = (unsigned)cnt.dyn, va = a, vw = (unsigned)(wsum % kAdlerMod), vm = cnt.missing ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) { vf += __shfl_down(vf, o); vd += __shfl_down(vd, o); va += __shfl_down(va, o); vw += __shfl_down(vw, o); vm += __shfl_down(vm, o); }
    if ((tid & 63) == 0) { sRed[0][tid >> 6] = vf; sRed[1][tid >> 6] = vd; sRed[2][tid >> 6] = va; sRed[3][tid >> 6] = vw; sRed[4][tid >> 6] = vm; }
    __syncthreads();
    if (tid == 0) {
        unsigned tf = 0, td = 0, ta = 0, tw = 0, tm = 0;
        for (int k = 0; k < kLzSubs / 64; k++) { tf += sRed[0][k]; td += sRed[1][k]; ta += sRed[2][k]; tw += sRed[3][k]; tm += sRed[4][k]; }
        S.chunk_bits[gslot * 2] = tf;
        S.chunk_bits[gslot * 2 + 1] = td;
        S.chunk_missing[gslot] = tm ? 1 : 0;
        S.chunk_adler[gslot * 2] = ta % kAdlerMod;
        S.chunk_adler[gslot * 2 + 1] = tw % kAdlerMod;
    }
}

// Pass 3 (one workgroup per stream): which code, bit offsets of the chunks, zlib header, Adler-32, the stream's size.
__global__ __launch_bounds__(256) void k_lz_scan(LzStreams S)
{
    __shared__ unsigned long long sCarry, sTot[2];
    __shared__ unsigned long long sWave[4];
    __shared__ int sMissing, sFixed;
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int b = s / 3, l = s - 3 * b;
    long long n_dw;
    (void)lz_stream_of(S, s, n_dw);
    const int nchunks = (int)((n_dw + kLzChunkDw - 1) / kLzChunkDw);
    const long long g0 = (long long)b * S.chunk_off[3] + S.chunk_off[l];
    // totals under both codes, any missing symbol
    unsigned long long tf = 0, td = 0;
    int miss = 0;
    for (int i = tid; i < nchunks; i += 256) { tf += S.chunk_bits[(g0 + i) * 2]; td += S.chunk_bits[(g0 + i) * 2 + 1]; miss |= S.chunk_missing[g0 + i]; }
    for (int o = 32; o > 0; o >>= 1) { tf += __shfl_down(tf, o); td += __shfl_down(td, o); }
    if (tid == 0) { sMissing = 0; sTot[0] = 0; sTot[1] = 0; }
    __syncthreads();
    if (miss) sMissing = 1;
    if (lane == 0) { atomicAdd(&sTot[0], tf); atomicAdd(&sTot[1], td); }
    __syncthreads();
    const unsigned *tab = S.tables ? S.tables + l * kDefTableWords : nullptr;
    if (tid == 0) {
        // block header + tokens + end of block
        const unsigned long long bits_fixed = 3ull + sTot[0] + 7ull;
        const unsigned eob = tab ? tab[256] >> 16 : 0u;
        const unsigned long long bits_dyn = tab ? (unsigned long long)tab[kDefTabHdrBits] + sTot[1] + eob : ~0ull;
        const bool use_fixed = !tab || sMissing || eob == 0u || bits_fixed <= bits_dyn;
        sFixed = use_fixed ? 1 : 0;
        S.stream_fixed[s] = use_fixed ? 1 : 0;
        sCarry = 16ull + (use_fixed ? 3ull : (unsigned long long)tab[kDefTabHdrBits]);      // two zlib header bytes, then the block header
    }
    __syncthreads();
    const int which = sFixed ? 0 : 1;
    for (int base = 0; base < nchunks; base += 256) {
        const int i = base + tid;
        const unsigned long long v = i < nchunks ? S.chunk_bits[(g0 + i) * 2 + which] : 0ull;
        unsigned long long inc = v;
        for (int o = 1; o < 64; o <<= 1) { const unsigned long long t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        if (lane == 63) sWave[tid >> 6] = inc;
        __syncthreads();
        unsigned long long wbase = 0;
        for (int k = 0; k < (tid >> 6); k++) wbase += sWave[k];
        const unsigned long long carry = sCarry;
        if (i < nchunks) S.chunk_pos[g0 + i] = carry + wbase + inc - v;
        __syncthreads();
        if (tid == 255) sCarry = carry + wbase + inc;
        __syncthreads();
    }
    unsigned char *out = S.out + (unsigned long long)s * S.stream_stride;
    const unsigned long long end_bits = sCarry + (sFixed ? 7ull : (unsigned long long)(tab[256] >> 16));      // ... + end of block
    const unsigned long long body_end = (end_bits + 7ull) / 8ull;
    const unsigned long long total = body_end + 4ull;
    if (total > S.stream_stride) { if (tid == 0) { S.sizes[s] = (long long)total; atomicMax(S.error, 1); } return; }
    // The emit pass stores a chunk's inner words and ORs its first and last word into the stream (they are shared with the neighbours, the
    // block header and the end-of-block code): zero exactly those words, and everything from the last chunk's end to the Adler-32.
    unsigned *ow = reinterpret_cast<unsigned *>(out);
    for (int i = tid; i < nchunks; i += 256) {
        const unsigned long long p0 = S.chunk_pos[g0 + i], nb = S.chunk_bits[(g0 + i) * 2 + which];
        ow[p0 >> 5] = 0u;
        if (nb) ow[(p0 + nb - 1ull) >> 5] = 0u;
    }
    for (unsigned long long i = (nchunks ? sCarry >> 5 : 0ull) + tid; i < (total + 3ull) / 4ull; i += 256) ow[i] = 0u;
    if (tid == 0) ow[0] = 0u;
    __syncthreads();
    if (tid == 0) {
        S.sizes[s] = (long long)total;
        out[0] = 0x78; out[1] = sFixed || !tab ? 0x01 : 0xDA;      // CMF: deflate, 32 KiB window; FLG: check bits (level hint: fastest / maximum)
        // block header bits (from bit 16 of the stream)
        if (sFixed) out[2] = 0x03;                                 // BFINAL = 1, BTYPE = 01
        else {
            const unsigned hb = tab[kDefTabHdrBits];
            for (unsigned wd = 0; 32u * wd < hb; wd++) {
                const unsigned v = tab[kDefTabHdr + wd];
                for (int k = 0; k < 4; k++) out[2 + 4 * wd + k] = (unsigned char)(v >> (8 * k));      // (bits past hb are zero in the table)
            }
        }
        // end-of-block symbol behind the last chunk's tokens
        {
            const unsigned e = sFixed ? fixed_entry(256) : tab[256];
            unsigned long long v = (unsigned long long)(e & 0xffffu) << (sCarry & 7ull);
            for (unsigned long long p = sCarry >> 3; v; p++, v >>= 8) out[p] |= (unsigned char)v;
        }
        unsigned A = 1, Bs = 0;
        const unsigned *ad = S.chunk_adler + g0 * 2;
        const long long n_bytes = 4 * n_dw;
        for (int i = 0; i < nchunks; i++) {
            const long long len = (long long)(i + 1) * kLzChunkBytes <= n_bytes ? kLzChunkBytes : n_bytes - (long long)i * kLzChunkBytes;
            Bs = (unsigned)((Bs + (unsigned long long)(len % kAdlerMod) * A + ad[2 * i + 1]) % kAdlerMod);
            A = (A + ad[2 * i]) % kAdlerMod;
        }
        const unsigned adler = (Bs << 16) | A;
        out[body_end] = (unsigned char)(adler >> 24); out[body_end + 1] = (unsigned char)(adler >> 16);
        out[body_end + 2] = (unsigned char)(adler >> 8); out[body_end + 3] = (unsigned char)adler;
    }
}

// Pass 4: the chunks' bit strings, assembled in LDS and ORed into the stream at their bit offset.
constexpr int kLzOutWords = kLzChunkBytes * 15 / 8 / 4 + 8;       // the worst a dynamic code can do: 15 bits per literal byte
__global__ __launch_bounds__(kLzSubs) void k_lz_emit(LzStreams S)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lz_smem[];
    unsigned *sIn = reinterpret_cast<unsigned *>(lz_smem);           // [kLzChunkDw + 4]
    unsigned *sOut = sIn + kLzChunkDw + 4;                           // [kLzOutWords]
    unsigned *sTab = sOut + kLzOutWords;                             // [kDefTabHdr]
    unsigned *sWaveBits = sTab + kDefTabHdr;                         // [kLzSubs / 64]
    const int slot = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    int l, c;
    lz_locate(S, slot, l, c);
    const int s = b * 3 + l;
    long long n_dw;
    const unsigned *src = lz_stream_of(S, s, n_dw);
    const long long c0 = (long long)c * kLzChunkDw;
    if (c0 >= n_dw || *S.error) return;
    const long long gslot = (long long)b * S.chunk_off[3] + slot;
    const int len = (int)(n_dw - c0 < kLzChunkDw ? n_dw - c0 : kLzChunkDw);
    for (int i = tid * 4; i < kLzChunkDw; i += kLzSubs * 4) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (i < len) v = *reinterpret_cast<const uint4 *>(src + c0 + i);
        *reinterpret_cast<uint4 *>(sIn + i) = v;
    }
    const int fixed = S.stream_fixed[s];
    const unsigned *tab = fixed ? nullptr : lz_load_table(S, l, sTab);
    for (int i = tid; i < kLzOutWords; i += kLzSubs) sOut[i] = 0u;
    const unsigned long long pos = S.chunk_pos[gslot];               // bit offset of this chunk's tokens in the stream
    const unsigned mine = S.sub_bits[(gslot * kLzSubs + tid) * 2 + (fixed ? 0 : 1)];
    unsigned inc = mine;
    for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) sWaveBits[tid >> 6] = inc;
    __syncthreads();
    unsigned off = (unsigned)(pos & 31ull) + inc - mine;
    for (int k = 0; k < (tid >> 6); k++) off += sWaveBits[k];
    const int s0 = tid * kLzSubDw;
    const int sn = len - s0 < kLzSubDw ? (len - s0 > 0 ? len - s0 : 0) : kLzSubDw;
    if (sn > 0) {
        BitSink sink;
        sink.start(sOut, off);
        LzEmitBits v{ sink, tab };
        lz_walk(S.tokens + gslot * kLzChunkDw + s0, sIn + s0, sn, v);
        sink.finish();
    }
    __syncthreads();
    unsigned total = (unsigned)(pos & 31ull);
    for (int k = 0; k < kLzSubs / 64; k++) total += sWaveBits[k];
    const unsigned nwords = (total + 31u) / 32u;
    unsigned *dst = reinterpret_cast<unsigned *>(S.out + (unsigned long long)s * S.stream_stride) + (pos >> 5);
    // the first and the last word are shared with the neighbouring chunks (and the block header / end of block): OR; the rest are this chunk's alone
    for (unsigned i = tid; i < nwords; i += kLzSubs) {
        const unsigned v = sOut[i];
        if (i == 0 || i == nwords - 1) { if (v) atomicOr(&dst[i], v); }
        else dst[i] = v;
    }
}

// ---- host side: the dynamic code of one layer from its symbol histogram -------------------------------------------------------------------
// The same construction as tests/deflate_reference.py (adaptive_table / huffman_lengths / canonical_codes), the readable restatement and
// test reference: the tables are compared word for word (tests/test_host_logic.py).
namespace {
// code lengths of a Huffman code for `counts` (symbols with count 0 get length 0), none longer than `limit`
std::vector<int> huffman_lengths_host(const std::vector<long long> &counts, int limit)
{
    const int ns = (int)counts.size();
    std::vector<int> lengths((size_t)ns, 0), used;
    for (int i = 0; i < ns; i++) if (counts[(size_t)i] > 0) used.push_back(i);
    if (used.empty()) return lengths;
    if (used.size() == 1) { lengths[(size_t)used[0]] = 1; return lengths; }
    // two-queue construction: leaves sorted by (count, symbol), internal nodes appear in non-decreasing weight order
    std::vector<int> leaves = used;
    std::stable_sort(leaves.begin(), leaves.end(), [&](int a, int b) { return counts[(size_t)a] < counts[(size_t)b]; });      // (stable: ties by symbol)
    const int n = (int)leaves.size();
    std::vector<long long> weight((size_t)(2 * n - 1), 0);
    std::vector<int> parent((size_t)(2 * n - 1), 0), depth((size_t)(2 * n - 1), 0);
    for (int k = 0; k < n; k++) weight[(size_t)k] = counts[(size_t)leaves[(size_t)k]];
    int li = 0, ii = n, nxt = n;
    while (nxt < 2 * n - 1) {
        int picked[2];
        for (int t = 0; t < 2; t++) {
            if (li < n && (ii >= nxt || weight[(size_t)li] <= weight[(size_t)ii])) picked[t] = li++;
            else picked[t] = ii++;
        }
        weight[(size_t)nxt] = weight[(size_t)picked[0]] + weight[(size_t)picked[1]];
        parent[(size_t)picked[0]] = parent[(size_t)picked[1]] = nxt;
        nxt++;
    }
    for (int node = 2 * n - 3; node >= 0; node--) depth[(size_t)node] = depth[(size_t)parent[(size_t)node]] + 1;
    for (int k = 0; k < n; k++) lengths[(size_t)leaves[(size_t)k]] = std::max(depth[(size_t)k], 1);
    // length limit: clamp, then while the Kraft sum exceeds 1 lengthen the rarest symbol that can still be lengthened ...
    for (int i : used) lengths[(size_t)i] = std::min(lengths[(size_t)i], limit);
    long long kraft = 0;
    for (int i : used) kraft += 1LL << (limit - lengths[(size_t)i]);
    std::vector<int> by_rarity = used;
    {
        const std::vector<int> len0 = lengths;          // (the sort key uses the lengths as they are now)
        std::stable_sort(by_rarity.begin(), by_rarity.end(), [&](int a, int b) {
            if (counts[(size_t)a] != counts[(size_t)b]) return counts[(size_t)a] < counts[(size_t)b];
            return len0[(size_t)a] > len0[(size_t)b];
        });
    }
    while (kraft > (1LL << limit)) {
        for (int i : by_rarity)
            if (lengths[(size_t)i] < limit) { kraft -= 1LL << (limit - lengths[(size_t)i] - 1); lengths[(size_t)i]++; break; }
    }
    // ... and give back what the repair left over to the most frequent symbols
    std::vector<int> by_count = used;
    std::stable_sort(by_count.begin(), by_count.end(), [&](int a, int b) { return counts[(size_t)a] > counts[(size_t)b]; });
    for (int i : by_count)
        while (lengths[(size_t)i] > 1 && kraft + (1LL << (limit - lengths[(size_t)i])) <= (1LL << limit)) {
            kraft += 1LL << (limit - lengths[(size_t)i]);
            lengths[(size_t)i]--;
        }
    return lengths;
}

std::vector<unsigned> canonical_codes_host(const std::vector<int> &lengths)      // RFC 1951 3.2.2
{
    int max_len = 0;
    for (int l : lengths) max_len = std::max(max_len, l);
    std::vector<int> bl_count((size_t)max_len + 2, 0);
    for (int l : lengths) if (l) bl_count[(size_t)l]++;
    std::vector<unsigned> next_code((size_t)max_len + 2, 0), out(lengths.size(), 0);
    unsigned code = 0;
    for (int bits = 1; bits <= max_len; bits++) { code = (code + (unsigned)bl_count[(size_t)bits - 1]) << 1; next_code[(size_t)bits] = code; }
    for (size_t i = 0; i < lengths.size(); i++) if (lengths[i]) out[i] = next_code[(size_t)lengths[i]]++;
    return out;
}

unsigned rev_bits(unsigned code, int n) { unsigned r = 0; for (int i = 0; i < n; i++) { r = (r << 1) | (code & 1u); code >>= 1; } return r; }

struct HostBits {
    std::vector<unsigned> words;
    int n = 0;
    void put(unsigned v, int nb)
    {
        for (int i = 0; i < nb; i++, n++) {
            if ((size_t)(n >> 5) >= words.size()) words.push_back(0u);
            if ((v >> i) & 1u) words[(size_t)(n >> 5)] |= 1u << (n & 31);
        }
    }
    void put_code(unsigned code, int nb) { put(rev_bits(code, nb), nb); }
};
}  // namespace

// hist: [320] = 286 literal / length counts, then 30 distance counts; table: [kDefTableWords].  cover_all: every symbol gets a code
// (count + 1); otherwise only those that occur (and end-of-block).  Returns 0, or -1 if the block header would not fit the table.
int deflate_build_table_host(const int *hist, int cover_all, unsigned *table)
{
    static const int kClOrder[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
    std::vector<long long> ll((size_t)kDefLitLen), dd((size_t)kDefDist);
    for (int i = 0; i < kDefLitLen; i++) ll[(size_t)i] = (long long)hist[i] + (cover_all ? 1 : 0);
    for (int i = 0; i < kDefDist; i++) dd[(size_t)i] = (long long)hist[kDefLitLen + i] + (cover_all ? 1 : 0);
    if (!cover_all) {
        ll[256] = std::max(ll[256], 1LL);
        int nused = 0;
        for (long long c : ll) nused += c > 0;
        if (nused < 2) ll[ll[0] == 0 ? 0 : 1] = 1;
        int dused = 0;
        for (long long c : dd) dused += c > 0;
        // RFC 1951: one distance code of one bit is legal, none at all is written as one code of length 1
        if (dused == 0) dd[0] = 1;
    }
    const std::vector<int> ll_len = huffman_lengths_host(ll, 15);
    const std::vector<int> d_len = huffman_lengths_host(dd, 15);
    const std::vector<unsigned> ll_code = canonical_codes_host(ll_len), d_code = canonical_codes_host(d_len);
    std::vector<int> seq = ll_len;
    seq.insert(seq.end(), d_len.begin(), d_len.end());
    struct ClSym { int s; unsigned extra; int ebits; };
    std::vector<ClSym> syms;
    for (size_t i = 0; i < seq.size();) {
        const int v = seq[i];
        int run = 1;
        while (i + (size_t)run < seq.size() && seq[i + (size_t)run] == v) run++;
        i += (size_t)run;
        if (v == 0) {
            while (run >= 11) { const int r = std::min(run, 138); syms.push_back({ 18, (unsigned)(r - 11), 7 }); run -= r; }
            if (run >= 3) { syms.push_back({ 17, (unsigned)(run - 3), 3 }); run = 0; }
            for (; run > 0; run--) syms.push_back({ 0, 0u, 0 });
        } else {
            syms.push_back({ v, 0u, 0 });
            run--;
            while (run >= 3) { const int r = std::min(run, 6); syms.push_back({ 16, (unsigned)(r - 3), 2 }); run -= r; }
            for (; run > 0; run--) syms.push_back({ v, 0u, 0 });
        }
    }
    std::vector<long long> cl_hist(19, 0);
    for (const ClSym &c : syms) cl_hist[(size_t)c.s]++;
    const std::vector<int> cl_len = huffman_lengths_host(cl_hist, 7);
    const std::vector<unsigned> cl_code = canonical_codes_host(cl_len);
    int hclen = 19;
    while (hclen > 4 && cl_len[(size_t)kClOrder[hclen - 1]] == 0) hclen--;
    HostBits h;
    h.put(1u, 1);                          // BFINAL = 1: the stream is one block
    h.put(2u, 2);                          // BTYPE = 10
    h.put((unsigned)(kDefLitLen - 257), 5);   // HLIT
    h.put((unsigned)(kDefDist - 1), 5);       // HDIST: all thirty distance codes
    h.put((unsigned)(hclen - 4), 4);       // HCLEN
    for (int k = 0; k < hclen; k++) h.put((unsigned)cl_len[(size_t)kClOrder[k]], 3);
    for (const ClSym &c : syms) {
        h.put_code(cl_code[(size_t)c.s], cl_len[(size_t)c.s]);
        if (c.ebits) h.put(c.extra, c.ebits);
    }
    if (h.n > kDefHdrWords * 32) return -1;
    for (int i = 0; i < kDefTableWords; i++) table[i] = 0u;
    for (int i = 0; i < kDefLitLen; i++) table[i] = rev_bits(ll_code[(size_t)i], ll_len[(size_t)i]) | ((unsigned)ll_len[(size_t)i] << 16);
    for (int i = 0; i < kDefDist; i++) table[kDefLitLen + i] = rev_bits(d_code[(size_t)i], d_len[(size_t)i]) | ((unsigned)d_len[(size_t)i] << 16);
    table[kDefTabHdrBits] = (unsigned)h.n;
    for (size_t w = 0; w < h.words.size(); w++) table[kDefTabHdr + w] = h.words[w];
    return 0;
}

unsigned long long deflate_stream_bound(unsigned long long raw_bytes)
{
    // zlib header + block header (<= 131 words) + at most 9 bits per byte under the fixed code (the scan falls back to it when smaller) + end of block + Adler-32
    return 2 + 4 * kDefHdrWords + raw_bytes + raw_bytes / 8 + 16;
}

static long long chunks_of(long long coeffs) { return (coeffs + kLzChunkDw - 1) / kLzChunkDw; }

static void lz_layout(const long long *coeff_cap, int chunk_off[4])
{
    chunk_off[0] = 0;
    for (int l = 0; l < 3; l++) chunk_off[l + 1] = chunk_off[l] + (int)chunks_of(coeff_cap[l]);
}

static unsigned long long al256(unsigned long long v) { return (v + 255ull) & ~255ull; }

unsigned long long deflate_workspace_bytes(int batch, const long long *coeff_cap)
{
    int co[4];
    lz_layout(coeff_cap, co);
    const unsigned long long n = (unsigned long long)batch * co[3];
    return 256 + al256(n * kLzChunkBytes) + al256(n * 2 * sizeof(unsigned)) * 2 + al256(n) + al256(n * kLzSubs * 2 * sizeof(unsigned short)) +
           al256(n * sizeof(unsigned long long)) + al256((unsigned long long)batch * 3);
}

static LzStreams lz_args(const int *coeffs, const long long *counts, int batch, long long coeff_stride, const long long *coeff_off, const long long *coeff_cap,
                         void *workspace)
{
    LzStreams S;
    memset(&S, 0, sizeof S);
    S.coeffs = coeffs; S.counts = counts; S.coeff_stride = coeff_stride;
    for (int l = 0; l < 3; l++) { S.coeff_off[l] = coeff_off[l]; S.coeff_cap[l] = coeff_cap[l]; }
    lz_layout(coeff_cap, S.chunk_off);
    const unsigned long long n = (unsigned long long)batch * S.chunk_off[3];
    char *w = static_cast<char *>(workspace);
    S.error = reinterpret_cast<int *>(w); w += 256;
    S.tokens = reinterpret_cast<unsigned *>(w); w += al256(n * kLzChunkBytes);
    S.chunk_bits = reinterpret_cast<unsigned *>(w); w += al256(n * 2 * sizeof(unsigned));
    S.chunk_adler = reinterpret_cast<unsigned *>(w); w += al256(n * 2 * sizeof(unsigned));
    S.chunk_missing = reinterpret_cast<unsigned char *>(w); w += al256(n);
    S.sub_bits = reinterpret_cast<unsigned short *>(w); w += al256(n * kLzSubs * 2 * sizeof(unsigned short));
    S.chunk_pos = reinterpret_cast<unsigned long long *>(w); w += al256(n * sizeof(unsigned long long));
    S.stream_fixed = reinterpret_cast<unsigned char *>(w);
    return S;
}

static void lz_launch_parse(hipStream_t st, const LzStreams &S, int batch)
{
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lz_parse), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LzLds)); attr = true; }
    (void)hipMemsetAsync(S.error, 0, 256, st);
    if (S.chunk_off[3] > 0) hipLaunchKernelGGL(k_lz_parse, dim3(S.chunk_off[3], batch), dim3(kLzThreads), sizeof(LzLds), st, S);
}

void launch_deflate_parse(hipStream_t st, const int *coeffs, const long long *counts, int batch, long long coeff_stride, const long long *coeff_off,
                          const long long *coeff_cap, int *hist, void *workspace)
{
    LzStreams S = lz_args(coeffs, counts, batch, coeff_stride, coeff_off, coeff_cap, workspace);
    S.hist = hist;
    if (hist) (void)hipMemsetAsync(hist, 0, 3 * kDefHistBins * sizeof(int), st);
    lz_launch_parse(st, S, batch);
}

void launch_deflate(hipStream_t st, const int *coeffs, const long long *counts, int batch, long long coeff_stride, const long long *coeff_off,
                    const long long *coeff_cap, const unsigned *tables, int reuse_parse, unsigned char *out, unsigned long long stream_stride, long long *sizes,
                    void *workspace)
{
    LzStreams S = lz_args(coeffs, counts, batch, coeff_stride, coeff_off, coeff_cap, workspace);
    S.tables = tables;
    S.out = out; S.stream_stride = stream_stride; S.sizes = sizes;
    if (!reuse_parse) lz_launch_parse(st, S, batch);
    constexpr size_t emit_lds = (size_t)(kLzChunkDw + 4 + kLzOutWords + kDefTabHdr + kLzSubs / 64 + 4) * sizeof(unsigned);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lz_emit), hipFuncAttributeMaxDynamicSharedMemorySize, (int)emit_lds); attr = true; }
    if (S.chunk_off[3] > 0) hipLaunchKernelGGL(k_lz_sizes, dim3(S.chunk_off[3], batch), dim3(kLzSubs), 0, st, S);
    hipLaunchKernelGGL(k_lz_scan, dim3(batch * 3), dim3(256), 0, st, S);
    if (S.chunk_off[3] > 0) hipLaunchKernelGGL(k_lz_emit, dim3(S.chunk_off[3], batch), dim3(kLzSubs), emit_lds, st, S);
}

}  // namespace aej
