// deflate.hip -- OPT-IN GPU entropy stage for the .ajpg container: every layer's int32 coefficient array as a zlib stream
// (RFC 1950 / 1951) that the reference's decoder reads with zlib.decompress (src/jpeg/jpeg.py:659).  The reference writes these streams
// with zlib.compress(level=9) on the host (jpeg.py:588-590); with the hot path on the GPU that call IS Jpeg.compress end to end
// (98 % of compress_many on natural 4K images, 1.4 MP/s per host core: profiles/r04_bench_extra_natural.json).  This stage trades
// compression ratio for five orders of magnitude of speed and is never the default: the default container stays byte-identical to the
// reference's.
//
// Format.  A stream is cut into chunks of 32 KiB of input; a chunk is ONE Huffman block followed by an empty stored block, which pads to
// a byte boundary (the Z_SYNC_FLUSH marker 00 00 FF FF), so chunks are compressed independently and concatenated bytewise; a final empty
// block and the Adler-32 close the stream.  The block is coded with the smaller of two codes: RFC 1951's fixed code (BTYPE = 01), or a
// dynamic code (BTYPE = 10) that the HOST builds once per layer of a batch from the symbol histogram k_deflate_hist counts
// (adaptive_edge_aware_jpeg_amd/deflate_tables.py: length-limited Huffman, block header) -- the kernels are table-driven and never build
// a tree.  Without a table every block is fixed-Huffman.  LZ77 matches are
// restricted to distances 1 and 4 -- the previous byte and the same byte of the previous coefficient -- which is what an array of
// mostly-zero, small-magnitude little-endian int32 values offers: zero runs (distance 1, up to 258 bytes per 12-13-bit token), the
// three sign / zero bytes of a small coefficient behind another small coefficient (distance 4, length 3: 12 bits).  A thread parses 128
// input bytes greedily; matches do not cross its sub-block (a zero run costs one 12-13-bit token per 128 bytes).
#include "aej_common.h"
#include "aej_launch.h"
#include <string.h>

#include <algorithm>
#include <vector>

namespace aej {

// Input bytes per thread.  A match cannot leave its sub-block, so longer sub-blocks compress better (runs of zero coefficients): 256 bytes
// instead of 128 is -13 % on sparse layers and -3 % on natural images (512 adds little) -- but a 32 KiB block then has 128 threads, half the
// waves per CU for the same LDS, and the three passes take 4.7 instead of 1.9 ms per 400 MB (profiles/r04_deflate_kernels.txt).  The
// stage exists for speed: 128.  (deflate_tables.SUB is the Python restatement's copy of this constant; 128 and 256 are both built.)
constexpr int kDefSub = 128;
constexpr int kDefThreads = 32768 / kDefSub;
constexpr int kDefChunk = kDefSub * kDefThreads;  // 32 KiB of input per workgroup
constexpr int kDefSubStride = kDefSub + 4;        // LDS stride of a sub-block: 65 dwords, so equal offsets of different threads fall into different banks
constexpr int kDefOutWords = (kDefChunk * 9 / 8 + 64) / 4;      // fixed code: a literal costs at most 9 bits; a block whose dynamic code needs more falls back to it
constexpr unsigned kAdlerMod = 65521u;
constexpr int kDefTableWords = 385;               // AEJ_DEFLATE_TABLE_WORDS: 286 literal / length codes, 2 distance codes, header bit count, 96 header words
constexpr int kDefHistBins = 288;                 // 286 literal / length symbols, then the matches at distance 1 and at distance 4

struct DeflateStreams {
    const int *coeffs;            // [B][coeff_stride]
    const long long *counts;      // [B][3][4]: n_coeffs first
    long long coeff_stride, coeff_off[3];
    unsigned char *out;           // [B * 3][stream_stride]
    unsigned long long stream_stride;
    long long *sizes;             // [B * 3] bytes of each finished stream
    int *chunk_bytes;             // [B * 3][max_chunks] compressed bytes per chunk, then (after the scan) its exclusive offset
    unsigned *chunk_adler;        // [B * 3][max_chunks][2] sum of bytes, sum of (len - i) * byte, both mod 65521
    unsigned short *sub_bits;     // [B * 3][max_chunks][kDefThreads] bits of each thread's tokens under the code the chunk uses
    unsigned char *chunk_fixed;   // [B * 3][max_chunks] 1 = the chunk's block uses the fixed code
    const unsigned *tables;       // [3][kDefTableWords] per-layer dynamic codes (deflate_tables.py), or null: fixed code everywhere
    int *hist;                    // [3][kDefHistBins] (k_deflate_hist only)
    int *error;                   // [1] set when a stream does not fit its slot
    int max_chunks;
};

// fixed-Huffman code of a literal / length symbol, bit-reversed for the LSB-first stream; returns the number of bits
__device__ __forceinline__ int fixed_code(int sym, unsigned &code)
{
    int n;
    unsigned c;
    if (sym < 144) { c = 0x30u + (unsigned)sym; n = 8; }
    else if (sym < 256) { c = 0x190u + (unsigned)(sym - 144); n = 9; }
    else if (sym < 280) { c = (unsigned)(sym - 256); n = 7; }
    else { c = 0xC0u + (unsigned)(sym - 280); n = 8; }
    code = __brev(c) >> (32 - n);
    return n;
}

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

// Greedy parse of one sub-block of n bytes at stream position gpos: at every position the longer of the runs "equal to the byte one /
// four positions back" (clipped to the sub-block) becomes a match when it is at least 3 long, else the byte is a literal.
// `left4` holds the four bytes before the sub-block (byte k = position k - 4).  What happens to a token is the visitor's business:
//   lit(byte)   /   match(length symbol, extra value, extra bits, distance-is-4)
//
// The runs come from two 128-bit masks -- bit p of E1 / E4 = "byte p equals the byte one / four positions before it" -- built once per
// sub-block from its 32 dwords with whole-word arithmetic; the run that starts at p is then the number of consecutive ones from bit p
// (a shift and a count-trailing-zeros).  The first version walked the runs byte by byte: the lanes of a wave sit in runs of very
// different lengths, every step of the token loop cost the longest run among 64 lanes, and the three passes took 13.4 ms per 400 MB
// (profiles/r04_deflate_kernels.txt).
struct SubMasks { unsigned long long e1[kDefSub / 64], e4[kDefSub / 64]; };
static_assert(kDefSub == 128 || kDefSub == 256, "two or four mask words");

__device__ __forceinline__ unsigned zero_byte_nibble(unsigned x)      // bit k = "byte k of x is zero"
{
    unsigned t = (x & 0x7f7f7f7fu) + 0x7f7f7f7fu;
    t = ~(t | x | 0x7f7f7f7fu);                      // 0x80 in every zero byte, exact (no borrow between bytes)
    return ((t >> 7) * 0x01020408u) >> 24;           // the four flags (bits 0, 8, 16, 24) gathered into bits 0..3
}

__device__ __forceinline__ SubMasks deflate_masks(const unsigned *subw /* kDefSub / 4 dwords, LDS */, unsigned left4, long long gpos)
{
    SubMasks m;
#pragma unroll
    for (int k = 0; k < kDefSub / 64; k++) { m.e1[k] = 0ull; m.e4[k] = 0ull; }
    unsigned prev = left4;
#pragma unroll
    for (int i = 0; i < kDefSub / 4; i++) {          // (fully unrolled: the word index i / 16 is a constant, the masks stay in registers)
        const unsigned w = subw[i];
        const unsigned long long n4 = zero_byte_nibble(w ^ prev);                              // byte k against the byte four back
        const unsigned long long n1 = zero_byte_nibble(w ^ ((w << 8) | (prev >> 24)));         // byte k against the byte before it
        m.e4[i / 16] |= n4 << (4 * (i % 16));
        m.e1[i / 16] |= n1 << (4 * (i % 16));
        prev = w;
    }
    if (gpos == 0) { m.e1[0] &= ~1ull; m.e4[0] &= ~15ull; }      // nothing lies before the first bytes of a stream
    return m;
}

// consecutive ones of the kDefSub-bit mask from bit p (the word is picked by compares: no indexed register access)
__device__ __forceinline__ int ones_from(const unsigned long long (&m)[kDefSub / 64], int p)
{
    if constexpr (kDefSub == 128) {                 // two words: branch-free (0.35 / 0.44 / 0.92 ms for the three passes against 0.48 / 0.63 / 1.24 with the loop)
        unsigned long long a, b;
        if (p < 64) { a = p ? (m[0] >> p) | (m[1] << (64 - p)) : m[0]; b = m[1] >> p; }
        else { a = m[1] >> (p - 64); b = 0ull; }
        const unsigned long long na = ~a, nb = ~b;
        return na ? __builtin_ctzll(na) : 64 + (nb ? __builtin_ctzll(nb) : 64);
    }
    int run = 0;
    while (p < kDefSub) {
        const int k = p >> 6, s = p & 63;
        unsigned long long word = m[0];
#pragma unroll
        for (int j = 1; j < kDefSub / 64; j++) word = k == j ? m[j] : word;
        const unsigned long long inv = ~(word >> s);             // (the zeros shifted in at the top end the count at the word's edge)
        const int avail = 64 - s;
        const int ones = inv ? __builtin_ctzll(inv) : 64;
        if (ones < avail) return run + ones;
        run += avail;
        p += avail;
    }
    return run;
}

template <typename V>
__device__ __forceinline__ void deflate_parse(const unsigned char *sub, unsigned left4, int n, long long gpos, V &&visit)
{
    const SubMasks m = deflate_masks(reinterpret_cast<const unsigned *>(sub), left4, gpos);
    int p = 0;
    while (p < n) {
        const int room = n - p;
        int l1 = ones_from(m.e1, p), l4 = ones_from(m.e4, p);
        l1 = l1 < room ? l1 : room;
        l4 = l4 < room ? l4 : room;
        const int L = l1 >= l4 ? l1 : l4;
        if (L >= 3) {
            const int l = L - 3;
            int sym, e = 0;
            unsigned extra = 0;
            if (l < 8) sym = 257 + l;
            else { e = 29 - __clz(l); sym = 257 + 4 * (e + 1) + ((l >> e) & 3); extra = (unsigned)l & ((1u << e) - 1u); }
            visit.match(sym, extra, e, l4 > l1);
            p += L;
        } else {
            visit.lit((int)sub[p]);
            p++;
        }
    }
}

// a code as the kernels use it: the code's bits reversed (the stream is LSB first) | number of bits << 16
__device__ __forceinline__ unsigned fixed_entry(int sym) { unsigned c; const int n = fixed_code(sym, c); return c | ((unsigned)n << 16); }
__device__ __forceinline__ unsigned fixed_dist_entry(bool four) { return (four ? (__brev(3u) >> 27) : 0u) | (5u << 16); }

struct CountBits {               // bits of the tokens under the fixed code and under the table's code
    const unsigned *tab;         // LDS copy of the layer's table, or null
    int fixed = 0, dyn = 0;
    bool missing = false;        // a token has no code in the table (a table counted on other data): the chunk takes the fixed code
    __device__ __forceinline__ void lit(int b)
    {
        fixed += (int)(fixed_entry(b) >> 16);
        if (tab) { const int nb = (int)(tab[b] >> 16); dyn += nb; missing = missing || nb == 0; }
    }
    __device__ __forceinline__ void match(int sym, unsigned, int e, bool four)
    {
        fixed += (int)(fixed_entry(sym) >> 16) + e + 5;
        if (tab) {
            const int nb = (int)(tab[sym] >> 16), nd = (int)(tab[286 + (four ? 1 : 0)] >> 16);
            dyn += nb + e + nd;
            missing = missing || nb == 0 || nd == 0;
        }
    }
};
struct CountSymbols {            // histogram of the tokens (LDS)
    int *hist;
    __device__ __forceinline__ void lit(int b) { atomicAdd(&hist[b], 1); }
    __device__ __forceinline__ void match(int sym, unsigned, int, bool four) { atomicAdd(&hist[sym], 1); atomicAdd(&hist[286 + (four ? 1 : 0)], 1); }
};
struct EmitBits {
    BitSink &sink;
    const unsigned *tab;         // null: fixed code
    __device__ __forceinline__ void lit(int b) { const unsigned e = tab ? tab[b] : fixed_entry(b); sink.put(e & 0xffffu, (int)(e >> 16)); }
    __device__ __forceinline__ void match(int sym, unsigned extra, int ne, bool four)
    {
        const unsigned e = tab ? tab[sym] : fixed_entry(sym);
        const int nb = (int)(e >> 16);
        sink.put((e & 0xffffu) | (extra << nb), nb + ne);
        const unsigned d = tab ? tab[286 + (four ? 1 : 0)] : fixed_dist_entry(four);
        sink.put(d & 0xffffu, (int)(d >> 16));
    }
};

// stage a chunk in LDS (coalesced 16-byte loads; sub-blocks at a stride of 65 dwords) with the four bytes before it
__device__ __forceinline__ void deflate_stage(const unsigned char *src, long long n_bytes, long long c0, int len, unsigned char *sIn /* [4 + threads * stride] */)
{
    const int tid = threadIdx.x;
    if (tid < 4) sIn[tid] = c0 >= 4 ? src[c0 - 4 + tid] : 0;
    // coefficient arrays start on 256-byte boundaries, their capacities are multiples of 64 bytes and a chunk is 32 KiB: 16-byte loads are
    // aligned and stay inside the layer's slot; what they bring beyond `len` is never looked at
    for (int i = tid * 16; i < kDefChunk; i += kDefThreads * 16) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (i < len) v = *reinterpret_cast<const uint4 *>(src + c0 + i);
        unsigned *dst = reinterpret_cast<unsigned *>(sIn + 4 + (i / kDefSub) * kDefSubStride + (i % kDefSub));
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
}

// the four bytes before a thread's sub-block: the chunk's halo (sIn[0..3]) for thread 0, else the tail of the sub-block before it, which
// ends four bytes short of this one (stride kDefSub + 4)
__device__ __forceinline__ unsigned deflate_left4(const unsigned char *sIn, const unsigned char *sub, int tid)
{
    const unsigned char *q = tid == 0 ? sIn : sub - 8;
    return (unsigned)q[0] | ((unsigned)q[1] << 8) | ((unsigned)q[2] << 16) | ((unsigned)q[3] << 24);
}

__device__ __forceinline__ const unsigned char *stream_of(const DeflateStreams &S, int s, long long &n_bytes)
{
    const int b = s / 3, l = s - 3 * b;
    n_bytes = 4 * S.counts[(long long)s * 4];
    return reinterpret_cast<const unsigned char *>(S.coeffs + (long long)b * S.coeff_stride + S.coeff_off[l]);
}

// Pass 0 (only when dynamic codes are wanted): how often every symbol occurs, per layer.
__global__ __launch_bounds__(kDefThreads) void k_deflate_hist(DeflateStreams S)
{
    __shared__ __attribute__((aligned(16))) unsigned char sIn[4 + kDefThreads * kDefSubStride + 12];
    __shared__ int sHist[kDefHistBins];
    const int c = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
    long long n_bytes;
    const unsigned char *src = stream_of(S, s, n_bytes);
    const long long c0 = (long long)c * kDefChunk;
    if (c0 >= n_bytes) return;
    const int len = (int)(n_bytes - c0 < kDefChunk ? n_bytes - c0 : kDefChunk);
    deflate_stage(src, n_bytes, c0, len, sIn);
    for (int i = tid; i < kDefHistBins; i += kDefThreads) sHist[i] = 0;
    __syncthreads();
    const int n = min(kDefSub, len - tid * kDefSub);
    if (n > 0) {
        const unsigned char *sub = sIn + 4 + tid * kDefSubStride;
        CountSymbols v{ sHist };
        deflate_parse(sub, deflate_left4(sIn, sub, tid), n, c0 + (long long)tid * kDefSub, v);
    }
    __syncthreads();
    int *gh = S.hist + (s % 3) * kDefHistBins;
    for (int i = tid; i < kDefHistBins; i += kDefThreads) if (sHist[i]) atomicAdd(&gh[i], sHist[i]);
}

__device__ __forceinline__ const unsigned *deflate_load_table(const DeflateStreams &S, int s, unsigned *sTab)
{
    if (!S.tables) return nullptr;
    const unsigned *t = S.tables + (s % 3) * kDefTableWords;
    for (int i = threadIdx.x; i < kDefTableWords; i += kDefThreads) sTab[i] = t[i];
    return sTab;
}

// Pass 1: bits per sub-block, which code the chunk's block uses, compressed bytes and Adler-32 partial sums per chunk.
__global__ __launch_bounds__(kDefThreads) void k_deflate_sizes(DeflateStreams S)
{
    __shared__ __attribute__((aligned(16))) unsigned char sIn[4 + kDefThreads * kDefSubStride + 12];
    __shared__ unsigned sTab[kDefTableWords];
    __shared__ unsigned sRed[4][kDefThreads / 64];
    __shared__ int sUseFixed;
    const int c = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
    long long n_bytes;
    const unsigned char *src = stream_of(S, s, n_bytes);
    const long long c0 = (long long)c * kDefChunk;
    const long long slot = (long long)s * S.max_chunks + c;
    if (c0 >= n_bytes) { if (tid == 0) S.chunk_bytes[slot] = 0; return; }
    const int len = (int)(n_bytes - c0 < kDefChunk ? n_bytes - c0 : kDefChunk);
    deflate_stage(src, n_bytes, c0, len, sIn);
    const unsigned *tab = deflate_load_table(S, s, sTab);
    __syncthreads();
    const int n = min(kDefSub, len - tid * kDefSub);
    const unsigned char *sub = sIn + 4 + tid * kDefSubStride;
    CountBits cnt;
    cnt.tab = tab;
    unsigned a = 0, m = 0;
    if (n > 0) {
        deflate_parse(sub, deflate_left4(sIn, sub, tid), n, c0 + (long long)tid * kDefSub, cnt);
        // sum of the bytes and sum of j * byte[j], a dword at a time (bytes past n are not part of the stream: masked off)
        const unsigned *subw = reinterpret_cast<const unsigned *>(sub);
#pragma unroll
        for (int i = 0; i < kDefSub / 4; i++) {
            unsigned w = subw[i];
            const int left = n - 4 * i;
            if (left < 4) w = left > 0 ? w & ((1u << (8 * left)) - 1u) : 0u;
            const unsigned sum = __builtin_amdgcn_udot4(w, 0x01010101u, 0u, false);
            a += sum;
            m += (unsigned)(4 * i) * sum + __builtin_amdgcn_udot4(w, 0x03020100u, 0u, false);
        }
    }
    // the last thread also writes the end-of-block symbol
    if (tid == kDefThreads - 1) { cnt.fixed += 7; if (tab) { cnt.dyn += (int)(tab[256] >> 16); cnt.missing = cnt.missing || (tab[256] >> 16) == 0; } }
    const int any_missing = __syncthreads_or(cnt.missing ? 1 : 0);
    // chunk totals: bits under either code, sum of bytes, sum of (len - i) * byte = sum_t [(len - o_t) * a_t - m_t]
    unsigned long long w = n > 0 ? (unsigned long long)(len - tid * kDefSub) * a - m : 0ull;
    unsigned vf = (unsigned)cnt.fixed, vd = (unsigned)cnt.dyn, va = a, vw = (unsigned)(w % kAdlerMod);
    for (int o = 32; o > 0; o >>= 1) { vf += __shfl_down(vf, o); vd += __shfl_down(vd, o); va += __shfl_down(va, o); vw += __shfl_down(vw, o); }
    if ((tid & 63) == 0) { sRed[0][tid >> 6] = vf; sRed[1][tid >> 6] = vd; sRed[2][tid >> 6] = va; sRed[3][tid >> 6] = vw; }
    __syncthreads();
    if (tid == 0) {
        unsigned tf = 0, td = 0, ta = 0, tw = 0;
        for (int k = 0; k < kDefThreads / 64; k++) { tf += sRed[0][k]; td += sRed[1][k]; ta += sRed[2][k]; tw += sRed[3][k]; }
        // block header + tokens + end of block, then the header of the empty stored block; the dynamic code only when it is smaller AND the
        // block fits the emit kernel's LDS buffer (a chunk the layer's code does not suit can cost up to 15 bits per byte)
        const unsigned bits_fixed = 3u + tf + 3u, bits_dyn = tab ? tab[288] + td + 3u : 0xffffffffu;
        const bool use_fixed = !tab || any_missing || bits_fixed <= bits_dyn || (bits_dyn + 7u) / 8u + 4u > (unsigned)(kDefOutWords * 4);
        sUseFixed = use_fixed ? 1 : 0;
        S.chunk_fixed[slot] = use_fixed ? 1 : 0;
        S.chunk_bytes[slot] = (int)(((use_fixed ? bits_fixed : bits_dyn) + 7u) / 8u) + 4;    // ... padded to a byte, LEN = 0000, NLEN = FFFF
        S.chunk_adler[slot * 2] = ta % kAdlerMod;
        S.chunk_adler[slot * 2 + 1] = tw % kAdlerMod;
    }
    __syncthreads();
    S.sub_bits[slot * kDefThreads + tid] = (unsigned short)(sUseFixed ? cnt.fixed : cnt.dyn);
}

// Pass 2 (one workgroup per stream): offsets of the chunks, the stream's size, its zlib header, final block and Adler-32.
__global__ __launch_bounds__(256) void k_deflate_scan(DeflateStreams S)
{
    __shared__ long long sCarry;
    __shared__ int sWave[4];
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    long long n_bytes;
    (void)stream_of(S, s, n_bytes);
    const int nchunks = (int)((n_bytes + kDefChunk - 1) / kDefChunk);
    int *cb = S.chunk_bytes + (long long)s * S.max_chunks;
    if (tid == 0) sCarry = 2;                       // the two header bytes
    __syncthreads();
    for (int base = 0; base < nchunks; base += 256) {
        const int i = base + tid;
        const int v = i < nchunks ? cb[i] : 0;
        int inc = v;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        if (lane == 63) sWave[tid >> 6] = inc;
        __syncthreads();
        int wbase = 0;
        for (int k = 0; k < (tid >> 6); k++) wbase += sWave[k];
        const long long carry = sCarry;
        if (i < nchunks) cb[i] = (int)(carry + wbase + inc - v);      // exclusive offset inside the stream (streams are < 2 GiB)
        __syncthreads();
        if (tid == 255) sCarry = carry + wbase + inc;
        __syncthreads();
    }
    if (tid == 0) {
        const long long body_end = sCarry;
        const long long total = body_end + 2 + 4;
        S.sizes[s] = total;
        if ((unsigned long long)total > S.stream_stride) { *S.error = 1; return; }
        unsigned char *out = S.out + (unsigned long long)s * S.stream_stride;
        out[0] = 0x78; out[1] = 0x01;                // CMF: deflate, 32 KiB window; FLG: check bits, fastest level
        out[body_end] = 0x03; out[body_end + 1] = 0x00;      // final block: BFINAL = 1, fixed Huffman, end of block
        unsigned A = 1, B = 0;
        const unsigned *ad = S.chunk_adler + (long long)s * S.max_chunks * 2;
        for (int i = 0; i < nchunks; i++) {
            const long long len = (long long)(i + 1) * kDefChunk <= n_bytes ? kDefChunk : n_bytes - (long long)i * kDefChunk;
            B = (unsigned)((B + (unsigned long long)(len % kAdlerMod) * A + ad[2 * i + 1]) % kAdlerMod);
            A = (A + ad[2 * i]) % kAdlerMod;
        }
        const unsigned adler = (B << 16) | A;
        out[body_end + 2] = (unsigned char)(adler >> 24); out[body_end + 3] = (unsigned char)(adler >> 16);
        out[body_end + 4] = (unsigned char)(adler >> 8);  out[body_end + 5] = (unsigned char)adler;
    }
}

// Pass 3: the chunks' bit strings, assembled in LDS and copied to their place in the stream.
__global__ __launch_bounds__(kDefThreads) void k_deflate_emit(DeflateStreams S)
{
    __shared__ __attribute__((aligned(16))) unsigned char sIn[4 + kDefThreads * kDefSubStride + 12];
    __shared__ unsigned sOut[kDefOutWords];
    __shared__ unsigned sTab[kDefTableWords];
    __shared__ unsigned sWaveBits[kDefThreads / 64];
    const int c = blockIdx.x, s = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    long long n_bytes;
    const unsigned char *src = stream_of(S, s, n_bytes);
    const long long c0 = (long long)c * kDefChunk;
    if (c0 >= n_bytes || *S.error) return;
    const long long slot = (long long)s * S.max_chunks + c;
    const int len = (int)(n_bytes - c0 < kDefChunk ? n_bytes - c0 : kDefChunk);
    deflate_stage(src, n_bytes, c0, len, sIn);
    const unsigned *tab = S.chunk_fixed[slot] ? nullptr : deflate_load_table(S, s, sTab);
    for (int i = tid; i < kDefOutWords; i += kDefThreads) sOut[i] = 0u;
    // bit offset of this thread's tokens: the block header + the bits of the threads before it
    const unsigned mine = S.sub_bits[slot * kDefThreads + tid];
    unsigned inc = mine;
    for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) sWaveBits[tid >> 6] = inc;
    __syncthreads();
    const unsigned hdr_bits = tab ? tab[288] : 3u;
    unsigned off = hdr_bits + inc - mine;
    for (int k = 0; k < (tid >> 6); k++) off += sWaveBits[k];
    const int n = min(kDefSub, len - tid * kDefSub);
    BitSink sink;
    sink.start(sOut, tid == 0 ? 0u : off);
    if (tid == 0) {
        if (tab) for (unsigned w = 0; 32u * w < hdr_bits; w++) sink.put(tab[289 + w], (int)min(32u, hdr_bits - 32u * w));      // BFINAL = 0, BTYPE = 10, the code lengths
        else sink.put(2u, 3);                                                                                                   // BFINAL = 0, BTYPE = 01
    }
    if (n > 0) {
        const unsigned char *sub = sIn + 4 + tid * kDefSubStride;
        EmitBits v{ sink, tab };
        deflate_parse(sub, deflate_left4(sIn, sub, tid), n, c0 + (long long)tid * kDefSub, v);
    }
    if (tid == kDefThreads - 1) { const unsigned e = tab ? tab[256] : fixed_entry(256); sink.put(e & 0xffffu, (int)(e >> 16)); }      // end of block
    sink.finish();
    __syncthreads();
    // the empty stored block's header (three zero bits) is already there; pad to a byte, then 00 00 FF FF
    unsigned total = hdr_bits;
    for (int k = 0; k < kDefThreads / 64; k++) total += sWaveBits[k];
    const unsigned body = (total + 3u + 7u) / 8u;
    unsigned char *ob = reinterpret_cast<unsigned char *>(sOut);
    if (tid == 0) { ob[body + 2] = 0xFF; ob[body + 3] = 0xFF; }
    __syncthreads();
    const unsigned nout = body + 4u;
    unsigned char *dst = S.out + (unsigned long long)s * S.stream_stride + S.chunk_bytes[slot];
    for (unsigned i = tid; i < nout; i += kDefThreads) dst[i] = ob[i];
}

// ---- host side: the dynamic code of one layer from its symbol histogram -------------------------------------------------------------------
// The same construction as adaptive_edge_aware_jpeg_amd/deflate_tables.py (adaptive_table / huffman_lengths / canonical_codes), which
// stays the readable restatement and the test reference: the tables are compared word for word (tests/test_host_logic.py).  In Python
// the three tables of a call cost 2.7 ms -- a quarter of compress_many(entropy="gpu") -- here tens of microseconds.
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

// hist: [288] = 286 literal / length counts, then the matches at distance 1 and 4; table: [kDefTableWords].  cover_all: every symbol gets
// a code (count + 1); otherwise only those that occur (and end-of-block).  Returns 0, or -1 if the block header would not fit the table.
int deflate_build_table_host(const int *hist, int cover_all, unsigned *table)
{
    constexpr int kLitLen = 286;
    static const int kClOrder[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
    std::vector<long long> ll((size_t)kLitLen);
    for (int i = 0; i < kLitLen; i++) ll[(size_t)i] = (long long)hist[i] + (cover_all ? 1 : 0);
    if (!cover_all) {
        ll[256] = std::max(ll[256], 1LL);
        int nused = 0;
        for (long long c : ll) nused += c > 0;
        if (nused < 2) ll[ll[0] == 0 ? 0 : 1] = 1;
    }
    const std::vector<int> ll_len = huffman_lengths_host(ll, 15);
    const std::vector<int> d_len = { 1, 0, 0, 1 };
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
    h.put(0u, 1);                          // BFINAL = 0
    h.put(2u, 2);                          // BTYPE = 10
    h.put((unsigned)(kLitLen - 257), 5);   // HLIT
    h.put(3u, 5);                          // HDIST: four distance codes
    h.put((unsigned)(hclen - 4), 4);       // HCLEN
    for (int k = 0; k < hclen; k++) h.put((unsigned)cl_len[(size_t)kClOrder[k]], 3);
    for (const ClSym &c : syms) {
        h.put_code(cl_code[(size_t)c.s], cl_len[(size_t)c.s]);
        if (c.ebits) h.put(c.extra, c.ebits);
    }
    if (h.n > (kDefTableWords - 289) * 32) return -1;
    for (int i = 0; i < kDefTableWords; i++) table[i] = 0u;
    for (int i = 0; i < kLitLen; i++) table[i] = rev_bits(ll_code[(size_t)i], ll_len[(size_t)i]) | ((unsigned)ll_len[(size_t)i] << 16);
    table[286] = rev_bits(d_code[0], d_len[0]) | ((unsigned)d_len[0] << 16);
    table[287] = rev_bits(d_code[3], d_len[3]) | ((unsigned)d_len[3] << 16);
    table[288] = (unsigned)h.n;
    for (size_t w = 0; w < h.words.size(); w++) table[289 + w] = h.words[w];
    return 0;
}

unsigned long long deflate_stream_bound(unsigned long long raw_bytes)
{
    const unsigned long long chunks = (raw_bytes + kDefChunk - 1) / kDefChunk;
    return 2 + raw_bytes + raw_bytes / 8 + chunks * 8 + 16;
}

int deflate_max_chunks(long long max_coeffs) { return (int)((4 * max_coeffs + kDefChunk - 1) / kDefChunk); }

unsigned long long deflate_workspace_bytes(int streams, int max_chunks)
{
    const unsigned long long n = (unsigned long long)streams * max_chunks;
    return 256 + ((n * sizeof(int) + 255) & ~255ull) + ((n * 2 * sizeof(unsigned) + 255) & ~255ull) + ((n * kDefThreads * sizeof(unsigned short) + 255) & ~255ull) +
           ((n + 255) & ~255ull);
}

static DeflateStreams deflate_args(const int *coeffs, const long long *counts, int batch, long long coeff_stride, const long long *coeff_off, int max_chunks)
{
    DeflateStreams S;
    memset(&S, 0, sizeof S);
    S.coeffs = coeffs; S.counts = counts; S.coeff_stride = coeff_stride;
    for (int l = 0; l < 3; l++) S.coeff_off[l] = coeff_off[l];
    S.max_chunks = max_chunks;
    (void)batch;
    return S;
}

void launch_deflate_hist(hipStream_t st, const int *coeffs, const long long *counts, int batch, long long coeff_stride, const long long *coeff_off,
                         int max_chunks, int *hist)
{
    DeflateStreams S = deflate_args(coeffs, counts, batch, coeff_stride, coeff_off, max_chunks);
    S.hist = hist;
    (void)hipMemsetAsync(hist, 0, 3 * kDefHistBins * sizeof(int), st);
    if (max_chunks > 0) hipLaunchKernelGGL(k_deflate_hist, dim3(max_chunks, batch * 3), dim3(kDefThreads), 0, st, S);
}

void launch_deflate(hipStream_t st, const int *coeffs, const long long *counts, int batch, long long coeff_stride, const long long *coeff_off,
                    int max_chunks, const unsigned *tables, unsigned char *out, unsigned long long stream_stride, long long *sizes, void *workspace)
{
    DeflateStreams S = deflate_args(coeffs, counts, batch, coeff_stride, coeff_off, max_chunks);
    S.tables = tables;
    S.out = out; S.stream_stride = stream_stride; S.sizes = sizes;
    const unsigned long long n = (unsigned long long)batch * 3 * max_chunks;
    char *w = static_cast<char *>(workspace);
    S.error = reinterpret_cast<int *>(w); w += 256;
    S.chunk_bytes = reinterpret_cast<int *>(w); w += (n * sizeof(int) + 255) & ~255ull;
    S.chunk_adler = reinterpret_cast<unsigned *>(w); w += (n * 2 * sizeof(unsigned) + 255) & ~255ull;
    S.sub_bits = reinterpret_cast<unsigned short *>(w); w += (n * kDefThreads * sizeof(unsigned short) + 255) & ~255ull;
    S.chunk_fixed = reinterpret_cast<unsigned char *>(w);
    (void)hipMemsetAsync(S.error, 0, 256, st);
    if (max_chunks > 0) hipLaunchKernelGGL(k_deflate_sizes, dim3(max_chunks, batch * 3), dim3(kDefThreads), 0, st, S);
    hipLaunchKernelGGL(k_deflate_scan, dim3(batch * 3), dim3(256), 0, st, S);
    if (max_chunks > 0) hipLaunchKernelGGL(k_deflate_emit, dim3(max_chunks, batch * 3), dim3(kDefThreads), 0, st, S);
}

}  // namespace aej
