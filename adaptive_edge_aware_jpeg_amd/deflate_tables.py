"""Host side of the OPT-IN GPU entropy stage (csrc/deflate.hip): Huffman code tables and block headers for the deflate blocks the
kernels write.  The kernels are table-driven -- a table holds, for one layer of a batch, the bit-reversed code and length of every
literal / length symbol and of the two distance symbols the parser uses (distance 1 and 4), plus the bits every block starts with:

* ``fixed_table()``: RFC 1951's fixed code (block header 3 bits) -- no statistics needed;
* ``adaptive_table(litlen_hist, dist_hist)``: a length-limited Huffman code built from the symbol histogram the kernels counted
  (``aej_deflate_histogram``), with the dynamic-block header (HLIT / HDIST / HCLEN, run-length coded code lengths) that transmits it.

Layout of a table (``TABLE_WORDS`` uint32, mirrored by ``DefTable`` in deflate.hip):
``[0 .. 285]`` literal / length symbols: ``reversed_code | nbits << 16``;  ``[286], [287]`` distance symbols 0 and 3 likewise;
``[288]`` number of header bits;  ``[289 ..]`` the header bits, LSB first, 32 per word.

The reference writes these streams with ``zlib.compress(level=9)`` (src/jpeg/jpeg.py:588-590) and reads them with ``zlib.decompress``
(jpeg.py:659), which accepts any conforming stream: ``encode_reference()`` below is a pure-Python restatement of the kernels' parser and
bit packer, used by the CPU tests to check tables and headers against ``zlib.decompress`` without a GPU.
"""
import heapq

import numpy as np

N_LITLEN = 286
HEADER_WORDS = 96
TABLE_WORDS = 289 + HEADER_WORDS
CL_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]      # RFC 1951 3.2.7


_REV15 = None


def _rev(code, n):
    """the n-bit code with its bits in reverse order (the stream is LSB first, Huffman codes go in MSB first)"""
    global _REV15
    if _REV15 is None:                       # all 15-bit reversals, built once (vectorised)
        v = np.arange(1 << 15, dtype=np.uint32)
        r = np.zeros_like(v)
        for b in range(15):
            r |= ((v >> b) & 1) << (14 - b)
        _REV15 = r.tolist()
    return _REV15[code] >> (15 - n) if n else 0


def huffman_lengths(counts, limit):
    """Code lengths of a Huffman code for `counts` (symbols with count 0 get length 0), none longer than `limit`: plain Huffman, then the
    classic repair -- clamp, and while the Kraft sum exceeds 1 lengthen the rarest symbol that can still be lengthened."""
    counts = [int(c) for c in counts]
    used = [i for i, c in enumerate(counts) if c > 0]
    lengths = [0] * len(counts)
    if not used:
        return lengths
    if len(used) == 1:
        lengths[used[0]] = 1
        return lengths
    # two-queue construction: leaves sorted by (count, symbol), internal nodes appear in non-decreasing weight order
    leaves = sorted(used, key=lambda i: (counts[i], i))
    n = len(leaves)
    weight = [counts[i] for i in leaves] + [0] * (n - 1)      # nodes 0 .. n-1 = leaves (in sorted order), n .. 2n-2 = internal
    parent = [0] * (2 * n - 1)
    li, ii, nxt = 0, n, n
    while nxt < 2 * n - 1:
        picked = []
        for _ in range(2):
            if li < n and (ii >= nxt or weight[li] <= weight[ii]):
                picked.append(li); li += 1
            else:
                picked.append(ii); ii += 1
        weight[nxt] = weight[picked[0]] + weight[picked[1]]
        parent[picked[0]] = parent[picked[1]] = nxt
        nxt += 1
    depth = [0] * (2 * n - 1)
    for node in range(2 * n - 3, -1, -1):                        # parents have larger indices: depths top-down
        depth[node] = depth[parent[node]] + 1
    for k, i in enumerate(leaves):
        lengths[i] = max(depth[k], 1)
    for i in used:
        lengths[i] = min(lengths[i], limit)
    kraft = sum(1 << (limit - lengths[i]) for i in used)
    by_rarity = sorted(used, key=lambda i: (counts[i], -lengths[i]))
    while kraft > (1 << limit):
        for i in by_rarity:
            if lengths[i] < limit:
                kraft -= 1 << (limit - lengths[i] - 1)
                lengths[i] += 1
                break
    # give back what the repair left over to the most frequent symbols
    for i in sorted(used, key=lambda i: -counts[i]):
        while lengths[i] > 1 and kraft + (1 << (limit - lengths[i])) <= (1 << limit):
            kraft += 1 << (limit - lengths[i])
            lengths[i] -= 1
    return lengths


def canonical_codes(lengths):
    """RFC 1951 3.2.2: codes of one length are consecutive, in symbol order."""
    max_len = max(lengths) if lengths else 0
    bl_count = [0] * (max_len + 2)
    for l in lengths:
        if l:
            bl_count[l] += 1
    code, next_code = 0, [0] * (max_len + 2)
    for bits in range(1, max_len + 1):
        code = (code + bl_count[bits - 1]) << 1
        next_code[bits] = code
    out = [0] * len(lengths)
    for i, l in enumerate(lengths):
        if l:
            out[i] = next_code[l]
            next_code[l] += 1
    return out


class _Bits:
    def __init__(self):
        self.v, self.n = 0, 0

    def put(self, value, nbits):
        self.v |= (int(value) & ((1 << nbits) - 1)) << self.n
        self.n += nbits

    def put_code(self, code, nbits):          # Huffman codes go in most significant bit first
        self.put(_rev(code, nbits), nbits)


def _pack(table_lit, table_dist, header):
    t = np.zeros(TABLE_WORDS, np.uint32)
    for i, (code, n) in enumerate(table_lit):
        t[i] = _rev(code, n) | (n << 16)
    for k, (code, n) in enumerate(table_dist):
        t[286 + k] = _rev(code, n) | (n << 16)
    if header.n > HEADER_WORDS * 32:
        raise ValueError("block header longer than the table holds")
    t[288] = header.n
    for w in range((header.n + 31) // 32):
        t[289 + w] = (header.v >> (32 * w)) & 0xFFFFFFFF
    return t


def fixed_table():
    lit = []
    for s in range(N_LITLEN):
        if s < 144:
            lit.append((0x30 + s, 8))
        elif s < 256:
            lit.append((0x190 + s - 144, 9))
        elif s < 280:
            lit.append((s - 256, 7))
        else:
            lit.append((0xC0 + s - 280, 8))
    h = _Bits()
    h.put(0, 1)            # BFINAL = 0
    h.put(1, 2)            # BTYPE = 01
    return _pack(lit, [(0, 5), (3, 5)], h)


def adaptive_table(litlen_hist, dist_hist, cover_all=True):
    """litlen_hist: counts of the 286 literal / length symbols over the blocks that will use the table (the end-of-block symbol is added
    here); dist_hist: counts of distance symbols 0 (distance 1) and 3 (distance 4).

    ``cover_all`` (default): every symbol gets a code (count + 1), so the table is valid for ANY input, whatever it was counted on -- at
    the cost of a 15-bit code for symbols that never occur and ~100 bytes of header per 32 KiB block.  ``cover_all=False``: only the
    symbols that occur (and the end-of-block symbol) get codes -- for a table that is used on exactly the data it was counted on, as
    ``Jpeg.deflate_batch`` does (same parser, same data: the kernels fall back to the fixed code for any block that would need a missing
    code, so a mismatch costs size, never correctness); the header shrinks to 40-60 bytes, which is 13 % of the output on very
    compressible data."""
    if cover_all:
        ll = [int(c) + 1 for c in litlen_hist[:N_LITLEN]]
    else:
        ll = [int(c) for c in litlen_hist[:N_LITLEN]]
        ll[256] = max(ll[256], 1)                              # end of block
        if sum(1 for c in ll if c) < 2:                        # (a complete code needs two symbols)
            ll[0 if ll[0] == 0 else 1] = 1
    ll_len = huffman_lengths(ll, 15)
    # both distance symbols always get a code (a block may use either), one bit each: lengths 1, 0, 0, 1 are a complete code
    d_len = [1, 0, 0, 1]
    ll_code, d_code = canonical_codes(ll_len), canonical_codes(d_len)
    # the code lengths, run-length coded with the code-length alphabet (RFC 1951 3.2.7)
    seq = ll_len + d_len
    syms, i = [], 0
    while i < len(seq):
        v, run = seq[i], 1
        while i + run < len(seq) and seq[i + run] == v:
            run += 1
        i += run
        if v == 0:
            while run >= 11:
                r = min(run, 138); syms.append((18, r - 11, 7)); run -= r
            if run >= 3:
                syms.append((17, run - 3, 3)); run = 0
            syms += [(0, 0, 0)] * run
        else:
            syms.append((v, 0, 0)); run -= 1
            while run >= 3:
                r = min(run, 6); syms.append((16, r - 3, 2)); run -= r
            syms += [(v, 0, 0)] * run
    cl_hist = [0] * 19
    for s, _, _ in syms:
        cl_hist[s] += 1
    cl_len = huffman_lengths(cl_hist, 7)
    cl_code = canonical_codes(cl_len)
    hclen = 19
    while hclen > 4 and cl_len[CL_ORDER[hclen - 1]] == 0:
        hclen -= 1
    h = _Bits()
    h.put(0, 1)                      # BFINAL = 0
    h.put(2, 2)                      # BTYPE = 10
    h.put(N_LITLEN - 257, 5)         # HLIT
    h.put(len(d_len) - 1, 5)         # HDIST
    h.put(hclen - 4, 4)              # HCLEN
    for k in range(hclen):
        h.put(cl_len[CL_ORDER[k]], 3)
    for s, extra, ebits in syms:
        h.put_code(cl_code[s], cl_len[s])
        if ebits:
            h.put(extra, ebits)
    lit = [(ll_code[s], ll_len[s]) for s in range(N_LITLEN)]
    return _pack(lit, [(d_code[0], d_len[0]), (d_code[3], d_len[3])], h)


# ---- pure-Python restatement of the kernels (CPU tests only; small inputs) ------------------------------------------------------------
SUB, CHUNK = 128, 32768                 # bytes a thread parses on its own (csrc/deflate.hip kDefSub: keep the two equal); bytes per deflate block


def _length_symbol(L):
    if L == 258:
        return 285, 0, 0
    l = L - 3
    if l < 8:
        return 257 + l, 0, 0
    e = l.bit_length() - 3
    return 257 + 4 * (e + 1) + ((l >> e) & 3), l & ((1 << e) - 1), e


def parse_tokens(data, start, n):
    """The kernels' greedy parse of the sub-block data[start : start + n]: ('lit', byte) and ('match', length, distance in (1, 4))."""
    out, p = [], 0
    while p < n:
        g = start + p
        b = data[g]
        l1 = l4 = 0
        if g >= 1 and b == data[g - 1]:
            l1 = 1
            while p + l1 < n and l1 < 258 and data[g + l1] == b:
                l1 += 1
        if g >= 4 and l1 < 258 and b == data[g - 4]:
            l4 = 1
            while p + l4 < n and l4 < 258 and data[g + l4] == data[g + l4 - 4]:
                l4 += 1
        L = max(l1, l4)
        if L >= 3:
            out.append(("match", L, 1 if l1 >= l4 else 4))
            p += L
        else:
            out.append(("lit", b))
            p += 1
    return out


def histogram_reference(data):
    lit, dist = np.zeros(N_LITLEN, np.int64), np.zeros(2, np.int64)
    for s in range(0, len(data), SUB):
        for tok in parse_tokens(data, s, min(SUB, len(data) - s)):
            if tok[0] == "lit":
                lit[tok[1]] += 1
            else:
                lit[_length_symbol(tok[1])[0]] += 1
                dist[0 if tok[2] == 1 else 1] += 1
    return lit, dist


def encode_reference(data, table):
    """bytes -> zlib stream exactly as csrc/deflate.hip lays it out (chunks of 32 KiB, one block + an empty stored block each)."""
    import zlib
    data = bytes(data)
    hdr_bits = int(table[288])
    hdr = 0
    for w in range((hdr_bits + 31) // 32):
        hdr |= int(table[289 + w]) << (32 * w)
    out = bytearray(b"\x78\x01")
    for c0 in range(0, len(data), CHUNK):
        b = _Bits()
        b.put(hdr, hdr_bits)
        for s in range(c0, min(c0 + CHUNK, len(data)), SUB):
            for tok in parse_tokens(data, s, min(SUB, len(data) - s)):
                if tok[0] == "lit":
                    e = int(table[tok[1]])
                    b.put(e & 0xFFFF, e >> 16)
                else:
                    sym, extra, ebits = _length_symbol(tok[1])
                    e = int(table[sym])
                    b.put(e & 0xFFFF, e >> 16)
                    b.put(extra, ebits)
                    d = int(table[286 if tok[2] == 1 else 287])
                    b.put(d & 0xFFFF, d >> 16)
        e = int(table[256])
        b.put(e & 0xFFFF, e >> 16)               # end of block
        b.put(0, 3)                              # empty stored block: BFINAL = 0, BTYPE = 00
        nbytes = (b.n + 7) // 8
        out += b.v.to_bytes(nbytes, "little") + b"\x00\x00\xff\xff"
    out += b"\x03\x00" + zlib.adler32(data).to_bytes(4, "big")
    return bytes(out)
