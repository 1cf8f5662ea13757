"""Readable restatement of the host side of the OPT-IN GPU entropy stage (csrc/deflate.hip) -- TEST INFRASTRUCTURE, not product: the
package calls the library's C++ helper ``aej_deflate_build_tables``; the CPU tests compare that helper with ``adaptive_table`` below word
for word, and drive tables and block headers through ``zlib.decompress`` with a small token-level encoder.

A table holds, for one layer of a batch, the bit-reversed code and length of every literal / length symbol and of the thirty distance
symbols, plus the bits the stream's one deflate block starts with (layout: include/aej.h, aej_deflate_build_tables):
``[0 .. 285]`` literal / length symbols ``reversed_code | nbits << 16``; ``[286 .. 315]`` distance symbols likewise; ``[316]`` number of
header bits; ``[317 ..]`` the header bits, LSB first, 32 per word.

The reference writes these streams with ``zlib.compress(level=9)`` (src/jpeg/jpeg.py:588-590) and reads them with ``zlib.decompress``
(jpeg.py:659), which accepts any conforming stream.
"""
import numpy as np

N_LITLEN, N_DIST = 286, 30
HIST_BINS = 320
HEADER_WORDS = 131
TABLE_WORDS = N_LITLEN + N_DIST + 1 + HEADER_WORDS
HDR_BITS_AT, HDR_AT = N_LITLEN + N_DIST, N_LITLEN + N_DIST + 1
CL_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]      # RFC 1951 3.2.7


def _rev(code, n):
    """the n-bit code with its bits in reverse order (the stream is LSB first, Huffman codes go in MSB first)"""
    r = 0
    for _ in range(n):
        r = (r << 1) | (code & 1)
        code >>= 1
    return r


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
        t[N_LITLEN + k] = _rev(code, n) | (n << 16)
    if header.n > HEADER_WORDS * 32:
        raise ValueError("block header longer than the table holds")
    t[HDR_BITS_AT] = header.n
    for w in range((header.n + 31) // 32):
        t[HDR_AT + w] = (header.v >> (32 * w)) & 0xFFFFFFFF
    return t


def fixed_table():
    """RFC 1951's fixed code in the table layout (the kernels hold it in code; here for the token-level encoder)."""
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
    h.put(1, 1)            # BFINAL = 1
    h.put(1, 2)            # BTYPE = 01
    return _pack(lit, [(d, 5) for d in range(N_DIST)], h)


def adaptive_table(litlen_hist, dist_hist, cover_all=True):
    """litlen_hist: counts of the 286 literal / length symbols of the streams that will use the table; dist_hist: counts of the 30
    distance symbols.  ``cover_all``: every symbol gets a code (count + 1), so the table is valid for ANY input; otherwise only the
    symbols that occur (and end-of-block) -- a stream that needs a missing code is written with the fixed code by the kernels, so a
    mismatch costs size, never correctness."""
    add = 1 if cover_all else 0
    ll = [int(c) + add for c in litlen_hist[:N_LITLEN]]
    dd = [int(c) + add for c in dist_hist[:N_DIST]]
    if not cover_all:
        ll[256] = max(ll[256], 1)                              # end of block
        if sum(1 for c in ll if c) < 2:                        # (a complete code needs two symbols)
            ll[0 if ll[0] == 0 else 1] = 1
        if not any(dd):
            dd[0] = 1                                          # RFC 1951: no distance code at all is written as one code of length 1
    ll_len, d_len = huffman_lengths(ll, 15), huffman_lengths(dd, 15)
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
    h.put(1, 1)                      # BFINAL = 1: a stream is one block
    h.put(2, 2)                      # BTYPE = 10
    h.put(N_LITLEN - 257, 5)         # HLIT
    h.put(N_DIST - 1, 5)             # HDIST
    h.put(hclen - 4, 4)              # HCLEN
    for k in range(hclen):
        h.put(cl_len[CL_ORDER[k]], 3)
    for s, extra, ebits in syms:
        h.put_code(cl_code[s], cl_len[s])
        if ebits:
            h.put(extra, ebits)
    return _pack([(ll_code[s], ll_len[s]) for s in range(N_LITLEN)], [(d_code[s], d_len[s]) for s in range(N_DIST)], h)


# ---- a small token-level encoder (CPU tests only): any valid LZ77 parse of `data` + a table -> a zlib stream ----------------------------
def length_symbol(L):
    if L == 258:
        return 285, 0, 0
    l = L - 3
    if l < 8:
        return 257 + l, 0, 0
    e = l.bit_length() - 3
    return 257 + 4 * (e + 1) + ((l >> e) & 3), l & ((1 << e) - 1), e


def distance_symbol(D):
    x = D - 1
    if x < 4:
        return x, 0, 0
    e = x.bit_length() - 2
    return 2 * (e + 1) + ((x >> e) & 1), x & ((1 << e) - 1), e


def greedy_tokens(data, window=4096, min_len=3):
    """A plain greedy parse with a small brute-force search (test inputs are a few KiB): ('lit', byte) / ('match', length, distance)."""
    out, p, n = [], 0, len(data)
    while p < n:
        best_l, best_d = 0, 0
        for q in range(max(0, p - window), p):
            l = 0
            while p + l < n and l < 258 and data[q + l] == data[p + l]:
                l += 1
            if l > best_l:
                best_l, best_d = l, p - q
        if best_l >= min_len:
            out.append(("match", best_l, best_d))
            p += best_l
        else:
            out.append(("lit", data[p]))
            p += 1
    return out


def histogram_of(tokens):
    hist = np.zeros(HIST_BINS, np.int64)
    for tok in tokens:
        if tok[0] == "lit":
            hist[tok[1]] += 1
        else:
            hist[length_symbol(tok[1])[0]] += 1
            hist[N_LITLEN + distance_symbol(tok[2])[0]] += 1
    hist[256] += 1
    return hist


def encode_tokens(data, tokens, table):
    """zlib stream of `data` from its parse: header, the one block the table's header bits open, end of block, Adler-32."""
    import zlib
    b = _Bits()
    hdr_bits = int(table[HDR_BITS_AT])
    hdr = 0
    for w in range((hdr_bits + 31) // 32):
        hdr |= int(table[HDR_AT + w]) << (32 * w)
    b.put(hdr, hdr_bits)
    for tok in tokens:
        if tok[0] == "lit":
            e = int(table[tok[1]])
            assert e >> 16, "no code for this literal"
            b.put(e & 0xFFFF, e >> 16)
        else:
            sym, extra, ebits = length_symbol(tok[1])
            e = int(table[sym])
            assert e >> 16, "no code for this length"
            b.put(e & 0xFFFF, e >> 16)
            b.put(extra, ebits)
            dsym, dextra, debits = distance_symbol(tok[2])
            d = int(table[N_LITLEN + dsym])
            assert d >> 16, "no code for this distance"
            b.put(d & 0xFFFF, d >> 16)
            b.put(dextra, debits)
    e = int(table[256])
    b.put(e & 0xFFFF, e >> 16)               # end of block
    return b"\x78\x01" + b.v.to_bytes((b.n + 7) // 8, "little") + zlib.adler32(bytes(data)).to_bytes(4, "big")
