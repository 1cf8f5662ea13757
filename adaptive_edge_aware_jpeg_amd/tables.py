"""Host-side tables of the codec (tiny, exact, built once per settings change).

Mirrors ``JpegCompressionSettings`` constants and ``Jpeg.precompute_caches`` of the reference
(src/jpeg/jpeg.py:40-59, 62-147, 216-238, 688-766).  Integer / dyadic-float arithmetic only, done with
NumPy in the reference's own dtypes so the quantisation matrices are the reference's bit for bit.
"""
import math

import numpy as np

# jpeg.py:40-49 / 50-59 (standard JPEG Annex K tables)
LUMINANCE_QUANTIZATION_MATRIX = np.array([
    [16, 11, 10, 16, 24, 40, 51, 61], [12, 12, 14, 19, 26, 58, 60, 55], [14, 13, 16, 24, 40, 57, 69, 56],
    [14, 17, 22, 29, 51, 87, 80, 62], [18, 22, 37, 56, 68, 109, 103, 77], [24, 35, 55, 64, 81, 104, 113, 92],
    [49, 64, 78, 87, 103, 121, 120, 101], [72, 92, 95, 98, 112, 100, 103, 99]], dtype=np.float32)
CHROMINANCE_QUANTIZATION_MATRIX = np.array([
    [17, 18, 24, 47, 99, 99, 99, 99], [18, 21, 26, 66, 99, 99, 99, 99], [24, 26, 56, 99, 99, 99, 99, 99],
    [47, 66, 99, 99, 99, 99, 99, 99], [99] * 8, [99] * 8, [99] * 8, [99] * 8], dtype=np.float32)

_R22 = np.array([[1, 1], [2, 2], [2, 2]])
_R14 = np.array([[1, 1], [1, 4], [1, 4]])
# jpeg.py:62-147: per-space (rows, cols) down-sampling ratios; layer 0 uses the luminance table
DOWNSAMPLING_RATIOS = {
    "ICaCb": _R14, "ICtCp": _R14, "JzAzBz": _R22, "OKLAB": _R22, "YCbCr": _R22, "YCoCg": _R22, "YCoCg-R": _R22,
}


def block_sizes(block_size_range):
    """jpeg.py:219"""
    lo, hi = block_size_range
    return [2 ** i for i in range(int(math.log2(lo)), int(math.log2(hi)) + 1)]


def zigzag_ordering(size):
    """Indices that flatten a size x size block in zigzag order (jpeg.py:726-766)."""
    if not isinstance(size, (int, np.integer)) or size < 0:
        raise ValueError("Block size must be a non-negative integer")
    r, c = np.divmod(np.arange(size * size), size) if size else (np.zeros(0, int), np.zeros(0, int))
    d = r + c
    # within an anti-diagonal: odd d runs top->bottom (row ascending), even d bottom->top
    key = np.where(d % 2 == 1, r, -r)
    order = np.lexsort((key, d))
    return order.astype(np.int32)


def quality_factor(block_size, block_size_range, quality_range):
    """jpeg.py:688-705"""
    bmin, bmax = block_size_range
    qmin, qmax = quality_range
    if bmin == bmax:
        return int((qmin + qmax) / 2)
    return int(qmin + (qmax - qmin) * (1 - math.log(block_size / bmin) / math.log(bmax / bmin)))


def _bilinear_axis(n_src, n_dst):
    """source index pairs and float32 weights of cv.resize(INTER_LINEAR) along one axis (half-pixel centres,
    clamped at both ends)."""
    scale = n_src / n_dst
    f = ((np.arange(n_dst) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    s[lo], f[lo] = 0, 0
    hi = s >= n_src - 1
    s[hi], f[hi] = n_src - 1, 0
    return s, np.minimum(s + 1, n_src - 1), (np.float32(1) - f).astype(np.float32), f


def resize_table(table, size):
    """cv.resize(table, (size, size), interpolation=INTER_LINEAR) for the 8x8 float32 table (jpeg.py:722)."""
    table = np.asarray(table, dtype=np.float32)
    n = table.shape[0]
    if size == n:
        return table.copy()
    if 2 * size == n:      # OpenCV turns an exact 2x INTER_LINEAR shrink into the 2x2 box mean
        return ((table[0::2, 0::2] + table[0::2, 1::2]) + (table[1::2, 0::2] + table[1::2, 1::2])) * np.float32(0.25)
    i0, i1, a0, a1 = _bilinear_axis(n, size)
    rows = table[:, i0] * a0[None, :] + table[:, i1] * a1[None, :]
    return (rows[i0, :] * a0[:, None] + rows[i1, :] * a1[:, None]).astype(np.float32)


def quantization_matrix(default_matrix, size, quality):
    """Jpeg._get_quantization_matrix (jpeg.py:707-724)"""
    scale_factor = 5000 / quality if quality < 50 else 200 - 2 * quality
    scaled = np.floor((scale_factor * default_matrix + 50) / 100)
    resized = resize_table(scaled, size)
    return np.clip(resized, 1, None).astype(np.int32)


def largest_power_of_2(n):
    """jpeg/utils.py:24-41 (largest power of two strictly below n for n > 2)"""
    if n <= 0:
        raise ValueError("n must be positive.")
    if n <= 2:
        return n
    return 2 ** math.floor(math.log2(n - 1))
