"""GPU (-m gpu): the reference's own unit tests, re-stated against this package.

test/unit_tests/test_color_conversions.py:27-78  -> full 256^3 sRGB grid through every colour space and back:
    max and mean abs error < 1e-4, each direction < 1500 ms.
test/unit_tests/test_compression_speed.py:32-100 -> lena, YCoCg, quality (75,75), fixed blocks 4/8/16/32, 3 iterations of
    compress + decompress with a fresh decoder (the reference only prints timings; here the round trip is also checked).
"""
import os
import time

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import torch
    assert torch.cuda.is_available()
    import adaptive_edge_aware_jpeg_amd as pkg
    return pkg


@pytest.fixture(scope="module")
def grid():
    g = np.array(np.meshgrid(np.arange(256), np.arange(256), np.arange(256), indexing="ij")).reshape(3, -1).T / 255.0
    return g          # float64, exactly as test_color_conversions.py:31-33 builds it


def test_color_space_error_and_performance(A, grid):
    A.convert("sRGB", "YCbCr", grid[:1024])        # warm-up (library load, context)
    for space in A.get_color_spaces():
        t0 = time.perf_counter()
        conv = A.convert("sRGB", space, grid)
        fwd_ms = (time.perf_counter() - t0) * 1000
        t0 = time.perf_counter()
        back = A.convert(space, "sRGB", conv)
        bwd_ms = (time.perf_counter() - t0) * 1000
        err = np.abs(grid - back)
        assert err.max() < 1e-4, f"Max error for {space} too high: {err.max()}"
        assert err.mean() < 1e-4
        assert fwd_ms < 1500 and bwd_ms < 1500, (space, fwd_ms, bwd_ms)     # includes the PCIe copies of 16.7 M colours


@pytest.mark.parametrize("block", [4, 8, 16, 32])
def test_compression_round_trip_fixed_blocks(A, lena, block):
    img = A.Image.load(os.path.join(GOLDEN, "lena.png"))
    codec = A.Jpeg(A.JpegCompressionSettings("YCoCg", (75, 75), (block, block)))
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        data = codec.compress(img)
        t1 = time.perf_counter()
        out = A.Jpeg(A.JpegCompressionSettings()).decompress(data)
        t2 = time.perf_counter()
        times.append(((t1 - t0) * 1000, (t2 - t1) * 1000))
    print(f"block {block}: compress {np.mean([t[0] for t in times]):.1f} ms, decompress {np.mean([t[1] for t in times]):.1f} ms, "
          f"{len(data)} bytes")
    assert out.data.shape == img.data.shape and out.extension == ".png"
    psnr = 10 * np.log10(1.0 / np.mean((out.data - img.data) ** 2))
    assert psnr > 30.0, psnr
