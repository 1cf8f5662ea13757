"""CPU: the float32 quantiser of the HIP DCT kernels (csrc/dct.hip quantise_f32) against np.round(float32 / int32)
(src/jpeg/jpeg.py:499-502: float64 quotient, round half to even).  tests/native/quantise_check.c restates the kernel's
instruction sequence (mul by an approximate reciprocal, rint, fma remainder, compare with q / 2, ties to even) in IEEE float32 C,
perturbs the reciprocal by +-4 ulp (the hardware's v_rcp_f32 is a 1-ulp approximation) and compares with
rint((double)y / q) on random coefficients, exact ties, their float neighbours and out-of-range inputs (which must take the
float64 fallback).  The GPU parity tests then cover the kernel's copy of the sequence end to end."""
import os
import re
import subprocess

from conftest import ROOT


def test_float32_quantiser_equals_float64_round(tmp_path):
    exe = str(tmp_path / "quantise_check")
    src = os.path.join(ROOT, "tests", "native", "quantise_check.c")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-o", exe, src, "-lm"])
    r = subprocess.run([exe, "8000000"], capture_output=True, text=True, timeout=600)
    m = re.search(r"cases (\d+) mismatches (\d+) fallbacks (\d+)", r.stdout)
    assert r.returncode == 0 and m, (r.stdout, r.stderr[-2000:])
    assert int(m.group(1)) > 5e7 and int(m.group(2)) == 0
    assert 0 < int(m.group(3)) < int(m.group(1)) // 10        # the guard fires on the adversarial ranges only


def test_kernel_and_check_share_the_constants():
    """the range guards of the kernel (2^22 quantiser, 2^17 / 2^18 coefficient / quotient) are the ones the C check assumes"""
    k = open(os.path.join(ROOT, "adaptive_edge_aware_jpeg_amd", "csrc", "dct.hip")).read()
    c = open(os.path.join(ROOT, "tests", "native", "quantise_check.c")).read()
    assert "(1 << 22)" in k and "(1 << 22)" in c
    assert "262144.0f" in k and "262144.0f" in c and "131072.0f" in k
