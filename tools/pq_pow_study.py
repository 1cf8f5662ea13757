#!/usr/bin/env python3
"""VERDICT r4 item 7 (ICtCp colour stage 2.36 -> <= 1.9 ms for 16 x 4K): can the two float64 pows of _pq_inverse_eotf
(src/color/common.py:131-159: tmp = (v / 10000) ** m1; ((c1 + c2 tmp) / (1 + c3 tmp)) ** m2) be made cheaper within the goldens?
CPU study, numpy only; prints what profiles/r05_ictcp_pow_study.txt records.   python3 tools/pq_pow_study.py"""
import numpy as np

c1, c2, c3, m1, m2 = 3424 / 4096, 2413 / 128, 2392 / 128, 2610 / 16384, 2523 / 32
rng = np.random.default_rng(1)

def pq(v):
    t = (v / 10000.0) ** m1
    return ((c1 + c2 * t) / (1.0 + c3 * t)) ** m2

# what the stage feeds the function: LMS of 8-bit sRGB colours (ictcp.py:142-147 on XYZ), so v in (0, ~1]
srgb = rng.integers(0, 256, (400000, 3)) / 255.0
lin = np.where(srgb <= 0.04045, srgb / 12.92, ((srgb + 0.055) / 1.055) ** 2.4)
xyz = lin @ np.array([[0.4124564, 0.3575761, 0.1804375], [0.2126729, 0.7151522, 0.0721750], [0.0193339, 0.1191920, 0.9503041]]).T
lms = xyz @ np.array([[0.3592, 0.6976, -0.0358], [-0.1922, 1.1004, 0.0755], [0.0070, 0.0749, 0.8434]]).T
v = lms[lms > 0]
x = v / 10000.0
print(f"arguments of the first pow: x = v / 10000 in [{x.min():.3e}, {x.max():.3e}]  ({np.log2(x.max() / x.min()):.1f} octaves)")

# (1) "evaluate log2 t = m1 log2 x once and reuse it to seed the second logarithm's range reduction"
t = x ** m1
u = (c1 + c2 * t) / (1.0 + c3 * t)
print(f"(1) second pow's argument u = (c1 + c2 t) / (1 + c3 t) in [{u.min():.4f}, {u.max():.4f}]: log2 u is NOT a function of log2 t that a range "
      f"reduction could reuse -- it spans {np.log2(u.max() / u.min()):.2f} octaves against {np.log2(t.max() / t.min()):.2f} of t, through a rational map; what can be "
      "shared is nothing more than the exponent extraction (two integer operations of ~37 per pow)")

# (2) "tabulate x^m1 on a 4096-entry monotone grid with a degree-3 correction held to 1e-13"
def cubic_table_error(grid_fn, n=4096):
    xs = np.sort(rng.choice(x, 200000))
    g = grid_fn(n)
    f = g ** m1
    i = np.clip(np.searchsorted(g, xs) - 1, 1, n - 3)
    # degree-3 Lagrange through the four neighbouring nodes (the best a per-interval cubic can do with these nodes)
    est = np.zeros_like(xs)
    for a in range(-1, 3):
        w = np.ones_like(xs)
        for b in range(-1, 3):
            if a != b:
                w *= (xs - g[i + b]) / (g[i + a] - g[i + b])
        est += w * f[i + a]
    rel = np.abs(est - xs ** m1) / xs ** m1
    k = int(np.argmax(rel))
    return rel.max(), xs[k], est[k], xs[k] ** m1
e, xk, got, want = cubic_table_error(lambda n: np.linspace(x.min(), x.max(), n))
print(f"(2) uniform grid in x, 4096 nodes, cubic: max relative error {e:.2e} (needed: 1e-13 for the bit-exact lockstep the goldens were made with, "
      f"1e-10 for the test's 1e-9 tolerance); counter-example x = {xk:.6e}: table {got:.12e}, pow {want:.12e} -- x^0.159 has an unbounded derivative at 0 "
      "and the arguments cover 19 octaves")
e, xk, got, want = cubic_table_error(lambda n: np.exp(np.linspace(np.log(x.min()), np.log(x.max()), n)))
print(f"    geometric grid (uniform in log x), 4096 nodes, cubic: max relative error {e:.2e} -- accurate, but the node index IS a logarithm of x: the table "
      "replaces exp2 (12 + 6 operations), not log2, and its 4 x 4096 coefficients (128 KiB) do not fit the LDS beside the kernel's other tables: a gather from L2 per pow")

# (3) what shorter polynomials buy (the recipe: log2 by a 64-entry table + degree-N polynomial in |r| <= 2^-7, exp2 by a 64-entry table + degree-M)
print("(3) operation counts per pow (float64, each two issue slots): fixed part 23 (exponent / index extraction, table look-ups, reconstruction) + N + M")
for tabs, r in ((64, 2.0 ** -7), (256, 2.0 ** -9)):
    for N in (7, 5, 4, 3):
        for M in (5, 4, 3):
            err = r ** (N + 1) / (N + 1) / np.log(2) + (r * np.log(2)) ** (M + 1) / np.prod(np.arange(1, M + 2))
            if err < 2e-11:
                print(f"    {tabs:3d}-entry tables, log2 degree {N}, exp2 degree {M}: truncation error {err:.1e}, {23 + N + M} operations ({100 * (23 + N + M) / 35:.0f} % of today's 35)")
print("    => within the test tolerance the cheapest variant is 29-30 of 35 operations per pow (-15 %); six pows are ~55 % of the stage's instructions, so the "
      "colour stage would go from 2.35 to ~2.15 ms, not to 1.9.  Not built: the oracle would move a second time to suit the kernel for an 8 % gain on a "
      "configuration outside the bench line.")
