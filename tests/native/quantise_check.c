/* Brute-force check of the float32 quantiser used by the HIP DCT kernels (csrc/dct.hip quantise()) against the reference
 * arithmetic np.round(float32 / int32) = rint((double)y / (double)q) (src/jpeg/jpeg.py:499-502).  Built and run by
 * tests/test_quantiser_arith.py (CPU).  The candidate below is the same sequence of IEEE-754 single operations the kernel
 * executes (mul, rint, fma, compares); the reciprocal is perturbed by up to +-4 ulp to model the hardware's approximate
 * v_rcp_f32. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static int reference(float y, int q) { return (int)rint((double)y / (double)q); }

/* returns 0 and sets *out, or 1 when the kernel would take its float64 fallback */
static int candidate(float y, int q, float rq, int *out)
{
    if (q > (1 << 22)) return 1;
    const float qf = (float)q;
    const float t = y * rq;
    if (!(fabsf(t) < 262144.0f)) return 1;
    const float k = rintf(t);
    const float r = fmaf(-k, qf, y);          /* exact remainder y - k q */
    const float h = 0.5f * qf, ar = fabsf(r);
    int ki = (int)k;
    if (ar > h || (ar == h && (ki & 1))) ki += r > 0.f ? 1 : -1;
    *out = ki;
    return 0;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rng(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static long long bad = 0, fallbacks = 0, cases = 0;
static void check(float y, int q)
{
    if (!isfinite(y)) return;
    const float exact_rq = 1.0f / (float)q;
    for (int d = -4; d <= 4; d += 2) {
        float rq = u2f(f2u(exact_rq) + d);
        int got;
        cases++;
        if (candidate(y, q, rq, &got)) { fallbacks++; continue; }
        if (got != reference(y, q)) {
            if (bad < 10) fprintf(stderr, "MISMATCH y=%a q=%d rq_ulps=%d got=%d want=%d\n", y, q, d, got, reference(y, q));
            bad++;
        }
    }
}

int main(int argc, char **argv)
{
    const long long n = argc > 1 ? atoll(argv[1]) : 20000000;
    /* 1. random coefficients of the magnitudes the path produces (|y| <= 127 * 256) against the quantisers it builds (1 .. 6050)
     *    and against arbitrary ones */
    for (long long i = 0; i < n; i++) {
        uint64_t r = rng();
        int q = (i & 1) ? 1 + (int)(r % 6050) : 1 + (int)((r >> 20) % (1u << (1 + (r >> 58) % 23)));
        float mag = (float)((r >> 8) & 0xffffff) * (1.0f / 16777216.0f);
        float y = ldexpf(mag, (int)((r >> 40) % 40) - 20) * ((r & 1) ? 1.f : -1.f);
        check(y, q);
    }
    /* 2. exact ties and their float neighbours: y = (k + 0.5) q, +-1 ulp */
    for (long long i = 0; i < n / 4; i++) {
        uint64_t r = rng();
        int q = 1 + (int)(r % 6050);
        int k = (int)((r >> 16) % 40000) - 20000;
        float y = ((float)k + 0.5f) * (float)q;
        check(y, q);
        check(u2f(f2u(y) + 1), q);
        check(u2f(f2u(y) - 1), q);
        check(nextafterf((float)k * (float)q, INFINITY), q);
    }
    /* 3. every float in a few binades against small quantisers */
    for (int q = 1; q <= 64; q += (q < 16 ? 1 : 7))
        for (uint32_t u = f2u(0.25f); u < f2u(64.0f); u += 97) { check(u2f(u), q); check(-u2f(u), q); }
    /* 4. huge quotients / quantisers: must either be right or fall back */
    for (long long i = 0; i < n / 16; i++) {
        uint64_t r = rng();
        int q = 1 + (int)(r % 0x7fffffff);
        float y = u2f((uint32_t)(r >> 32));
        check(y, q);
        check(y, 1 + (int)(r % 3));
    }
    printf("cases %lld mismatches %lld fallbacks %lld\n", cases, bad, fallbacks);
    return bad ? 1 : 0;
}
