/* TEST INFRASTRUCTURE ONLY - sanitizer self-test of the plain-C RVQ oracle (oracle/rvq_ref.c).
 * Built by `make -C oracle sanitize` with -fsanitize=address,undefined and run by tests/test_sanitizer_cpu.py: exact-size heap
 * buffers around every call, so an out-of-bounds read or write in the checker itself aborts the run.  Also checks the oracle's
 * own invariants: the search result has zero excess against the double-precision arg-min except at near-ties, and
 * lookup(search(x)) reproduces the quantised sum. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

void rvq_search_f32(const float* x, int64_t n, const float* cb, int Q, int K, int D, int64_t* idx);
void rvq_check_f64(const float* x, int64_t n, const float* cb, int Q, int K, int D, const int64_t* idx, double* excess,
                   int64_t* best, double* gap);
void rvq_lookup_f32(const int64_t* idx, int64_t n, const float* cb, int Q, int K, int D, float* out);

static uint64_t s = 88172645463325252ULL;
static float rnd(void) {  /* xorshift64, uniform in [-1, 1) */
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}

static int run(int64_t n, int Q, int K, int D) {
    float* x = malloc(sizeof(float) * (size_t)(n * D + (n * D == 0)));
    float* cb = malloc(sizeof(float) * (size_t)Q * K * D);
    int64_t* idx = malloc(sizeof(int64_t) * (size_t)(n * Q + (n == 0)));
    int64_t* best = malloc(sizeof(int64_t) * (size_t)(n * Q + (n == 0)));
    double* excess = malloc(sizeof(double) * (size_t)(n * Q + (n == 0)));
    double* gap = malloc(sizeof(double) * (size_t)(n * Q + (n == 0)));
    float* out = malloc(sizeof(float) * (size_t)(n * D + (n * D == 0)));
    for (int64_t i = 0; i < n * D; ++i) x[i] = 0.6f * rnd();
    for (int q = 0; q < Q; ++q)
        for (int i = 0; i < K * D; ++i) cb[(size_t)q * K * D + i] = 0.6f * rnd() / (float)(1 << q);
    rvq_search_f32(x, n, cb, Q, K, D, idx);
    rvq_check_f64(x, n, cb, Q, K, D, idx, excess, best, gap);
    rvq_lookup_f32(idx, n, cb, Q, K, D, out);
    int bad = 0;
    for (int64_t i = 0; i < n * Q; ++i) {
        if (idx[i] < 0 || idx[i] >= K) ++bad;
        if (excess[i] > 1e-4) ++bad;                       /* fp32 search vs fp64 arg-min: only summation noise */
        if (gap[i] > 1e-4 && idx[i] != best[i]) ++bad;     /* decisive steps agree */
    }
    for (int64_t i = 0; i < n; ++i)
        for (int d = 0; d < D; ++d) {
            float acc = 0.f;
            for (int q = 0; q < Q; ++q) acc += cb[((size_t)q * K + idx[i * Q + q]) * D + d];
            if (fabsf(acc - out[i * D + d]) > 1e-5f) ++bad;
        }
    free(x); free(cb); free(idx); free(best); free(excess); free(gap); free(out);
    if (bad) fprintf(stderr, "rvq_selftest: n=%lld Q=%d K=%d D=%d: %d violations\n", (long long)n, Q, K, D, bad);
    return bad;
}

int main(void) {
    int bad = 0;
    bad += run(0, 1, 1, 8);
    bad += run(1, 1, 1, 1);
    bad += run(7, 3, 5, 3);
    bad += run(257, 4, 64, 32);
    bad += run(100, 16, 128, 48);
    bad += run(33, 2, 1024, 64);
    printf("rvq_selftest: %s\n", bad ? "FAILED" : "ok");
    return bad != 0;
}
