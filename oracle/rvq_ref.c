/* TEST INFRASTRUCTURE ONLY - plain-C restatement of the residual-VQ arithmetic of the H-Codec hot path.
 *
 * The reference calls the third-party vector_quantize_pytorch.ResidualVQ (==1.22.15, not vendored, not installed;
 * never diffed against that package) at QuarkAudio-HCodec/HCodec-1.0/vq/codec.py:171-172 / :183-184; the same algorithm is stated
 * in-tree by vq/core_vq.py:
 *   :223-231  dist = -(|x|^2 - 2 x.e^T + |e|^2), index = argmax(dist)  (first maximum wins)
 *   :394-404  for each stage: idx = quantize(residual); residual -= E[idx]
 *   :406-412  decode = sum over stages of E_q[idx_q]
 * PINNED to that in-tree statement: tests/test_rvq_pin_cpu.py (bit-exact against core_vq.ResidualVectorQuantization
 * imported from /root/reference, and against tests/golden/rvq_corevq_*.npz produced by it).
 * Nothing in the product path links this file; only tests/, smoke() and bench.py's cpu_baseline leg load it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* fp32 search, sequential summation, same association as core_vq.py:225-229 */
void rvq_search_f32(const float* x, int64_t n, const float* cb, int Q, int K, int D, int64_t* idx) {
    float* r = (float*)malloc(sizeof(float) * D);
    float* e2 = (float*)malloc(sizeof(float) * (size_t)Q * K);
    for (int64_t i = 0; i < (int64_t)Q * K; ++i) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) s += cb[i * D + d] * cb[i * D + d];
        e2[i] = s;
    }
    for (int64_t v = 0; v < n; ++v) {
        memcpy(r, x + v * D, sizeof(float) * D);
        for (int q = 0; q < Q; ++q) {
            const float* e = cb + (int64_t)q * K * D;
            float x2 = 0.f;
            for (int d = 0; d < D; ++d) x2 += r[d] * r[d];
            float best = -INFINITY;
            int bi = 0;
            for (int k = 0; k < K; ++k) {
                float dot = 0.f;
                for (int d = 0; d < D; ++d) dot += r[d] * e[(int64_t)k * D + d];
                const float dist = -((x2 - 2.f * dot) + e2[(int64_t)q * K + k]);
                if (dist > best) { best = dist; bi = k; }
            }
            idx[v * Q + q] = bi;
            for (int d = 0; d < D; ++d) r[d] -= e[(int64_t)bi * D + d];
        }
    }
    free(r);
    free(e2);
}

/* Teacher-forced optimality check in double precision: the residual follows the GIVEN indices (fp32 updates, as any
 * fp32 implementation does); for each (vector, stage) report
 *   excess[v*Q+q] = |r - e_chosen|^2 - min_k |r - e_k|^2   (>= 0, double)
 *   best[v*Q+q]   = argmin_k in double (lowest index on exact ties)
 *   gap[v*Q+q]    = second smallest distance - smallest distance
 */
void rvq_check_f64(const float* x, int64_t n, const float* cb, int Q, int K, int D, const int64_t* idx, double* excess,
                   int64_t* best, double* gap) {
    float* r = (float*)malloc(sizeof(float) * D);
    for (int64_t v = 0; v < n; ++v) {
        memcpy(r, x + v * D, sizeof(float) * D);
        for (int q = 0; q < Q; ++q) {
            const float* e = cb + (int64_t)q * K * D;
            double d1 = INFINITY, d2 = INFINITY, dc = 0.0;
            int b1 = 0;
            const int64_t chosen = idx[v * Q + q];
            for (int k = 0; k < K; ++k) {
                double s = 0.0;
                for (int d = 0; d < D; ++d) {
                    const double t = (double)r[d] - (double)e[(int64_t)k * D + d];
                    s += t * t;
                }
                if (k == chosen) dc = s;
                if (s < d1) { d2 = d1; d1 = s; b1 = k; }
                else if (s < d2) d2 = s;
            }
            excess[v * Q + q] = dc - d1;
            best[v * Q + q] = b1;
            gap[v * Q + q] = d2 - d1;
            const int64_t c = chosen < 0 ? 0 : (chosen >= K ? K - 1 : chosen);
            for (int d = 0; d < D; ++d) r[d] -= e[c * D + d];
        }
    }
    free(r);
}

/* core_vq.py:406-412: out = ((E_0[i_0] + E_1[i_1]) + ...) */
void rvq_lookup_f32(const int64_t* idx, int64_t n, const float* cb, int Q, int K, int D, float* out) {
    for (int64_t v = 0; v < n; ++v)
        for (int d = 0; d < D; ++d) {
            float s = 0.f;
            for (int q = 0; q < Q; ++q) s += cb[((int64_t)q * K + idx[v * Q + q]) * D + d];
            out[v * D + d] = s;
        }
}
