// rvq.hip - multi-stage residual vector quantisation: nearest-codebook search and index look-up.
//
// Reference: the third-party vector_quantize_pytorch.ResidualVQ called at QuarkAudio-HCodec/HCodec-1.0/vq/codec.py:171-172
// and :183-184; arithmetic as stated in-tree by vq/core_vq.py:223-231 (dist = |x|^2 - 2 x.e + |e|^2, arg-max of the negative,
// first maximum wins), :394-404 (r <- r - e_idx per stage) and :406-412 (decode = sum of look-ups).
//
// Main path (D a multiple of 32): per stage the distance products  S = R E_q^T  ([n_vec, K], K = 1024 codes x D = 512) are a
// plain contraction and run on the library's implicit-GEMM kernel (conv_gemm.hip: code rows and residual rows staged through
// LDS, fp32 MFMA, >= 256 workgroups even for ~1000 vectors because the tile grid covers vectors x codes) - with the ARG-MIN IN ITS
// EPILOGUE (ConvParams::am_*, round 4): dist = (|r|^2 - 2 s) + |e|^2 with the reference's association is formed on the tile while
// it sits in LDS and every group of 32 codes is reduced to its (dist, index) winner, lowest index on ties, so S itself never
// reaches memory (round 3 wrote and re-read 4 KB per vector and stage; now 256 B).  Then ONE wave per vector (rvq_pick_kernel)
// folds the K / 32 group winners in ascending code order, writes the index, subtracts the chosen code from the residual and
// leaves the new |r|^2 for the next stage.  The stages are sequential by definition (core_vq.py:394-404), so a stage = 2
// launches; vectors are processed in chunks of 16384.
// Fallback (other D): all Q stages in ONE launch - a workgroup keeps the residuals of 32 vectors in LDS (fp32), each of its 4
// waves sweeps a quarter of the codebook 32 codes at a time with v_mfma_f32_32x32x2_f32 (A = residual tile from LDS, B = code
// rows straight from L2), keeps a per-lane running arg-min, then the winners are reduced with wave shuffles and one LDS
// exchange, ties resolving to the lowest index, and the residual update is fused.  It only uses n_vec / 32 workgroups.
#include <cstdlib>

#include "kernels.h"

namespace qa {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void argmin_merge(float& d, int& i, float d2, int i2) {
    if (d2 < d || (d2 == d && i2 < i)) {
        d = d2;
        i = i2;
    }
}

__global__ __launch_bounds__(256) void rvq_search_kernel(const float* __restrict__ x, long long n_vec,
                                                         const float* __restrict__ cb, const float* __restrict__ e2,
                                                         int Q, int K, int D, long long* __restrict__ indices) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LD = D + 4;
    float* sR = smem;                                   // [32][LD] residuals
    float* sX2 = smem + 32 * LD;                        // [32]
    float* sCd = sX2 + 32;                              // [4][32] candidate distance per wave
    int* sCi = reinterpret_cast<int*>(sCd + 4 * 32);    // [4][32] candidate index per wave
    int* sBest = sCi + 4 * 32;                          // [32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jl = lane & 31, hh = lane >> 5;
    const long long v0 = (long long)blockIdx.x * 32;
    const int d4 = D >> 2;

    for (int i = tid; i < 32 * d4; i += 256) {
        const int r = i / d4, c = (i % d4) * 4;
        const long long v = v0 + r;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (v < n_vec) t = *reinterpret_cast<const float4*>(x + v * D + c);
        *reinterpret_cast<float4*>(sR + r * LD + c) = t;
    }
    __syncthreads();

    for (int q = 0; q < Q; ++q) {
        const float* cbq = cb + (long long)q * K * D;
        const float* e2q = e2 + (long long)q * K;
        // |r|^2 per vector: wave w handles vectors w*8 .. w*8+7
        for (int r = wave * 8; r < wave * 8 + 8; ++r) {
            float s = 0.f;
            for (int c = lane * 4; c < D; c += 256) {
                const float4 t = *reinterpret_cast<const float4*>(sR + r * LD + c);
                s += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
            }
            s = wave_sum(s);
            if (lane == 0) sX2[r] = s;
        }
        __syncthreads();

        float best_d[16];
        int best_i[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            best_d[r] = INFINITY;
            best_i[r] = 0x7fffffff;
        }
        const float* ap = sR + jl * LD + 4 * hh;
        for (int tile = wave; tile * 32 < K; tile += 4) {
            const int code = tile * 32 + jl;
            const int code_c = code < K ? code : K - 1;
            const float* bp = cbq + (long long)code_c * D + 4 * hh;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 4
            for (int g = 0; g < D; g += 8) {
                const float4 a = *reinterpret_cast<const float4*>(ap + g);
                const float4 b = *reinterpret_cast<const float4*>(bp + g);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            }
            if (code < K) {
                const float ee = e2q[code];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int vi = (r & 3) + 8 * (r >> 2) + 4 * hh;
                    const float dist = (sX2[vi] - 2.f * acc[r]) + ee;  // same association as core_vq.py:225-229
                    if (dist < best_d[r]) {  // codes ascend per lane: strict '<' keeps the first minimum
                        best_d[r] = dist;
                        best_i[r] = code;
                    }
                }
            }
        }
        // reduce over the 32 lanes (codes) that share a half
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) {
                const float d2 = __shfl_xor(best_d[r], o, 64);
                const int i2 = __shfl_xor(best_i[r], o, 64);
                argmin_merge(best_d[r], best_i[r], d2, i2);
            }
            if (jl == 0) {
                const int vi = (r & 3) + 8 * (r >> 2) + 4 * hh;
                sCd[wave * 32 + vi] = best_d[r];
                sCi[wave * 32 + vi] = best_i[r];
            }
        }
        __syncthreads();
        if (tid < 32) {
            float d = sCd[tid];
            int i = sCi[tid];
            for (int w = 1; w < 4; ++w) argmin_merge(d, i, sCd[w * 32 + tid], sCi[w * 32 + tid]);
            sBest[tid] = i;
            if (v0 + tid < n_vec) indices[(v0 + tid) * Q + q] = i;
        }
        __syncthreads();
        // residual update r <- r - e[idx]
        for (int i = tid; i < 32 * d4; i += 256) {
            const int r = i / d4, c = (i % d4) * 4;
            const float4 e = *reinterpret_cast<const float4*>(cbq + (long long)sBest[r] * D + c);
            float4 t = *reinterpret_cast<float4*>(sR + r * LD + c);
            t.x -= e.x; t.y -= e.y; t.z -= e.z; t.w -= e.w;
            *reinterpret_cast<float4*>(sR + r * LD + c) = t;
        }
        __syncthreads();
    }
}

// residual workspace of a chunk: R = x, x2 = |x|^2 (one wave per vector)
__global__ __launch_bounds__(256) void rvq_prep_kernel(const float* __restrict__ x, long long n, int D, float* __restrict__ R,
                                                       float* __restrict__ x2) {
    const int lane = threadIdx.x & 63;
    const long long v = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n) return;
    float s = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const float4 t = *reinterpret_cast<const float4*>(x + v * D + c);
        *reinterpret_cast<float4*>(R + v * D + c) = t;
        s += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
    }
    s = wave_sum(s);
    if (lane == 0) x2[v] = s;
}

// One wave per vector: arg-min over the K / 32 group winners the distance GEMM's epilogue left (dist = (|r|^2 - 2 r.e) + |e|^2,
// core_vq.py:225-229 association; a group's winner is its lowest index of minimal distance, and groups ascend with the lane), lowest
// index on ties (torch.max returns the first maximum of the negated distance); then r <- r - e[idx], |r|^2 refreshed.
__global__ __launch_bounds__(256) void rvq_pick_kernel(const float* __restrict__ pd, const int* __restrict__ pi, int n_groups,
                                                       float* __restrict__ x2, const float* __restrict__ cbq, float* __restrict__ R,
                                                       long long n, int D, long long* __restrict__ indices, int Q, int q) {
    const int lane = threadIdx.x & 63;
    const long long v = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n) return;
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int g = lane; g < n_groups; g += 64) {  // groups ascend per lane: strict '<' keeps the first minimum
        const float d = pd[v * n_groups + g];
        const int i = pi[v * n_groups + g];
        if (d < best) {
            best = d;
            bi = i;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float d2 = __shfl_xor(best, o, 64);
        const int i2 = __shfl_xor(bi, o, 64);
        argmin_merge(best, bi, d2, i2);
    }
    if (bi == 0x7fffffff) bi = 0;  // all-NaN row: stay inside the codebook
    if (lane == 0) indices[v * Q + q] = bi;
    const float* e = cbq + (long long)bi * D;
    float s = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        float4 t = *reinterpret_cast<float4*>(R + v * D + c);
        const float4 ev = *reinterpret_cast<const float4*>(e + c);
        t.x -= ev.x; t.y -= ev.y; t.z -= ev.z; t.w -= ev.w;
        *reinterpret_cast<float4*>(R + v * D + c) = t;
        s += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
    }
    s = wave_sum(s);
    if (lane == 0) x2[v] = s;
}

// |e|^2 of every code vector: one wave per code
__global__ __launch_bounds__(256) void rvq_norms_kernel(const float* __restrict__ cb, float* __restrict__ e2, int QK,
                                                        int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= QK) return;
    float s = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const float4 t = *reinterpret_cast<const float4*>(cb + (long long)row * D + c);
        s += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
    }
    s = wave_sum(s);
    if (lane == 0) e2[row] = s;
}

// out[v, :] = ((E_0[i_0] + E_1[i_1]) + ...) in stage order, like the reference's running sum
__global__ __launch_bounds__(256) void rvq_lookup_kernel(const long long* __restrict__ indices, long long n_vec,
                                                         const float* __restrict__ cb, int Q, int K, int D,
                                                         float* __restrict__ out, long long ldo) {
    const int d4 = D >> 2;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_vec * d4) return;
    const long long v = gid / d4;
    const int c = (int)(gid % d4) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = 0; q < Q; ++q) {
        long long idx = indices[v * Q + q];
        // -1 = a DROPPED code: upstream's get_codes_from_indices masks it to a zero vector (vector_quantize_pytorch residual_vq.py:
        // `mask = indices == -1 ... all_codes.masked_fill(mask, 0.)`; the quantize-dropout convention) - the stage contributes nothing
        if (idx == -1) continue;
        idx = idx < 0 ? 0 : (idx >= K ? K - 1 : idx);  // memory safety only; callers validate
        const float4 e = *reinterpret_cast<const float4*>(cb + ((long long)q * K + idx) * D + c);
        acc.x += e.x; acc.y += e.y; acc.z += e.z; acc.w += e.w;
    }
    *reinterpret_cast<float4*>(out + v * ldo + c) = acc;
}

int launch_rvq_norms(const float* codebooks, float* e2, int QK, int D, hipStream_t s) {
    QA_REQUIRE(D % 4 == 0, "rvq: D=%d must be a multiple of 4", D);
    hipLaunchKernelGGL(rvq_norms_kernel, dim3((unsigned)ceil_div(QK, 4)), dim3(256), 0, s, codebooks, e2, QK, D);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

constexpr long long RVQ_CHUNK = 16384;

static bool rvq_gemm_ok(int K, int D) { return D % 32 == 0 && K % 4 == 0 && K >= 32; }

size_t rvq_scratch_floats(long long n_vec, int K, int D) {
    if (!rvq_gemm_ok(K, D) || n_vec <= 0) return 0;
    const long long ch = std::min(n_vec, RVQ_CHUNK);
    return (size_t)ch * ((size_t)K + D + 1) + 64;
}

int launch_rvq_search(const float* x, long long n_vec, const float* codebooks, const float* e2, int Q, int K, int D,
                      long long* indices, float* quantized, long long ldq, float* scratch, hipStream_t s) {
    QA_REQUIRE(D % 8 == 0 && D >= 8, "rvq_search: D=%d must be a multiple of 8", D);
    QA_REQUIRE(Q >= 1 && K >= 1, "rvq_search: Q=%d K=%d", Q, K);
    if (n_vec <= 0) return QA_OK;
    if (rvq_gemm_ok(K, D)) {
        QA_REQUIRE(scratch != nullptr, "rvq_search: the GEMM path needs rvq_scratch_floats() floats of workspace");
        const long long ch = std::min(n_vec, RVQ_CHUNK);
        const int ng = (K + 31) / 32;
        float* R = scratch;                  // [ch, D] residuals
        float* pd = R + (size_t)ch * D;      // [ch, K / 32] group winners: distance ...
        int* pi = reinterpret_cast<int*>(pd + (size_t)ch * ng);  // ... and code index
        float* x2 = pd + (size_t)ch * K;     // [ch]  (the layout rvq_scratch_floats has always promised: [ch, K] behind R)
        for (long long v0 = 0; v0 < n_vec; v0 += ch) {
            const long long n = std::min(ch, n_vec - v0);
            const unsigned grid = (unsigned)ceil_div(n, 4);
            hipLaunchKernelGGL(rvq_prep_kernel, dim3(grid), dim3(256), 0, s, x + v0 * D, n, D, R, x2);
            QA_LAUNCH_CHECK();
            for (int q = 0; q < Q; ++q) {
                const float* cbq = codebooks + (long long)q * K * D;
                qa_conv_args a{};
                a.x = R; a.w = cbq; a.y = pd;  // y is not written in arg-min mode (any non-null aligned pointer)
                a.B = 1; a.T_in = n; a.C_in = D; a.T_out = n; a.N = K;
                a.ldx = D; a.ldy = K; a.ldr = K; a.ldg = K;
                a.ksize = 1; a.stride = 1;
                ConvParams p;
                QA_TRY(conv_params_from_args(a, &p));
                p.am_x2 = x2; p.am_e2 = e2 + (long long)q * K; p.am_dist = pd; p.am_idx = pi; p.am_ld = ng;
                QA_TRY(launch_conv_gemm(p, s));
                HbmProf prof_(HK_RVQ_PICK, (double)n * (8.0 * ng + 12.0 * D + 16.0), s);  // group winners, residual read + write, code row
                hipLaunchKernelGGL(rvq_pick_kernel, dim3(grid), dim3(256), 0, s, pd, pi, ng, x2, cbq, R, n, D, indices + v0 * Q, Q, q);
                QA_LAUNCH_CHECK();
            }
        }
    } else {
        const size_t lds = (size_t)(32 * (D + 4) + 32 + 4 * 32) * sizeof(float) + (4 * 32 + 32) * sizeof(int);
        QA_REQUIRE(lds <= 160 * 1024, "rvq_search: D=%d needs %zu B of LDS", D, lds);
        QA_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(rvq_search_kernel), 160 * 1024));
        hipLaunchKernelGGL(rvq_search_kernel, dim3((unsigned)ceil_div(n_vec, 32)), dim3(256), lds, s, x, n_vec, codebooks,
                           e2, Q, K, D, indices);
        QA_LAUNCH_CHECK();
    }
    if (quantized) return launch_rvq_lookup(indices, n_vec, codebooks, Q, K, D, quantized, ldq, s);
    return QA_OK;
}

int launch_rvq_lookup(const long long* indices, long long n_vec, const float* codebooks, int Q, int K, int D, float* out,
                      long long ldo, hipStream_t s) {
    QA_REQUIRE(D % 4 == 0, "rvq_lookup: D=%d must be a multiple of 4", D);
    if (n_vec <= 0) return QA_OK;
    HbmProf prof_(HK_RVQ_LOOKUP, (double)n_vec * (8.0 * Q + 4.0 * D * (Q + 1)), s);  // indices + Q gathered code rows + the sum
    hipLaunchKernelGGL(rvq_lookup_kernel, dim3((unsigned)ceil_div(n_vec * (D / 4), 256)), dim3(256), 0, s, indices, n_vec,
                       codebooks, Q, K, D, out, ldo);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

}  // namespace qa
