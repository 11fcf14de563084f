// lm_kernels.hip - the UniSE AR-LM's prefill-side kernels: prompt assembly, RoPE + KV-cache append over a whole
// prompt, and the skinny-M weight-streaming GEMM (per-item linears of at most 32 rows: tiny prompts here, BiCodec's d-vector / AdaLN
// linears in bicodec.cpp).  The decode step lives in lm_decode.hip; the round-1 per-op decode kernels (embedding gather, arg-max,
// single-query attention behind QA_LM_UNFUSED) were removed in round 5.
#include "kernels.h"

namespace qa {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// llm_sft.py:110-128: prompt = [task, (enroll_sos, adapter(enroll)), mix_sos, adapter(mix)]  -> x [B, L, d]
__global__ __launch_bounds__(256) void assemble_prompt_kernel(float* __restrict__ x, const float* __restrict__ task_vec,
                                                              const float* __restrict__ enroll_sos,
                                                              const float* __restrict__ enroll_emb,
                                                              const float* __restrict__ mix_sos,
                                                              const float* __restrict__ mix_emb, int B, int Ne, int Nm,
                                                              int d) {
    const int L = 1 + (enroll_emb ? 1 + Ne : 0) + 1 + Nm;
    const int d4 = d >> 2;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)B * L * d4) return;
    const int c = (int)(gid % d4) * 4;
    const int pos = (int)((gid / d4) % L);
    const int b = (int)(gid / ((long long)d4 * L));
    const float* src;
    int p = pos;
    if (p == 0) {
        src = task_vec;
    } else {
        p -= 1;
        if (enroll_emb && p == 0) {
            src = enroll_sos;
        } else {
            if (enroll_emb) p -= 1;
            if (enroll_emb && p < Ne) {
                src = enroll_emb + ((long long)b * Ne + p) * d;
            } else {
                if (enroll_emb) p -= Ne;
                src = (p == 0) ? mix_sos : mix_emb + ((long long)b * Nm + (p - 1)) * d;
            }
        }
    }
    *reinterpret_cast<float4*>(x + ((long long)b * L + pos) * d + c) = *reinterpret_cast<const float4*>(src + c);
}

int launch_assemble_prompt(float* x, const float* task_vec, const float* enroll_sos, const float* enroll_emb,
                           const float* mix_sos, const float* mix_emb, int B, int Ne, int Nm, int d, hipStream_t s) {
    const int L = 1 + (enroll_emb ? 1 + Ne : 0) + 1 + Nm;
    const long long total = (long long)B * L * (d / 4);
    hipLaunchKernelGGL(assemble_prompt_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, x, task_vec,
                       enroll_sos, enroll_emb, mix_sos, mix_emb, B, Ne, Nm, d);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// ------------------------------------------------------------------------------------------------
// Skinny-M GEMM for the decode steps: y[M <= 16*MT, N] = epi(x[M, K] W[N, K]^T).  Weight streaming is the whole cost
// (226 MB of fp32 weights per step, SURVEY.md hard part 2), so: one workgroup per 16 output columns, its 8 waves split K,
// every lane streams 32 B-contiguous pieces of its weight row straight from HBM into v_mfma_f32_16x16x4_f32 (the batch
// rows are the M side), partial tiles are reduced through LDS in a fixed order (deterministic, no atomics) and the same
// fused epilogue as conv_gemm is applied.
// Two fusions remove launches from the decode step:
//   rms_eps > 0: the input is the un-normalised residual stream; RMSNorm's weight has been folded into W on the host
//                (W' = W diag(w)), so y = rstd[m] * (x W'^T) and only the per-row rstd is computed here (every workgroup
//                recomputes it from the 16 x K input it reads anyway);
//   DUAL:        W holds, per 16-column group, 16 "gate" rows followed by 16 "up" rows; y = silu(gate) * up (SwiGLU).
template <int MT, bool DUAL>
__global__ __launch_bounds__(512) void skinny_gemm_kernel(const float* __restrict__ x, long long ldx,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          const float* __restrict__ gate, long long ldg,
                                                          const float* __restrict__ res, long long ldr,
                                                          float* __restrict__ y, long long ldy, int M, int N, int K,
                                                          int act, float rms_eps) {
    constexpr int NA = DUAL ? 2 : 1;
    __shared__ float part[8][NA][MT][16][17];
    __shared__ float s_rstd[MT * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int kw = K / 8, k0 = wave * kw;
    if (rms_eps > 0.f) {
        for (int r = wave; r < MT * 16; r += 8) {
            const float* xr = x + (long long)min(r, M - 1) * ldx;
            float sq = 0.f;
            for (int c = lane * 4; c < K; c += 256) {
                const float4 t = *reinterpret_cast<const float4*>(xr + c);
                sq += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
            }
            sq = wave_sum(sq);
            if (lane == 0) s_rstd[r] = rsqrtf(sq / K + rms_eps);
        }
    }
    const int nrow = min(n0 + li, N - 1);
    const float* wp[NA];
    wp[0] = w + (long long)(DUAL ? (blockIdx.x * 32 + li) : nrow) * K + k0 + 8 * kq;
    if (DUAL) wp[NA - 1] = w + (long long)(blockIdx.x * 32 + 16 + li) * K + k0 + 8 * kq;
    const float* xp[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) xp[m] = x + (long long)min(m * 16 + li, M - 1) * ldx + k0 + 8 * kq;
    f32x4 acc[NA][MT];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[a][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int g = 0; g < kw; g += 32) {
        float4 w0[NA], w1[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            w0[a] = *reinterpret_cast<const float4*>(wp[a] + g);
            w1[a] = *reinterpret_cast<const float4*>(wp[a] + g + 4);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float4 a0 = *reinterpret_cast<const float4*>(xp[m] + g);
            const float4 a1 = *reinterpret_cast<const float4*>(xp[m] + g + 4);
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, w0[a].x, acc[a][m], 0, 0, 0);
                acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, w0[a].y, acc[a][m], 0, 0, 0);
                acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, w0[a].z, acc[a][m], 0, 0, 0);
                acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, w0[a].w, acc[a][m], 0, 0, 0);
                acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, w1[a].x, acc[a][m], 0, 0, 0);
                acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, w1[a].y, acc[a][m], 0, 0, 0);
                acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, w1[a].z, acc[a][m], 0, 0, 0);
                acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, w1[a].w, acc[a][m], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[wave][a][m][4 * kq + r][li] = acc[a][m][r];
    __syncthreads();
    for (int i = tid; i < MT * 256; i += 512) {
        const int row = i >> 4, col = i & 15;
        const int n = n0 + col;
        if (row >= M || n >= N) continue;
        float v = 0.f, u = 0.f;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) {
            v += part[wv][0][row >> 4][row & 15][col];
            if (DUAL) u += part[wv][NA - 1][row >> 4][row & 15][col];
        }
        if (rms_eps > 0.f) {
            v *= s_rstd[row];
            u *= s_rstd[row];
        }
        if (bias) v += bias[n];
        if (DUAL) v = silu_f(v) * u;
        if (gate) v = silu_f(gate[(long long)row * ldg + n]) * v;
        v = apply_act(v, act);
        if (res) v += res[(long long)row * ldr + n];
        y[(long long)row * ldy + n] = v;
    }
}

int launch_skinny_gemm(const float* x, long long ldx, const float* w, const float* bias, const float* gate,
                       long long ldg, const float* res, long long ldr, float* y, long long ldy, int M, int N, int K,
                       int act, hipStream_t s, float rms_eps, int dual) {
    QA_REQUIRE(M >= 1 && M <= 32, "skinny_gemm: M=%d must be in [1, 32]", M);
    QA_REQUIRE(K % 256 == 0 && (ldx % 4) == 0, "skinny_gemm: K=%d must be a multiple of 256", K);
    QA_REQUIRE(!dual || N % 16 == 0, "skinny_gemm: dual mode needs N %% 16 == 0");
    const dim3 grid((unsigned)ceil_div(N, 16));
#define QA_SK(MT, DUAL)                                                                                                  \
    hipLaunchKernelGGL((skinny_gemm_kernel<MT, DUAL>), grid, dim3(512), 0, s, x, ldx, w, bias, gate, ldg, res, ldr, y, ldy, M, N, \
                       K, act, rms_eps)
    if (M <= 16) {
        if (dual) QA_SK(1, true); else QA_SK(1, false);
    } else {
        if (dual) QA_SK(2, true); else QA_SK(2, false);
    }
#undef QA_SK
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// RoPE (rotate-half, position pos0 + t) on the q and k parts of a fused [B*n, 3d] buffer in place, and append k, v to the caches
__global__ __launch_bounds__(256) void rope_kv_kernel(float* __restrict__ qkv, const float* __restrict__ cs,
                                                      float* __restrict__ kc, float* __restrict__ vc, int B, int n, int H,
                                                      int hd, int pos0, int max_len) {
    const int half = hd >> 1, d = H * hd;
    const long long total = (long long)B * n * H * half;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int i = (int)(gid % half);
    const int h = (int)((gid / half) % H);
    const long long row = gid / ((long long)half * H);
    const int b = (int)(row / n), t = (int)(row % n);
    const int pos = pos0 + t;
    const float c = cs[((long long)pos * half + i) * 2], sn = cs[((long long)pos * half + i) * 2 + 1];
    float* q = qkv + row * 3 * d + h * hd;
    const float q1 = q[i], q2 = q[i + half];
    q[i] = q1 * c - q2 * sn;
    q[i + half] = q2 * c + q1 * sn;
    const float* k = q + d;
    const float* v = q + 2 * d;
    const long long dst = ((long long)b * max_len + pos) * d + h * hd;
    const float k1 = k[i], k2 = k[i + half];
    kc[dst + i] = k1 * c - k2 * sn;
    kc[dst + i + half] = k2 * c + k1 * sn;
    vc[dst + i] = v[i];
    vc[dst + i + half] = v[i + half];
}
int launch_rope_kv(float* qkv, const float* cs, float* kc, float* vc, int B, int n, int H, int hd, int pos0, int max_len,
                   hipStream_t s) {
    const long long total = (long long)B * n * H * (hd / 2);
    hipLaunchKernelGGL(rope_kv_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, qkv, cs, kc, vc, B, n, H, hd, pos0,
                       max_len);
    QA_LAUNCH_CHECK();
    return QA_OK;
}


}  // namespace qa
