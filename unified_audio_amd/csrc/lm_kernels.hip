// lm_kernels.hip - kernels specific to the UniSE AR-LM generate loop:
// prompt assembly, KV-cache append, token embedding gather, range-restricted greedy arg-max, the skinny-M
// weight-streaming GEMM used by every decode step, and single-query attention over the KV cache.
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

// K / V parts of a fused [B*n, 3d] QKV buffer (RoPE already applied) -> caches [B, max_len, d] at positions pos0..
__global__ __launch_bounds__(256) void kv_store_kernel(const float* __restrict__ qkv, float* __restrict__ kc,
                                                       float* __restrict__ vc, int B, int n, int pos0, int max_len,
                                                       int d) {
    const int d4 = d >> 2;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)B * n * d4) return;
    const int c = (int)(gid % d4) * 4;
    const long long row = gid / d4;
    const int b = (int)(row / n), t = (int)(row % n);
    const float* src = qkv + row * 3 * d + d + c;
    const long long dst = ((long long)b * max_len + pos0 + t) * d + c;
    *reinterpret_cast<float4*>(kc + dst) = *reinterpret_cast<const float4*>(src);
    *reinterpret_cast<float4*>(vc + dst) = *reinterpret_cast<const float4*>(src + d);
}

int launch_kv_store(const float* qkv, float* kc, float* vc, int B, int n, int pos0, int max_len, int d, hipStream_t s) {
    const long long total = (long long)B * n * (d / 4);
    hipLaunchKernelGGL(kv_store_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, qkv, kc, vc, B, n, pos0,
                       max_len, d);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

__global__ __launch_bounds__(256) void embed_kernel(const long long* __restrict__ tok, const float* __restrict__ table,
                                                    float* __restrict__ x, int B, int d) {
    const int d4 = d >> 2;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= B * d4) return;
    const int b = gid / d4, c = (gid % d4) * 4;
    *reinterpret_cast<float4*>(x + (long long)b * d + c) =
        *reinterpret_cast<const float4*>(table + tok[b] * (long long)d + c);
}
int launch_embed(const long long* tok, const float* table, float* x, int B, int d, hipStream_t s) {
    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)ceil_div((long long)B * (d / 4), 256)), dim3(256), 0, s, tok, table, x,
                       B, d);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

__global__ void fill_i64_kernel(long long* p, long long v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
int launch_fill_i64(long long* p, long long v, int n, hipStream_t s) {
    hipLaunchKernelGGL(fill_i64_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, p, v, n);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// Greedy sampling restricted to a vocabulary slice (llm_sft.py:150-153,180-182 + llm.py:286): logits [B, width] are the
// head outputs of tokens lo..lo+width-1 only.  First maximum wins.  tok[b] = lo + argmax; ids[b*ids_ld + col] = argmax
// (the offset-subtracted id the reference returns) when ids != nullptr.
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, int width, long long ld, int lo,
                                                     long long* __restrict__ tok, long long* __restrict__ ids,
                                                     long long ids_ld, int col) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = logits + (long long)b * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < width; i += 256) {
        const float v = row[i];
        if (v > best) {  // ascending i per thread: strict '>' keeps the first maximum
            best = v;
            bi = i;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(best, o, 64);
        const int i2 = __shfl_xor(bi, o, 64);
        if (v2 > best || (v2 == best && i2 < bi)) {
            best = v2;
            bi = i2;
        }
    }
    if (lane == 0) {
        sv[wave] = best;
        si[wave] = bi;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) {
                best = sv[w];
                bi = si[w];
            }
        tok[b] = lo + bi;
        if (ids) ids[(long long)b * ids_ld + col] = bi;
    }
}
int launch_argmax(const float* logits, int B, int width, long long ld, int lo, long long* tok, long long* ids,
                  long long ids_ld, int col, hipStream_t s) {
    hipLaunchKernelGGL(argmax_kernel, dim3(B), dim3(256), 0, s, logits, width, ld, lo, tok, ids, ids_ld, col);
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

// ------------------------------------------------------------------------------------------------
// Single-query attention over the KV cache (decode step): one workgroup per (batch, head), 4 waves split the keys.
// A wave covers 16 keys per iteration: lane = (key = lane >> 2, 16-float slice = lane & 3), i.e. 64 contiguous bytes per
// lane and 256 contiguous bytes per key, scores are finished with two shuffles, softmax is online per wave, and the
// four partial (m, l, o) states are merged through LDS.
template <int HD, int NW>
__global__ __launch_bounds__(NW * 64) void attention_decode_kernel(const float* __restrict__ q, long long ldq,
                                                               const float* __restrict__ kc,
                                                               const float* __restrict__ vc, long long kv_bstride,
                                                               long long ldkv, float* __restrict__ out, long long ldo,
                                                               int n_keys, float scale) {
    constexpr int SL = HD / 4;  // floats per lane slice
    __shared__ float s_m[NW], s_l[NW];
    __shared__ float s_o[NW][HD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, h = blockIdx.x;
    const int kl = lane >> 2, sl = lane & 3;
    const float* qp = q + (long long)b * ldq + h * HD + sl * SL;
    float qv[SL];
#pragma unroll
    for (int i = 0; i < SL; i += 4) {
        const float4 t = *reinterpret_cast<const float4*>(qp + i);
        qv[i] = t.x * scale; qv[i + 1] = t.y * scale; qv[i + 2] = t.z * scale; qv[i + 3] = t.w * scale;
    }
    const float* kb = kc + (long long)b * kv_bstride + h * HD + sl * SL;
    const float* vb = vc + (long long)b * kv_bstride + h * HD + sl * SL;
    float m_run = -INFINITY, l_run = 0.f;
    float o[SL];
#pragma unroll
    for (int i = 0; i < SL; ++i) o[i] = 0.f;
    for (int k0 = wave * 16; k0 < n_keys; k0 += NW * 16) {
        const int key = k0 + kl;
        const bool ok = key < n_keys;
        const int kk = ok ? key : n_keys - 1;
        float sdot = 0.f;
        const float* kp = kb + (long long)kk * ldkv;
#pragma unroll
        for (int i = 0; i < SL; i += 4) {
            const float4 t = *reinterpret_cast<const float4*>(kp + i);
            sdot = fmaf(qv[i], t.x, sdot);
            sdot = fmaf(qv[i + 1], t.y, sdot);
            sdot = fmaf(qv[i + 2], t.z, sdot);
            sdot = fmaf(qv[i + 3], t.w, sdot);
        }
        sdot += __shfl_xor(sdot, 1, 64);
        sdot += __shfl_xor(sdot, 2, 64);
        const float sc = ok ? sdot : -INFINITY;
        float tmax = sc;
#pragma unroll
        for (int of = 4; of < 64; of <<= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, of, 64));
        const float m_new = fmaxf(m_run, tmax);  // the first tile of every wave with k0 < n_keys has a valid key
        const float alpha = expf(m_run - m_new);
        const float p = ok ? expf(sc - m_new) : 0.f;
        float psum = p;
#pragma unroll
        for (int of = 4; of < 64; of <<= 1) psum += __shfl_xor(psum, of, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        const float* vp = vb + (long long)kk * ldkv;
#pragma unroll
        for (int i = 0; i < SL; i += 4) {
            const float4 t = *reinterpret_cast<const float4*>(vp + i);
            o[i] = fmaf(p, t.x, o[i] * alpha);
            o[i + 1] = fmaf(p, t.y, o[i + 1] * alpha);
            o[i + 2] = fmaf(p, t.z, o[i + 2] * alpha);
            o[i + 3] = fmaf(p, t.w, o[i + 3] * alpha);
        }
    }
    // sum the 16 key-lanes that share a slice
#pragma unroll
    for (int i = 0; i < SL; ++i) {
#pragma unroll
        for (int of = 4; of < 64; of <<= 1) o[i] += __shfl_xor(o[i], of, 64);
    }
    if (lane < 4) {
#pragma unroll
        for (int i = 0; i < SL; ++i) s_o[wave][lane * SL + i] = o[i];
        if (lane == 0) {
            s_m[wave] = m_run;
            s_l[wave] = l_run;
        }
    }
    __syncthreads();
    if (tid < HD) {
        float m = s_m[0];
        for (int w = 1; w < NW; ++w) m = fmaxf(m, s_m[w]);
        float l = 0.f, acc = 0.f;
        for (int w = 0; w < NW; ++w) {
            const float f = (s_m[w] == -INFINITY) ? 0.f : expf(s_m[w] - m);
            l += s_l[w] * f;
            acc += s_o[w][tid] * f;
        }
        out[(long long)b * ldo + h * HD + tid] = acc / l;
    }
}

int launch_attention_decode(const float* q, long long ldq, const float* kc, const float* vc, long long kv_bstride,
                            long long ldkv, float* out, long long ldo, int B, int H, int hd, int n_keys, float scale,
                            hipStream_t s) {
    QA_REQUIRE(n_keys >= 1, "attention_decode: empty cache");
    switch (hd) {
        case 64:
            hipLaunchKernelGGL((attention_decode_kernel<64, 16>), dim3(H, B), dim3(1024), 0, s, q, ldq, kc, vc, kv_bstride, ldkv, out,
                               ldo, n_keys, scale);
            break;
        case 128:
            hipLaunchKernelGGL((attention_decode_kernel<128, 16>), dim3(H, B), dim3(1024), 0, s, q, ldq, kc, vc, kv_bstride, ldkv,
                               out, ldo, n_keys, scale);
            break;
        case 32:
            hipLaunchKernelGGL((attention_decode_kernel<32, 16>), dim3(H, B), dim3(1024), 0, s, q, ldq, kc, vc, kv_bstride, ldkv, out,
                               ldo, n_keys, scale);
            break;
        default: set_error("attention_decode: head_dim=%d unsupported", hd); return QA_ERR_UNSUPPORTED;
    }
    QA_LAUNCH_CHECK();
    return QA_OK;
}

}  // namespace qa
