// ssl_kernels.hip - the pieces of the SSL front-end (HuBERT / wav2vec 2.0 feature extractor + encoder) that are not a
// contraction over >= 32 channels; everything else of that model runs on conv_gemm / attention / rownorm.
//
// Reference call sites: QuarkAudio-HCodec/HCodec-1.0/audio_tokenizer.py:35-48 (extract_wav2vec2_features: zero-pad 160,
// AutoModel(...)(wavs, output_hidden_states=True), mean of the hidden states, sign * |x|^0.3) and
// HCodec-1.5/audio_tokenizer.py:53-67 (hidden states 11, 14, 16).  The model body is third-party (transformers
// HubertModel / Wav2Vec2Model); its published algorithm is restated in oracle/ssl_ref.py.
#include "kernels.h"

namespace qa {

// ---- feature-extractor layer 0: Conv1d(1 -> C0, k, stride, no padding) over the zero-padded waveform ---------------------
// "group" flavour (HuBERT base, wav2vec2 base): GroupNorm(num_groups = C0) = per-(clip, channel) normalisation over TIME, then
// GELU.  The [B, T1, C0] activation is 2 GB at 32 x 10 s, so the conv output is never stored un-normalised: pass 1 computes
// it for the statistics only, pass 2 recomputes it (10 FMAs per element) and writes the normalised, activated result once.
// One workgroup = TCH consecutive frames x all channels; the input samples of the chunk and the filters sit in LDS.
constexpr int SSL_TCH = 64;
constexpr int SSL_KMAX = 16;

template <bool STATS>
__global__ __launch_bounds__(256) void ssl_conv0_kernel(const float* __restrict__ wav, const float* __restrict__ w_kc,
                                                        const float* __restrict__ bias, const float* __restrict__ stats,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ y, double* __restrict__ partial, int T, int T1,
                                                        int C0, int ksize, int stride, int pad, int act) {
    extern __shared__ float smem[];
    float* xs = smem;                                   // SSL_TCH * stride + ksize samples
    float* ws = smem + SSL_TCH * stride + SSL_KMAX;     // [ksize][C0]
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int t0 = chunk * SSL_TCH;
    const int nt = min(SSL_TCH, T1 - t0);
    const int span = (nt - 1) * stride + ksize;
    const long long s0 = (long long)t0 * stride - pad;  // first sample of the chunk in un-padded coordinates
    for (int i = tid; i < span; i += 256) {
        const long long s = s0 + i;
        xs[i] = (s >= 0 && s < T) ? wav[(long long)b * T + s] : 0.f;
    }
    for (int i = tid; i < ksize * C0; i += 256) ws[i] = w_kc[i];
    __syncthreads();
    for (int c = tid; c < C0; c += 256) {
        float wr[SSL_KMAX];
#pragma unroll
        for (int j = 0; j < SSL_KMAX; ++j) wr[j] = j < ksize ? ws[j * C0 + c] : 0.f;
        const float bc = bias ? bias[c] : 0.f;
        float mean = 0.f, rstd = 1.f, g = 1.f, be = 0.f;
        if (!STATS && stats) {
            mean = stats[((long long)b * C0 + c) * 2];
            rstd = stats[((long long)b * C0 + c) * 2 + 1];
            g = gamma[c];
            be = beta[c];
        }
        double s1 = 0.0, s2 = 0.0;
        for (int t = 0; t < nt; ++t) {
            const float* xp = xs + t * stride;
            float v = bc;
#pragma unroll
            for (int j = 0; j < SSL_KMAX; ++j)
                if (j < ksize) v = fmaf(xp[j], wr[j], v);
            if (STATS) {
                s1 += v;
                s2 += (double)v * v;
            } else {
                v = (v - mean) * rstd * g + be;
                y[((long long)b * T1 + t0 + t) * C0 + c] = apply_act(v, act);
            }
        }
        if (STATS) {
            double* p = partial + (((long long)b * gridDim.x + chunk) * C0 + c) * 2;
            p[0] = s1;
            p[1] = s2;
        }
    }
}

// per (clip, channel): fold the chunk partials in chunk order (deterministic) -> {mean, 1/sqrt(biased var + eps)}
__global__ void ssl_gn_finalize_kernel(const double* __restrict__ partial, float* __restrict__ stats, int nchunks, int C0, int T1,
                                       float eps) {
    const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C0) return;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < nchunks; ++k) {
        const double* p = partial + (((long long)b * nchunks + k) * C0 + c) * 2;
        s1 += p[0];
        s2 += p[1];
    }
    const double mean = s1 / T1;
    double var = s2 / T1 - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[((long long)b * C0 + c) * 2] = (float)mean;
    stats[((long long)b * C0 + c) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

size_t ssl_conv0_scratch_bytes(int B, int T1, int C0) {
    return (size_t)B * ceil_div(T1, SSL_TCH) * C0 * 2 * sizeof(double) + (size_t)B * C0 * 2 * sizeof(float);
}

// norm_group != 0: GroupNorm(C0 groups) + act fused (two passes over the waveform); else plain conv (+bias) + act
int launch_ssl_conv0(const float* wav, const float* w_kc, const float* bias, const float* gamma, const float* beta, float* y,
                     void* scratch, int B, int T, int T1, int C0, int ksize, int stride, int pad, int norm_group, float eps, int act,
                     hipStream_t s) {
    QA_REQUIRE(ksize <= SSL_KMAX && C0 % 4 == 0, "ssl conv0: ksize %d / C0 %d unsupported", ksize, C0);
    const int nchunks = (int)ceil_div(T1, SSL_TCH);
    const size_t lds = (size_t)(SSL_TCH * stride + SSL_KMAX + ksize * C0) * sizeof(float);
    QA_REQUIRE(lds <= 64 * 1024, "ssl conv0: %zu bytes of LDS", lds);
    double* partial = reinterpret_cast<double*>(scratch);
    float* stats = reinterpret_cast<float*>(partial + (size_t)B * nchunks * C0 * 2);
    if (norm_group) {
        hipLaunchKernelGGL(ssl_conv0_kernel<true>, dim3(nchunks, B), dim3(256), lds, s, wav, w_kc, bias, nullptr, nullptr, nullptr,
                           nullptr, partial, T, T1, C0, ksize, stride, pad, ACT_NONE);
        hipLaunchKernelGGL(ssl_gn_finalize_kernel, dim3((unsigned)ceil_div(C0, 256), B), dim3(256), 0, s, partial, stats, nchunks, C0,
                           T1, eps);
    }
    hipLaunchKernelGGL(ssl_conv0_kernel<false>, dim3(nchunks, B), dim3(256), lds, s, wav, w_kc, bias, norm_group ? stats : nullptr,
                       gamma, beta, y, nullptr, T, T1, C0, ksize, stride, pad, act);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// ---- WavLM: gate of the relative position bias (WavLMAttention.forward steps 1-3) -----------------------------------------
// gate[b, h, i] = a * (bb * const[h] - 1) + 2 with a = sigmoid(wA . x + bA), bb = sigmoid(wB . x + bB), x = hidden[b, i, h*hd : (h+1)*hd];
// wA / wB are the sums of rows 0..3 / 4..7 of gru_rel_pos_linear (the reference sums the 4 outputs before the sigmoid).
// One wave per (row, head).
__global__ __launch_bounds__(256) void ssl_gate_kernel(const float* __restrict__ hidden, const float* __restrict__ wab,
                                                       const float* __restrict__ bab, const float* __restrict__ cst,
                                                       float* __restrict__ gate, int B, int N, int H, int hd) {
    const int lane = threadIdx.x & 63;
    const long long job = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (job >= (long long)B * N * H) return;
    const int h = (int)(job % H);
    const long long row = job / H;
    const float* x = hidden + row * H * hd + (long long)h * hd;
    float sa = 0.f, sb = 0.f;
    for (int i = lane; i < hd; i += 64) {
        const float xv = x[i];
        sa = fmaf(xv, wab[i], sa);
        sb = fmaf(xv, wab[hd + i], sb);
    }
    sa = wave_sum(sa);
    sb = wave_sum(sb);
    if (lane == 0) {
        const float a = sigmoid_f(sa + bab[0]), bb = sigmoid_f(sb + bab[1]);
        const int b = (int)(row / N), i = (int)(row % N);
        gate[((long long)b * H + h) * N + i] = a * (bb * cst[h] - 1.f) + 2.f;
    }
}
int launch_ssl_gate(const float* hidden, const float* wab, const float* bab, const float* cst, float* gate, int B, int N, int H, int hd,
                    hipStream_t s) {
    hipLaunchKernelGGL(ssl_gate_kernel, dim3((unsigned)ceil_div((long long)B * N * H, 4)), dim3(256), 0, s, hidden, wab, bab, cst, gate, B,
                       N, H, hd);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// ---- elementwise tail ---------------------------------------------------------------------------------------------------
// dst = (first ? 0 : dst) + src                        (sum of the selected hidden states)
__global__ void ssl_accumulate_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n4, int first) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 a = reinterpret_cast<const float4*>(src)[i];
    if (!first) {
        const float4 d = reinterpret_cast<const float4*>(dst)[i];
        a.x += d.x; a.y += d.y; a.z += d.z; a.w += d.w;
    }
    reinterpret_cast<float4*>(dst)[i] = a;
}
int launch_ssl_accumulate(float* dst, const float* src, long long n, int first, hipStream_t s) {
    QA_REQUIRE(n % 4 == 0, "ssl accumulate: n %% 4 != 0");
    hipLaunchKernelGGL(ssl_accumulate_kernel, dim3((unsigned)ceil_div(n / 4, 256)), dim3(256), 0, s, dst, src, n / 4, first);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// y = act(x) in place (GELU after the per-layer LayerNorm of the "layer" feature-extractor flavour)
__global__ void ssl_act_kernel(float* __restrict__ x, long long n4, int act) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 a = reinterpret_cast<float4*>(x)[i];
    a.x = apply_act(a.x, act); a.y = apply_act(a.y, act); a.z = apply_act(a.z, act); a.w = apply_act(a.w, act);
    reinterpret_cast<float4*>(x)[i] = a;
}
int launch_ssl_act(float* x, long long n, int act, hipStream_t s) {
    QA_REQUIRE(n % 4 == 0, "ssl act: n %% 4 != 0");
    hipLaunchKernelGGL(ssl_act_kernel, dim3((unsigned)ceil_div(n / 4, 256)), dim3(256), 0, s, x, n / 4, act);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// out = sign * |m|^e with m = sum * scale and sign = +1 for m > 0, -1 otherwise (audio_tokenizer.py:43-46: "(x > 0) * 2 - 1");
// expo <= 0: out = m
__global__ void ssl_compress_kernel(const float* __restrict__ sum, float* __restrict__ out, long long n, float scale, float expo) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float m = sum[i] * scale;
    out[i] = expo > 0.f ? (m > 0.f ? 1.f : -1.f) * powf(fabsf(m), expo) : m;
}
int launch_ssl_compress(const float* sum, float* out, long long n, float scale, float expo, hipStream_t s) {
    hipLaunchKernelGGL(ssl_compress_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, sum, out, n, scale, expo);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

}  // namespace qa
