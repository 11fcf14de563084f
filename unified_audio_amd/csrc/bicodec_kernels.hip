// bicodec_kernels.hip - the byte-bound pieces of BiCodec.detokenize (QuarkAudio-UniSE/model/bicodec/bicodec.py:182-199) that are
// not contractions: token look-ups into folded tables, AdaLayerNorm, the d-vector broadcast add.  Everything with a contraction
// (linears, k7 / dilated k7 convolutions, the polyphase ConvTranspose1d) runs on conv_gemm.hip.
#include "kernels.h"

namespace qa {

// out[i, :] = table[clamp(tok[i]), :]   (FactorizedVectorQuantize.detokenize with out_project folded into the table)
__global__ __launch_bounds__(256) void gather_rows_kernel(const long long* __restrict__ tok, const float* __restrict__ table, float* __restrict__ out,
                                                          long long n, int V, int D) {
    const int d4 = D >> 2;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n * d4) return;
    const long long i = gid / d4;
    const int c = (int)(gid - i * d4) * 4;
    long long t = tok[i];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);  // memory safety only; callers validate (qa_codes_check)
    *reinterpret_cast<float4*>(out + i * D + c) = *reinterpret_cast<const float4*>(table + t * D + c);
}
int launch_gather_rows(const long long* tok, const float* table, float* out, long long n, int V, int D, hipStream_t s) {
    QA_REQUIRE(D % 4 == 0, "gather_rows: D=%d must be a multiple of 4", D);
    if (n <= 0) return QA_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)ceil_div(n * (D / 4), 256)), dim3(256), 0, s, tok, table, out, n, V, D);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// SpeakerEncoder.detokenize up to the flatten (speaker_encoder.py:111-114): zq [B, n, L] = table[tok[b, n]] (FSQ codes x project_out
// folded), then zq.transpose(1, 2).reshape(B, -1): out[b, c * N + n] = table[tok[b, n]][c]
__global__ __launch_bounds__(256) void gather_global_kernel(const long long* __restrict__ tok, const float* __restrict__ table,
                                                            float* __restrict__ out, int B, int N, int V, int L) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= B * N * L) return;
    const int n = gid % N, c = (gid / N) % L, b = gid / (N * L);
    long long t = tok[b * N + n];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    out[gid] = table[t * L + c];
}
int launch_gather_global(const long long* tok, const float* table, float* out, int B, int N, int V, int L, hipStream_t s) {
    hipLaunchKernelGGL(gather_global_kernel, dim3((unsigned)ceil_div((long long)B * N * L, 256)), dim3(256), 0, s, tok, table, out, B, N, V, L);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// AdaLayerNorm (blocks/vocos.py:113-136): y = LayerNorm(x, no affine, eps) * scale[b, :] + shift[b, :]; one wave per row of C channels
__global__ __launch_bounds__(256) void adaln_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                    long long ld_cond, float* __restrict__ y, long long rows, int T, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * C;
    float s = 0.f, ss = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 t = *reinterpret_cast<const float4*>(xr + c);
        s += (t.x + t.y) + (t.z + t.w);
    }
    s = wave_sum(s);
    const float mean = s / C;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 t = *reinterpret_cast<const float4*>(xr + c);
        const float a = t.x - mean, b = t.y - mean, cc = t.z - mean, d = t.w - mean;
        ss += (a * a + b * b) + (cc * cc + d * d);
    }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / C + eps);
    const long long b = row / T;
    const float* sc = scale + b * ld_cond;
    const float* sh = shift + b * ld_cond;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 t = *reinterpret_cast<const float4*>(xr + c);
        const float4 g = *reinterpret_cast<const float4*>(sc + c);
        const float4 h = *reinterpret_cast<const float4*>(sh + c);
        float4 o;
        o.x = (t.x - mean) * rstd * g.x + h.x;
        o.y = (t.y - mean) * rstd * g.y + h.y;
        o.z = (t.z - mean) * rstd * g.z + h.z;
        o.w = (t.w - mean) * rstd * g.w + h.w;
        *reinterpret_cast<float4*>(y + row * C + c) = o;
    }
}
int launch_adaln(const float* x, const float* scale, const float* shift, long long ld_cond, float* y, int B, int T, int C, float eps,
                 hipStream_t s) {
    QA_REQUIRE(C % 4 == 0 && ld_cond % 4 == 0, "adaln: C=%d / ld=%lld must be multiples of 4", C, ld_cond);
    const long long rows = (long long)B * T;
    hipLaunchKernelGGL(adaln_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, s, x, scale, shift, ld_cond, y, rows, T, C, eps);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// x[b, t, :] += v[b, :]   (bicodec.py:196: x = x + d_vector.unsqueeze(-1))
__global__ __launch_bounds__(256) void add_rowvec_kernel(float* __restrict__ x, const float* __restrict__ v, long long n4, int T, int C) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n4) return;
    const int c4 = C >> 2;
    const long long row = gid / c4;
    const int c = (int)(gid - row * c4) * 4;
    float4 t = *reinterpret_cast<float4*>(x + row * C + c);
    const float4 a = *reinterpret_cast<const float4*>(v + (row / T) * C + c);
    t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
    *reinterpret_cast<float4*>(x + row * C + c) = t;
}
int launch_add_rowvec(float* x, const float* v, int B, int T, int C, hipStream_t s) {
    QA_REQUIRE(C % 4 == 0, "add_rowvec: C=%d must be a multiple of 4", C);
    const long long n4 = (long long)B * T * (C / 4);
    hipLaunchKernelGGL(add_rowvec_kernel, dim3((unsigned)ceil_div(n4, 256)), dim3(256), 0, s, x, v, n4, T, C);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

}  // namespace qa
