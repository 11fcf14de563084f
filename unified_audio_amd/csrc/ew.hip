// ew.hip - HBM-bound row / elementwise kernels of the codec graph (channel-last [rows, C] activations):
// first conv (C_in = 1), RMSNorm, LayerNorm, depthwise conv (+ fused LayerNorm), GroupNorm (+ swish),
// RoPE, layout conversions, ISTFT spectrum + overlap-add.  All are one-pass, float4-coalesced, one wave64
// per activation row where a row reduction is needed.
#include "kernels.h"

namespace qa {

// ------------------------------------------------------------------------------------------------
// conv_in: SConv1d with C_in = 1 (reference: encoder.model.0, encoder_modules/conv.py:195-211; seanet.py:121-124)
// y[b, t, co] = bias[co] + sum_j w[co, j] * x[b, reflect(t - pad_left + j)]
// one thread = one output frame x 4 channels -> float4 stores, consecutive threads write consecutive 16 B.
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y, int B,
                                                      int T, int Cout, int ksize, int pad_left, int Lp) {
    const int c4n = Cout >> 2;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * T * c4n;
    if (gid >= total) return;
    const int c4 = (int)(gid % c4n);
    const long long m = gid / c4n;
    const int b = (int)(m / T), t = (int)(m - (long long)b * T);
    const float* xb = x + (long long)b * T;
    float4 acc = bias ? *reinterpret_cast<const float4*>(bias + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < ksize; ++j) {
        const int src = resolve_frame(t - pad_left + j, T, Lp, PAD_REFLECT);
        const float xv = src >= 0 ? xb[src] : 0.f;
        const float* wj = w + (long long)j * Cout + c4 * 4;  // library layout [ksize][Cout]
        acc.x = fmaf(xv, wj[0], acc.x);
        acc.y = fmaf(xv, wj[1], acc.y);
        acc.z = fmaf(xv, wj[2], acc.z);
        acc.w = fmaf(xv, wj[3], acc.w);
    }
    *reinterpret_cast<float4*>(y + m * Cout + c4 * 4) = acc;
}

int launch_conv_in(const float* x, const float* w_kc, const float* bias, float* y, int B, int T, int Cout, int ksize,
                   hipStream_t s) {
    QA_REQUIRE(Cout % 4 == 0, "conv_in: Cout=%d must be a multiple of 4", Cout);
    const int pad_total = ksize - 1, right = pad_total / 2, left = pad_total - right;
    const int max_pad = left > right ? left : right;
    const int Lp = (T <= max_pad) ? max_pad + 1 : T;
    const long long total = (long long)B * T * (Cout / 4);
    hipLaunchKernelGGL(conv_in_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, x, w_kc, bias, y, B, T,
                       Cout, ksize, left, Lp);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// ------------------------------------------------------------------------------------------------
// Row norms: one wave per row, up to 8 float4 per lane (C <= 2048).
constexpr int MAX_V4 = 8;

// mode 0: RMSNorm (transformer.py:77-96, eps inside the sqrt of mean(x^2)), mode 1: LayerNorm (biased variance)
template <int MODE>
__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ b, float* __restrict__ y,
                                                      long long rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * C;
    float4 v[MAX_V4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_V4; ++i) {
        const int c = lane * 4 + i * 256;
        if (c < C) {
            v[i] = *reinterpret_cast<const float4*>(xr + c);
            s += (MODE == 0) ? (v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w)
                             : (v[i].x + v[i].y + v[i].z + v[i].w);
        }
    }
    s = wave_sum(s);
    float mean = 0.f, rstd;
    if (MODE == 0) {
        rstd = rsqrtf(s / C + eps);
    } else {
        mean = s / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAX_V4; ++i) {
            if (lane * 4 + i * 256 < C) {
                const float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
                q += a * a + bb * bb + cc * cc + d * d;
            }
        }
        q = wave_sum(q);
        rstd = rsqrtf(q / C + eps);
    }
    float* yr = y + row * C;
#pragma unroll
    for (int i = 0; i < MAX_V4; ++i) {
        const int c = lane * 4 + i * 256;
        if (c >= C) continue;
        const float4 ww = *reinterpret_cast<const float4*>(w + c);
        float4 o;
        o.x = (v[i].x - mean) * rstd * ww.x;
        o.y = (v[i].y - mean) * rstd * ww.y;
        o.z = (v[i].z - mean) * rstd * ww.z;
        o.w = (v[i].w - mean) * rstd * ww.w;
        if (MODE == 1 && b) {
            const float4 bv = *reinterpret_cast<const float4*>(b + c);
            o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
        }
        *reinterpret_cast<float4*>(yr + c) = o;
    }
}

int launch_rmsnorm(const float* x, const float* w, float* y, long long rows, int C, float eps, hipStream_t s) {
    QA_REQUIRE(C % 4 == 0 && C <= 256 * MAX_V4, "rmsnorm: C=%d unsupported", C);
    hipLaunchKernelGGL(rownorm_kernel<0>, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, s, x, w, nullptr, y, rows,
                       C, eps);
    QA_LAUNCH_CHECK();
    return QA_OK;
}
int launch_layernorm(const float* x, const float* w, const float* b, float* y, long long rows, int C, float eps,
                     hipStream_t s) {
    QA_REQUIRE(C % 4 == 0 && C <= 256 * MAX_V4, "layernorm: C=%d unsupported", C);
    hipLaunchKernelGGL(rownorm_kernel<1>, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, s, x, w, b, y, rows, C,
                       eps);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// ------------------------------------------------------------------------------------------------
// Depthwise Conv1d (zero "same" padding, vq/conv.py:33-56) with optional fused LayerNorm over channels
// (ConvNeXtBlock: dwconv k7 -> LN, vq/conv.py:200-203; sub-pixel upsampler's dw k5, vq/conv.py:86-90).
// w layout [ksize][C] so that lanes read consecutive channels.
template <bool LN>
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, const float* __restrict__ lnw,
                                                     const float* __restrict__ lnb, float* __restrict__ y, int B, int T,
                                                     int C, int ksize, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)B * T) return;
    const int b = (int)(row / T), t = (int)(row - (long long)b * T);
    const int pad = ksize / 2;
    const float* xb = x + (long long)b * T * C;
    float4 v[MAX_V4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_V4; ++i) {
        const int c = lane * 4 + i * 256;
        if (c >= C) continue;
        float4 acc = *reinterpret_cast<const float4*>(bias + c);
        for (int j = 0; j < ksize; ++j) {
            const int src = t + j - pad;
            if (src < 0 || src >= T) continue;
            const float4 xv = *reinterpret_cast<const float4*>(xb + (long long)src * C + c);
            const float4 wv = *reinterpret_cast<const float4*>(w + (long long)j * C + c);
            acc.x = fmaf(xv.x, wv.x, acc.x);
            acc.y = fmaf(xv.y, wv.y, acc.y);
            acc.z = fmaf(xv.z, wv.z, acc.z);
            acc.w = fmaf(xv.w, wv.w, acc.w);
        }
        v[i] = acc;
        s += acc.x + acc.y + acc.z + acc.w;
    }
    float mean = 0.f, rstd = 1.f;
    if (LN) {
        mean = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAX_V4; ++i) {
            if (lane * 4 + i * 256 < C) {
                const float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
                q += a * a + bb * bb + cc * cc + d * d;
            }
        }
        rstd = rsqrtf(wave_sum(q) / C + eps);
    }
    float* yr = y + row * C;
#pragma unroll
    for (int i = 0; i < MAX_V4; ++i) {
        const int c = lane * 4 + i * 256;
        if (c >= C) continue;
        float4 o = v[i];
        if (LN) {
            const float4 ww = *reinterpret_cast<const float4*>(lnw + c);
            const float4 bv = *reinterpret_cast<const float4*>(lnb + c);
            o.x = (o.x - mean) * rstd * ww.x + bv.x;
            o.y = (o.y - mean) * rstd * ww.y + bv.y;
            o.z = (o.z - mean) * rstd * ww.z + bv.z;
            o.w = (o.w - mean) * rstd * ww.w + bv.w;
        }
        *reinterpret_cast<float4*>(yr + c) = o;
    }
}

int launch_dwconv(const float* x, const float* w_kc, const float* bias, const float* lnw, const float* lnb, float* y,
                  int B, int T, int C, int ksize, float eps, hipStream_t s) {
    QA_REQUIRE(C % 4 == 0 && C <= 256 * MAX_V4 && (ksize & 1), "dwconv: C=%d ksize=%d unsupported", C, ksize);
    const unsigned grid = (unsigned)ceil_div((long long)B * T, 4);
    if (lnw)
        hipLaunchKernelGGL(dwconv_kernel<true>, dim3(grid), dim3(256), 0, s, x, w_kc, bias, lnw, lnb, y, B, T, C, ksize,
                           eps);
    else
        hipLaunchKernelGGL(dwconv_kernel<false>, dim3(grid), dim3(256), 0, s, x, w_kc, bias, lnw, lnb, y, B, T, C,
                           ksize, eps);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// ------------------------------------------------------------------------------------------------
// GroupNorm(G, C, eps) over [B, T, C] channel-last (vq/conv.py:259-260 "Normalize"), optional swish after it
// (ResnetBlock.nonlinearity, vq/conv.py:303-304).  Two deterministic passes:
//   1. gn_partial: grid (chunks, B); each block reduces `rows_per_chunk` frames: thread = one float4 channel
//      column, fp32 running sum / sum of squares, then a group reduction in LDS -> partial[b][chunk][g][2] (double)
//   2. gn_apply: every block first folds the partials of its batch item in a fixed order (LDS), then streams.
constexpr int GN_ROWS = 64;

__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, double* __restrict__ partial,
                                                         int T, int C, int G) {
    extern __shared__ float sh[];  // [2][C]
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int t0 = chunk * GN_ROWS, t1 = min(T, t0 + GN_ROWS);
    const float* xb = x + (long long)b * T * C;
    for (int c = threadIdx.x * 4; c < C; c += blockDim.x * 4) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
        for (int t = t0; t < t1; ++t) {
            const float4 v = *reinterpret_cast<const float4*>(xb + (long long)t * C + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            q.x = fmaf(v.x, v.x, q.x); q.y = fmaf(v.y, v.y, q.y); q.z = fmaf(v.z, v.z, q.z); q.w = fmaf(v.w, v.w, q.w);
        }
        sh[c] = s.x; sh[c + 1] = s.y; sh[c + 2] = s.z; sh[c + 3] = s.w;
        sh[C + c] = q.x; sh[C + c + 1] = q.y; sh[C + c + 2] = q.z; sh[C + c + 3] = q.w;
    }
    __syncthreads();
    const int cpg = C / G;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        double s = 0.0, q = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            s += (double)sh[c];
            q += (double)sh[C + c];
        }
        double* out = partial + (((long long)b * nchunk + chunk) * G + g) * 2;
        out[0] = s;
        out[1] = q;
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const double* __restrict__ partial,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ y, int T, int C, int G, int nchunk, float eps,
                                                       int swish) {
    extern __shared__ float sh[];  // [2][G]: mean, rstd
    const int b = blockIdx.y;
    const int cpg = C / G;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        double s = 0.0, q = 0.0;
        for (int k = 0; k < nchunk; ++k) {
            const double* pp = partial + (((long long)b * nchunk + k) * G + g) * 2;
            s += pp[0];
            q += pp[1];
        }
        const double n = (double)T * cpg;
        const double mean = s / n;
        double var = q / n - mean * mean;
        if (var < 0.0) var = 0.0;
        sh[g] = (float)mean;
        sh[G + g] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int c4n = C >> 2;
    const long long n4 = (long long)T * c4n;
    const float* xb = x + (long long)b * T * C;
    float* yb = y + (long long)b * T * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const float4 v = *reinterpret_cast<const float4*>(xb + i * 4);
        const float4 ww = *reinterpret_cast<const float4*>(w + c);
        const float4 bv = *reinterpret_cast<const float4*>(bias + c);
        float4 o;
        // channels c..c+3 may straddle groups only if cpg % 4 != 0 -> resolve per element
        o.x = (v.x - sh[(c + 0) / cpg]) * sh[G + (c + 0) / cpg] * ww.x + bv.x;
        o.y = (v.y - sh[(c + 1) / cpg]) * sh[G + (c + 1) / cpg] * ww.y + bv.y;
        o.z = (v.z - sh[(c + 2) / cpg]) * sh[G + (c + 2) / cpg] * ww.z + bv.z;
        o.w = (v.w - sh[(c + 3) / cpg]) * sh[G + (c + 3) / cpg] * ww.w + bv.w;
        if (swish) {
            o.x = silu_f(o.x); o.y = silu_f(o.y); o.z = silu_f(o.z); o.w = silu_f(o.w);
        }
        *reinterpret_cast<float4*>(yb + i * 4) = o;
    }
}

size_t groupnorm_scratch_bytes(int B, int T, int G) { return (size_t)B * ceil_div(T, GN_ROWS) * G * 2 * sizeof(double); }

int launch_groupnorm(const float* x, const float* w, const float* bias, float* y, double* scratch, int B, int T, int C,
                     int G, float eps, int swish, hipStream_t s) {
    QA_REQUIRE(C % 4 == 0 && C % G == 0, "groupnorm: C=%d G=%d unsupported", C, G);
    const int nchunk = (int)ceil_div(T, GN_ROWS);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunk, B), dim3(256), 2 * C * sizeof(float), s, x, scratch, T, C, G);
    QA_LAUNCH_CHECK();
    const long long n4 = (long long)T * (C / 4);
    const unsigned gx = (unsigned)std::min<long long>(ceil_div(n4, 256), 64);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(gx, B), dim3(256), 2 * G * sizeof(float), s, x, scratch, w, bias, y, T, C,
                       G, nchunk, eps, swish);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// ------------------------------------------------------------------------------------------------
// Rotate-half RoPE applied in place to the q and k parts of a fused [rows, 3*d] QKV buffer
// (transformer.py:182-215: q*cos + rotate_half(q)*sin, positions 0..N-1, tables as RotaryEmbedding.forward).
__global__ __launch_bounds__(256) void rope_kernel(float* __restrict__ qkv, const float* __restrict__ cs, int B, int N,
                                                   int H, int hd, long long ld, int pos0) {
    // one thread: one (row, head, pair i < hd/2) for q and for k
    const int half = hd >> 1;
    const long long total = (long long)B * N * H * half;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int i = (int)(gid % half);
    const int h = (int)((gid / half) % H);
    const long long row = gid / ((long long)half * H);
    const int t = (int)(row % N) + pos0;
    const float c = cs[((long long)t * half + i) * 2], s = cs[((long long)t * half + i) * 2 + 1];
    const int d = H * hd;
    float* q = qkv + row * ld + h * hd;
    float* k = q + d;
    const float q1 = q[i], q2 = q[i + half];
    q[i] = q1 * c - q2 * s;
    q[i + half] = q2 * c + q1 * s;
    const float k1 = k[i], k2 = k[i + half];
    k[i] = k1 * c - k2 * s;
    k[i + half] = k2 * c + k1 * s;
}

int launch_rope(float* qkv, const float* cos_sin, int B, int N, int H, int hd, long long ld, int pos0, hipStream_t s) {
    const long long total = (long long)B * N * H * (hd / 2);
    hipLaunchKernelGGL(rope_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, qkv, cos_sin, B, N, H, hd, ld,
                       pos0);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// ------------------------------------------------------------------------------------------------
// Strided [B, C, T] (element strides sb, sc, st) -> contiguous channel-last [B, T, C] through a 32x32 LDS tile.
__global__ __launch_bounds__(256) void to_channel_last_kernel(const float* __restrict__ x, long long sb, long long sc,
                                                              long long st, float* __restrict__ y, int C, int T) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const bool t_fast = st <= sc;  // read along the contiguous axis of the source
    for (int k = ty; k < 32; k += 8) {
        const int c = t_fast ? c0 + k : c0 + tx;
        const int t = t_fast ? t0 + tx : t0 + k;
        if (c < C && t < T) tile[t - t0][c - c0] = x[b * sb + c * sc + t * st];
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int t = t0 + k, c = c0 + tx;
        if (c < C && t < T) y[((long long)b * T + t) * C + c] = tile[k][tx];
    }
}

int launch_to_channel_last(const float* x, long long sb, long long sc, long long st, float* y, int B, int C, int T,
                           hipStream_t s) {
    hipLaunchKernelGGL(to_channel_last_kernel, dim3((unsigned)ceil_div(T, 32), (unsigned)ceil_div(C, 32), B), dim3(256),
                       0, s, x, sb, sc, st, y, C, T);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// codes: library [n_vec = B*N, Q] -> reference [B, Q, N] (codec.py:173-174) and back (codec.py:179-180)
__global__ void codes_to_bqn_kernel(const long long* __restrict__ src, long long* __restrict__ dst, int B, int N, int Q) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)B * N * Q) return;
    const int n = (int)(gid % N);
    const int q = (int)((gid / N) % Q);
    const int b = (int)(gid / ((long long)N * Q));
    dst[gid] = src[((long long)b * N + n) * Q + q];
}
__global__ void codes_from_bqn_kernel(const long long* __restrict__ src, long long* __restrict__ dst, int B, int N,
                                      int Q) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)B * N * Q) return;
    const int q = (int)(gid % Q);
    const int n = (int)((gid / Q) % N);
    const int b = (int)(gid / ((long long)N * Q));
    dst[gid] = src[((long long)b * Q + q) * N + n];
}
int launch_codes_to_bqn(const long long* src, long long* dst, int B, int N, int Q, hipStream_t s) {
    const long long total = (long long)B * N * Q;
    if (total == 0) return QA_OK;
    hipLaunchKernelGGL(codes_to_bqn_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, src, dst, B, N, Q);
    QA_LAUNCH_CHECK();
    return QA_OK;
}
int launch_codes_from_bqn(const long long* src, long long* dst, int B, int N, int Q, hipStream_t s) {
    const long long total = (long long)B * N * Q;
    if (total == 0) return QA_OK;
    hipLaunchKernelGGL(codes_from_bqn_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, src, dst, B, N, Q);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// ------------------------------------------------------------------------------------------------
// ISTFT head, step 1 (vq/heads.py:137-146): head output [rows, 2*nb] = (log-mag | phase) ->
// spectrum rows [rows, ldS] = (mag*cos p | mag*sin p | 0 pad), mag = min(exp(m), 100).
__global__ __launch_bounds__(256) void istft_spec_kernel(const float* __restrict__ y, float* __restrict__ S,
                                                         long long rows, int nb, int ldy, int ldS) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = rows * (long long)(ldS - nb);  // one thread per (row, column in [0, ldS - nb))
    if (gid >= total) return;
    const int w = ldS - nb;
    const int k = (int)(gid % w);
    const long long r = gid / w;
    float re = 0.f, im = 0.f;
    if (k < nb) {
        const float m = fminf(expf(y[r * ldy + k]), 100.f);
        float sn, cs;
        sincosf(y[r * ldy + nb + k], &sn, &cs);
        re = m * cs;
        im = m * sn;
        S[r * ldS + k] = re;
    }
    S[r * ldS + nb + k] = im;  // columns [nb, ldS): imaginary part, then zeros
}

int launch_istft_spec(const float* y, float* S, long long rows, int nb, int ldy, int ldS, hipStream_t s) {
    QA_REQUIRE(ldS >= 2 * nb, "istft_spec: ldS=%d < 2*nb=%d", ldS, 2 * nb);
    const long long total = rows * (long long)(ldS - nb);
    hipLaunchKernelGGL(istft_spec_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, y, S, rows, nb, ldy,
                       ldS);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// ISTFT step 3 (vq/spectral_ops.py:58-73): overlap-add of the windowed frames, trim (win-hop)/2 each side,
// divide by the folded hann^2 envelope.  frames [B, T, n_fft] (already multiplied by the window inside the
// inverse-DFT basis), out [B, T*hop].
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, const float* __restrict__ win,
                                                        float* __restrict__ out, int B, int T, int n_fft, int hop) {
    const long long L = (long long)T * hop;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)B * L) return;
    const int b = (int)(gid / L);
    const long long n = gid - (long long)b * L;
    const long long p = n + (n_fft - hop) / 2;
    int t_hi = (int)(p / hop);
    if (t_hi > T - 1) t_hi = T - 1;
    long long lo = p - n_fft + 1;
    int t_lo = lo <= 0 ? 0 : (int)((lo + hop - 1) / hop);
    float acc = 0.f, env = 0.f;
    for (int t = t_lo; t <= t_hi; ++t) {
        const int k = (int)(p - (long long)t * hop);
        acc += frames[((long long)b * T + t) * n_fft + k];
        const float w = win[k];
        env = fmaf(w, w, env);
    }
    out[gid] = acc / env;
}

int launch_istft_ola(const float* frames, const float* win, float* out, int B, int T, int n_fft, int hop, hipStream_t s) {
    const long long total = (long long)B * T * hop;
    if (total == 0) return QA_OK;
    hipLaunchKernelGGL(istft_ola_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, frames, win, out, B, T,
                       n_fft, hop);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

}  // namespace qa
