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
                   hipStream_t s, int pad_left) {
    QA_REQUIRE(Cout % 4 == 0 && pad_left < ksize, "conv_in: Cout=%d must be a multiple of 4 (pad_left %d, ksize %d)", Cout, pad_left, ksize);
    const int pad_total = ksize - 1;
    const int left = pad_left >= 0 ? pad_left : pad_total - pad_total / 2, right = pad_total - left;
    const int max_pad = left > right ? left : right;
    const int Lp = (T <= max_pad) ? max_pad + 1 : T;
    const long long total = (long long)B * T * (Cout / 4);
    HbmProf prof_(HK_CONV_IN, 4.0 * ((double)B * T + (double)B * T * Cout), s);
    hipLaunchKernelGGL(conv_in_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, x, w_kc, bias, y, B, T,
                       Cout, ksize, left, Lp);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// ------------------------------------------------------------------------------------------------
// Row norms: one wave per row, up to 8 float4 per lane (C <= 2048).
constexpr int MAX_V4 = 8;

// mode 0: RMSNorm (transformer.py:77-96, eps inside the sqrt of mean(x^2)), mode 1: LayerNorm (biased variance).  Every arithmetic step is one of
// common.h's norm_* helpers: conv_gemm_kernel's norm-fused A-operand staging (r06) repeats them and must produce the same bits.
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
            s = norm_add(s, (MODE == 0) ? norm_sq4(v[i].x, v[i].y, v[i].z, v[i].w) : norm_sum4(v[i].x, v[i].y, v[i].z, v[i].w));
        }
    }
    s = wave_sum(s);
    float mean = 0.f, rstd;
    if (MODE == 0) {
        rstd = norm_rstd(s, C, eps);
    } else {
        mean = norm_mean(s, C);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAX_V4; ++i)
            if (lane * 4 + i * 256 < C) q = norm_add(q, norm_csq4(v[i].x, v[i].y, v[i].z, v[i].w, mean));
        q = wave_sum(q);
        rstd = norm_rstd(q, C, eps);
    }
    float* yr = y + row * C;
#pragma unroll
    for (int i = 0; i < MAX_V4; ++i) {
        const int c = lane * 4 + i * 256;
        if (c >= C) continue;
        const float4 ww = *reinterpret_cast<const float4*>(w + c);
        const float4 bv = (MODE == 1 && b) ? *reinterpret_cast<const float4*>(b + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 o;
        o.x = norm_apply(v[i].x, mean, rstd, ww.x, bv.x);
        o.y = norm_apply(v[i].y, mean, rstd, ww.y, bv.y);
        o.z = norm_apply(v[i].z, mean, rstd, ww.z, bv.z);
        o.w = norm_apply(v[i].w, mean, rstd, ww.w, bv.w);
        *reinterpret_cast<float4*>(yr + c) = o;
    }
}

int launch_rmsnorm(const float* x, const float* w, float* y, long long rows, int C, float eps, hipStream_t s) {
    QA_REQUIRE(C % 4 == 0 && C <= 256 * MAX_V4, "rmsnorm: C=%d unsupported", C);
    HbmProf prof_(HK_ROWNORM, 8.0 * (double)rows * C, s);
    hipLaunchKernelGGL(rownorm_kernel<0>, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, s, x, w, nullptr, y, rows,
                       C, eps);
    QA_LAUNCH_CHECK();
    return QA_OK;
}
int launch_layernorm(const float* x, const float* w, const float* b, float* y, long long rows, int C, float eps,
                     hipStream_t s) {
    QA_REQUIRE(C % 4 == 0 && C <= 256 * MAX_V4, "layernorm: C=%d unsupported", C);
    HbmProf prof_(HK_ROWNORM, 8.0 * (double)rows * C, s);
    hipLaunchKernelGGL(rownorm_kernel<1>, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, s, x, w, b, y, rows, C,
                       eps);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// ------------------------------------------------------------------------------------------------
// Depthwise Conv1d (zero "same" padding, vq/conv.py:33-56) with optional fused LayerNorm over channels
// (ConvNeXtBlock: dwconv k7 -> LN, vq/conv.py:200-203; sub-pixel upsampler's dw k5, vq/conv.py:86-90).
// w layout [ksize][C] so that lanes read consecutive channels.
// KS > 0: the tap count is a compile-time constant (7: ConvNeXt, 5: the sub-pixel upsampler) - every tap of a channel chunk is loaded before the first
// FMA (clamped source frame, taps outside the clip skipped) instead of load -> wait -> fma per tap; measured equal (42.9 -> 44.9 us at 32 x 500 x 1024):
// this kernel is bound by every input row passing through a CU once per tap, which dwconv_strip_kernel below removes.  It serves the shapes the strip
// kernel does not (C not in whole 256-channel chunks, other kernel sizes) and is its bit-exact reference.  KS = 0: any other kernel size.
template <bool LN, int KS>
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, const float* __restrict__ lnw,
                                                     const float* __restrict__ lnb, float* __restrict__ y, int B, int T,
                                                     int C, int ksize, float eps, int pad) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)B * T) return;
    const int b = (int)(row / T), t = (int)(row - (long long)b * T);
    const float* xb = x + (long long)b * T * C;
    float4 v[MAX_V4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_V4; ++i) {
        const int c = lane * 4 + i * 256;
        if (c >= C) continue;
        float4 acc = *reinterpret_cast<const float4*>(bias + c);
        if (KS > 0) {
            float4 xv[KS > 0 ? KS : 1], wv[KS > 0 ? KS : 1];
            float keep[KS > 0 ? KS : 1];
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                const int src = t + j - pad;
                keep[j] = (src >= 0 && src < T) ? 1.f : 0.f;
                xv[j] = *reinterpret_cast<const float4*>(xb + (long long)min(max(src, 0), T - 1) * C + c);
                wv[j] = *reinterpret_cast<const float4*>(w + (long long)j * C + c);
            }
#pragma unroll
            for (int j = 0; j < KS; ++j) {  // same fma chain, in tap order, as the run-time loop (a tap outside the clip adds +0 * w: exact)
                if (keep[j] == 0.f) continue;
                acc.x = fmaf(xv[j].x, wv[j].x, acc.x);
                acc.y = fmaf(xv[j].y, wv[j].y, acc.y);
                acc.z = fmaf(xv[j].z, wv[j].z, acc.z);
                acc.w = fmaf(xv[j].w, wv[j].w, acc.w);
            }
        } else {
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
        }
        v[i] = acc;
        s = norm_add(s, norm_sum4(acc.x, acc.y, acc.z, acc.w));
    }
    float mean = 0.f, rstd = 1.f;
    if (LN) {  // the statements of common.h's norm_* helpers: dwconv_strip_kernel repeats them and produces the same bits
        mean = norm_mean(wave_sum(s), C);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAX_V4; ++i)
            if (lane * 4 + i * 256 < C) q = norm_add(q, norm_csq4(v[i].x, v[i].y, v[i].z, v[i].w, mean));
        rstd = norm_rstd(wave_sum(q), C, eps);
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
            o.x = norm_apply(o.x, mean, rstd, ww.x, bv.x);
            o.y = norm_apply(o.y, mean, rstd, ww.y, bv.y);
            o.z = norm_apply(o.z, mean, rstd, ww.z, bv.z);
            o.w = norm_apply(o.w, mean, rstd, ww.w, bv.w);
        }
        *reinterpret_cast<float4*>(yr + c) = o;
    }
}

// The same operator as a ROW STRIP (r06): a workgroup of NW waves (wave i = channels 256 i .. 256 i + 255, C = 256 NW) walks RS consecutive output
// rows of one clip in passes of 8, keeping the last KS - 1 input rows of a pass in registers for the next one - every input row enters a CU once per strip
// (+ (KS - 1) / RS halo) instead of once per tap.  dwconv_kernel is bound by exactly that: a row norm of the same bytes runs at 6.0 TB/s on this device,
// dwconv_kernel at 3.0 (tools/copy_floor.py, profiles/r06_dwconv_strip_ab.txt).  Per output element the fma chain (bias, then the taps in order, taps
// outside the clip skipped) is the one of dwconv_kernel; the LayerNorm sums are formed in its order as well - per lane over the chunks i = 0 .. NW - 1
// (through LDS: the chunks live in different waves here), then the wave butterfly - so the two kernels agree BIT FOR BIT
// (tests/test_kernels_gpu.py::test_dwconv_strip_is_bit_identical_to_the_row_kernel).
constexpr int DW_RB = 8;  // output rows per pass
template <bool LN, int KS, int NW>
__global__ __launch_bounds__(64 * NW) void dwconv_strip_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                               const float* __restrict__ lnw, const float* __restrict__ lnb, float* __restrict__ y,
                                                               int T, int C, float eps, int pad, int RS, int strips) {
    __shared__ float s_sum[LN ? DW_RB : 1][NW][64];
    __shared__ float s_sq[LN ? DW_RB : 1][NW][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = wv * 256 + lane * 4;
    const int b = blockIdx.x / strips, t_begin = (blockIdx.x - b * strips) * RS, t_end = min(T, t_begin + RS);
    const float* xb = x + (long long)b * T * C + c;
    float* yb = y + (long long)b * T * C + c;
    float4 wr[KS];
#pragma unroll
    for (int j = 0; j < KS; ++j) wr[j] = *reinterpret_cast<const float4*>(w + (long long)j * C + c);
    const float4 bs = *reinterpret_cast<const float4*>(bias + c);
    float4 lw = make_float4(1.f, 1.f, 1.f, 1.f), lb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (LN) {
        lw = *reinterpret_cast<const float4*>(lnw + c);
        lb = *reinterpret_cast<const float4*>(lnb + c);
    }
    // win[k] = input row (t0 - pad + k) of the current pass; rows outside the clip are loaded from a clamped row and never used
    float4 win[DW_RB + KS - 1];
#pragma unroll
    for (int k = 0; k < KS - 1; ++k) win[DW_RB + k] = *reinterpret_cast<const float4*>(xb + (long long)min(max(t_begin - pad + k, 0), T - 1) * C);
    for (int t0 = t_begin; t0 < t_end; t0 += DW_RB) {
#pragma unroll
        for (int k = 0; k < KS - 1; ++k) win[k] = win[DW_RB + k];  // the halo of the previous pass
#pragma unroll
        for (int k = KS - 1; k < DW_RB + KS - 1; ++k) win[k] = *reinterpret_cast<const float4*>(xb + (long long)min(max(t0 - pad + k, 0), T - 1) * C);
        float4 acc[DW_RB];
#pragma unroll
        for (int r = 0; r < DW_RB; ++r) {
            acc[r] = bs;
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                const int src = t0 + r + j - pad;  // wave-uniform
                if (src < 0 || src >= T) continue;
                acc[r].x = fmaf(win[r + j].x, wr[j].x, acc[r].x);
                acc[r].y = fmaf(win[r + j].y, wr[j].y, acc[r].y);
                acc[r].z = fmaf(win[r + j].z, wr[j].z, acc[r].z);
                acc[r].w = fmaf(win[r + j].w, wr[j].w, acc[r].w);
            }
        }
        float mean[DW_RB], rstd[DW_RB];
        if (LN) {
#pragma unroll
            for (int r = 0; r < DW_RB; ++r) s_sum[r][wv][lane] = norm_sum4(acc[r].x, acc[r].y, acc[r].z, acc[r].w);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < DW_RB; ++r) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < NW; ++i) s = norm_add(s, s_sum[r][i][lane]);
                mean[r] = norm_mean(wave_sum(s), C);
                s_sq[r][wv][lane] = norm_csq4(acc[r].x, acc[r].y, acc[r].z, acc[r].w, mean[r]);
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < DW_RB; ++r) {
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < NW; ++i) q = norm_add(q, s_sq[r][i][lane]);
                rstd[r] = norm_rstd(wave_sum(q), C, eps);
            }
        }
#pragma unroll
        for (int r = 0; r < DW_RB; ++r) {
            if (t0 + r >= t_end) break;
            float4 o = acc[r];
            if (LN) {
                o.x = norm_apply(o.x, mean[r], rstd[r], lw.x, lb.x);
                o.y = norm_apply(o.y, mean[r], rstd[r], lw.y, lb.y);
                o.z = norm_apply(o.z, mean[r], rstd[r], lw.z, lb.z);
                o.w = norm_apply(o.w, mean[r], rstd[r], lw.w, lb.w);
            }
            *reinterpret_cast<float4*>(yb + (long long)(t0 + r) * C) = o;
        }
    }
}

int launch_dwconv(const float* x, const float* w_kc, const float* bias, const float* lnw, const float* lnb, float* y,
                  int B, int T, int C, int ksize, float eps, hipStream_t s, int pad_left) {
    QA_REQUIRE(C % 4 == 0 && C <= 256 * MAX_V4 && (ksize & 1) && pad_left < ksize, "dwconv: C=%d ksize=%d pad_left=%d unsupported", C,
               ksize, pad_left);
    const unsigned grid = (unsigned)ceil_div((long long)B * T, 4);
    const int pad = pad_left >= 0 ? pad_left : ksize / 2;
    HbmProf prof_(HK_DWCONV_LN, 8.0 * (double)B * T * C, s);
    // the strip kernel (QA_DWCONV_STRIP, default on): C = 256 .. 1024 in whole 256-channel chunks, k = 5 / 7
    if (knob(K_DWCONV_STRIP) != 0 && C % 256 == 0 && C <= 1024 && (ksize == 7 || ksize == 5) && (!lnw || lnb)) {
        const long long rows = (long long)B * T;
        // strip length: ~1000 workgroups per launch (4 per CU: the halo of a longer strip costs less than the parallelism it takes away - sweep of
        // RS = 8 .. 128 at 4 000 / 16 000 / 24 000 rows, profiles/r06_dwconv_strip_ab.txt: best at 8 / 16 / 24)
        int RS = (int)(rows / 1000) / DW_RB * DW_RB;
        RS = RS < DW_RB ? DW_RB : (RS > 64 ? 64 : RS);
        const int strips = (int)ceil_div(T, RS), NW = C / 256;
#define QA_DWS(LN, KS, NW) hipLaunchKernelGGL((dwconv_strip_kernel<LN, KS, NW>), dim3((unsigned)(B * strips)), dim3(64 * NW), 0, s, x, w_kc, bias, lnw, lnb, y, T, C, eps, pad, RS, strips)
#define QA_DWS_NW(LN, KS) do { if (NW == 4) QA_DWS(LN, KS, 4); else if (NW == 3) QA_DWS(LN, KS, 3); else if (NW == 2) QA_DWS(LN, KS, 2); else QA_DWS(LN, KS, 1); } while (0)
        if (lnw) { if (ksize == 7) QA_DWS_NW(true, 7); else QA_DWS_NW(true, 5); }
        else { if (ksize == 7) QA_DWS_NW(false, 7); else QA_DWS_NW(false, 5); }
#undef QA_DWS_NW
#undef QA_DWS
        QA_LAUNCH_CHECK();
        return QA_OK;
    }
#define QA_DW(LN, KS) hipLaunchKernelGGL((dwconv_kernel<LN, KS>), dim3(grid), dim3(256), 0, s, x, w_kc, bias, lnw, lnb, y, B, T, C, ksize, eps, pad)
    if (lnw) {
        if (ksize == 7) QA_DW(true, 7);
        else if (ksize == 5) QA_DW(true, 5);
        else QA_DW(true, 0);
    } else {
        if (ksize == 7) QA_DW(false, 7);
        else if (ksize == 5) QA_DW(false, 5);
        else QA_DW(false, 0);
    }
#undef QA_DW
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
    HbmProf prof_(HK_GROUPNORM, 8.0 * (double)B * T * C, s);  // algorithmic: x once in, y once out (the two-pass form reads x twice)
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
                                                   int H, int hd, long long ld, int pos0, int interleaved) {
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
    // rotate-half pairs (i, i + hd/2) (HF Llama / codec transformers) or interleaved pairs (2i, 2i+1) (mimi, module/rope.py:13-69)
    const int i0 = interleaved ? 2 * i : i, i1 = interleaved ? 2 * i + 1 : i + half;
    const float q1 = q[i0], q2 = q[i1];
    q[i0] = q1 * c - q2 * s;
    q[i1] = q2 * c + q1 * s;
    const float k1 = k[i0], k2 = k[i1];
    k[i0] = k1 * c - k2 * s;
    k[i1] = k2 * c + k1 * s;
}

int launch_rope(float* qkv, const float* cos_sin, int B, int N, int H, int hd, long long ld, int pos0, hipStream_t s,
                int interleaved) {
    const long long total = (long long)B * N * H * (hd / 2);
    HbmProf prof_(HK_ROPE, 16.0 * (double)B * N * H * hd, s);  // q and k parts: read + write
    hipLaunchKernelGGL(rope_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, qkv, cos_sin, B, N, H, hd, ld,
                       pos0, interleaved);
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
    HbmProf prof_(HK_ISTFT_SPEC, 4.0 * (double)rows * (2.0 * nb + ldS), s);
    hipLaunchKernelGGL(istft_spec_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, y, S, rows, nb, ldy,
                       ldS);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// H-Codec 2.0 encoder front (HCodec-2.0/vq/codec_encoder.py:66-71): STFT rows (re | im) [rows, 2*nb] ->
// (log(max(|X|, 1e-5)) | angle(X) / pi | 0 pad) [rows, ldo]
__global__ __launch_bounds__(256) void stft_post_kernel(const float* __restrict__ ri, float* __restrict__ out, long long rows,
                                                        int nb, int ldi, int ldo) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int w = ldo - nb;
    if (gid >= rows * (long long)w) return;
    const int k = (int)(gid % w);
    const long long r = gid / w;
    float ph = 0.f;
    if (k < nb) {
        const float re = ri[r * ldi + k], im = ri[r * ldi + nb + k];
        out[r * ldo + k] = logf(fmaxf(hypotf(re, im), 1e-5f));
        ph = atan2f(im, re) / 3.14159265358979323846f;
    }
    out[r * ldo + nb + k] = ph;
}
int launch_stft_post(const float* ri, float* out, long long rows, int nb, int ldi, int ldo, hipStream_t s) {
    QA_REQUIRE(ldo >= 2 * nb, "stft_post: ldo=%d < 2*nb", ldo);
    const long long total = rows * (long long)(ldo - nb);
    HbmProf prof_(HK_STFT_POST, 4.0 * (double)rows * (2.0 * nb + ldo), s);
    hipLaunchKernelGGL(stft_post_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, ri, out, rows, nb, ldi, ldo);
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
    HbmProf prof_(HK_ISTFT_OLA, 4.0 * ((double)B * T * n_fft + (double)B * T * hop), s);
    hipLaunchKernelGGL(istft_ola_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, frames, win, out, B, T,
                       n_fft, hop);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// ------------------------------------------------------------------------------------------------
// H-Codec 1.5 adaptive frame rate (HCodec-1.5/adaptive/modeling_flexicodec_new.py:828-921): cosine similarity of
// consecutive semantic frames, a new group wherever sim <= threshold, groups capped at max_tokens frames.
// One workgroup per batch item; the scan over T (<= a few hundred frames) is sequential in thread 0.
//   seg[b][t]   group index of frame t          start[b][g], len[b][g] (0 for g >= nseg[b])      nseg[b]
//   gmax        max over b of nseg (atomicMax; zeroed by the launcher)
__global__ __launch_bounds__(256) void align_kernel(const float* __restrict__ sem, int T, int D, float thr, int max_tokens,
                                                    int* __restrict__ seg, int* __restrict__ start, int* __restrict__ len,
                                                    int* __restrict__ nseg, int* __restrict__ gmax) {
    extern __shared__ float sh[];  // [T] 1/max(norm, eps), then [T] sim
    float* inv = sh;
    float* sim = sh + T;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = sem + (long long)b * T * D;
    for (int t = wave; t < T; t += 4) {
        float s = 0.f;
        for (int c = lane * 4; c < D; c += 256) {
            const float4 v = *reinterpret_cast<const float4*>(x + (long long)t * D + c);
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        s = wave_sum(s);
        if (lane == 0) inv[t] = 1.f / fmaxf(sqrtf(s), 1e-8f);  // F.cosine_similarity: x / clamp_min(||x||, eps)
    }
    __syncthreads();
    for (int t = wave; t < T - 1; t += 4) {
        const float ia = inv[t], ib = inv[t + 1];
        float s = 0.f;
        for (int c = lane * 4; c < D; c += 256) {
            const float4 a = *reinterpret_cast<const float4*>(x + (long long)t * D + c);
            const float4 bb = *reinterpret_cast<const float4*>(x + (long long)(t + 1) * D + c);
            s += (a.x * ia) * (bb.x * ib) + (a.y * ia) * (bb.y * ib) + (a.z * ia) * (bb.z * ib) + (a.w * ia) * (bb.w * ib);
        }
        s = wave_sum(s);
        if (lane == 0) sim[t] = s;
    }
    __syncthreads();
    if (tid == 0) {
        int g = -1, in_seg = 0;
        for (int t = 0; t < T; ++t) {
            const bool new_group = (t == 0) || (sim[t - 1] <= thr);
            in_seg = new_group ? 0 : in_seg + 1;
            if (in_seg % max_tokens == 0) {
                ++g;
                start[(long long)b * T + g] = t;
                len[(long long)b * T + g] = 0;
            }
            seg[(long long)b * T + t] = g;
            len[(long long)b * T + g] += 1;
        }
        for (int k = g + 1; k < T; ++k) {
            start[(long long)b * T + k] = T;
            len[(long long)b * T + k] = 0;
        }
        nseg[b] = g + 1;
        atomicMax(gmax, g + 1);
    }
}

int launch_align(const float* sem, int B, int T, int D, float thr, int max_tokens, int* seg, int* start, int* len,
                 int* nseg, int* gmax, hipStream_t s) {
    QA_REQUIRE(D % 4 == 0 && T >= 1 && max_tokens >= 1, "align: bad shape");
    QA_HIP(hipMemsetAsync(gmax, 0, sizeof(int), s));
    hipLaunchKernelGGL(align_kernel, dim3(B), dim3(256), 2 * T * sizeof(float), s, sem, T, D, thr, max_tokens, seg, start, len,
                       nseg, gmax);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// QueryTokenAggregator input (mimi/transformer.py:766-806): frames interleaved with one query token after each group;
// query = mean of the group's frames + query_embedding; padded groups (g >= nseg[b]) sit at the end and carry the bare
// query embedding.  out [B, T + G, D].
__global__ __launch_bounds__(256) void agg_build_kernel(const float* __restrict__ feats, const int* __restrict__ seg,
                                                        const int* __restrict__ start, const int* __restrict__ len,
                                                        const int* __restrict__ nseg, const float* __restrict__ qemb,
                                                        float* __restrict__ out, int T, int G, int D) {
    const int b = blockIdx.y, p = blockIdx.x;  // p: source element, frames 0..T-1 then queries T..T+G-1
    const int S = T + G;
    float* ob = out + (long long)b * S * D;
    if (p < T) {
        const int dst = p + seg[(long long)b * T + p];
        for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4)
            *reinterpret_cast<float4*>(ob + (long long)dst * D + c) =
                *reinterpret_cast<const float4*>(feats + ((long long)b * T + p) * D + c);
        return;
    }
    const int g = p - T;
    if (g >= nseg[b]) {
        for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4)
            *reinterpret_cast<float4*>(ob + (long long)(T + g) * D + c) = *reinterpret_cast<const float4*>(qemb + c);
        return;
    }
    const int st = start[(long long)b * T + g], n = len[(long long)b * T + g];
    const int dst = st + n + g;
    const float inv = 1.f / (float)n;
    for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = st; t < st + n; ++t) {
            const float4 v = *reinterpret_cast<const float4*>(feats + ((long long)b * T + t) * D + c);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        const float4 q = *reinterpret_cast<const float4*>(qemb + c);
        acc.x = acc.x * inv + q.x; acc.y = acc.y * inv + q.y; acc.z = acc.z * inv + q.z; acc.w = acc.w * inv + q.w;
        *reinterpret_cast<float4*>(ob + (long long)dst * D + c) = acc;
    }
}
int launch_agg_build(const float* feats, const int* seg, const int* start, const int* len, const int* nseg, const float* qemb,
                     float* out, int B, int T, int G, int D, hipStream_t s) {
    hipLaunchKernelGGL(agg_build_kernel, dim3(T + G, B), dim3(128), 0, s, feats, seg, start, len, nseg, qemb, out, T, G, D);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// transformer output at the query positions -> [B, G, D], zero for padded groups (mimi/transformer.py:812-824)
__global__ __launch_bounds__(128) void agg_gather_kernel(const float* __restrict__ x, const int* __restrict__ start,
                                                         const int* __restrict__ len, const int* __restrict__ nseg,
                                                         float* __restrict__ out, int T, int G, int D) {
    const int b = blockIdx.y, g = blockIdx.x;
    const bool valid = g < nseg[b];
    const int src = valid ? start[(long long)b * T + g] + len[(long long)b * T + g] + g : 0;
    for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) v = *reinterpret_cast<const float4*>(x + ((long long)b * (T + G) + src) * D + c);
        *reinterpret_cast<float4*>(out + ((long long)b * G + g) * D + c) = v;
    }
}
int launch_agg_gather(const float* x, const int* start, const int* len, const int* nseg, float* out, int B, int T, int G,
                      int D, hipStream_t s) {
    hipLaunchKernelGGL(agg_gather_kernel, dim3(G, B), dim3(128), 0, s, x, start, len, nseg, out, T, G, D);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// library indices [B*G, Q] -> reference layout [B, Q, G] with the group length injected:
// code' = (len - 1) * K + code (codec_adaptive.py:68-73; len = 0 for padded groups gives code - K, as in the reference)
__global__ void codes_inject_kernel(const long long* __restrict__ idx, const int* __restrict__ len, long long* __restrict__ dst,
                                    int B, int T, int G, int Q, int K) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)B * Q * G) return;
    const int g = (int)(gid % G);
    const int q = (int)((gid / G) % Q);
    const int b = (int)(gid / ((long long)G * Q));
    dst[gid] = (long long)(len[(long long)b * T + g] - 1) * K + idx[((long long)b * G + g) * Q + q];
}
int launch_codes_inject(const long long* idx, const int* len, long long* dst, int B, int T, int G, int Q, int K,
                        hipStream_t s) {
    const long long total = (long long)B * Q * G;
    hipLaunchKernelGGL(codes_inject_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, idx, len, dst, B, T, G, Q, K);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

__device__ __forceinline__ long long floordiv_ll(long long a, long long b) {
    long long q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
    return q;
}

// total frames per item from length-injected codes: len = floor(code / K) + 1 of quantizer 0 (codec_adaptive.py:75-80)
__global__ void adaptive_frames_kernel(const long long* __restrict__ codes, int B, int Q, int G, int K, int* __restrict__ totals,
                                       int* __restrict__ tmax) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int tot = 0;
    for (int g = 0; g < G; ++g) {
        const long long l = floordiv_ll(codes[((long long)b * Q) * G + g], K) + 1;
        tot += l > 0 ? (int)l : 0;
    }
    totals[b] = tot;
    atomicMax(tmax, tot);
}
int launch_adaptive_frames(const long long* codes, int B, int Q, int G, int K, int* totals, int* tmax, hipStream_t s) {
    QA_HIP(hipMemsetAsync(tmax, 0, sizeof(int), s));
    hipLaunchKernelGGL(adaptive_frames_kernel, dim3((unsigned)ceil_div(B, 64)), dim3(64), 0, s, codes, B, Q, G, K, totals, tmax);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// _deaggregate_features_from_token_lengths on index tensors (modeling_flexicodec_new.py:1007-1041, codec_adaptive.py:184-189):
// codes [B, Q, G] (length-injected) -> plain indices [B*T, Q], each group repeated len times, rows past an item's total = 0.
// Lengths come from `len_codes` (the reference ends up using the semantic stream's lengths for both streams).
__global__ __launch_bounds__(64) void deaggregate_kernel(const long long* __restrict__ codes, const long long* __restrict__ len_codes,
                                                          long long* __restrict__ out, int Q, int G, int T, int K) {
    const int b = blockIdx.x;
    __shared__ int s_start[1024];
    __shared__ int s_n;
    if (threadIdx.x == 0) {
        int pos = 0;
        for (int g = 0; g < G && g < 1024; ++g) {
            s_start[g] = pos;
            const long long l = floordiv_ll(len_codes[((long long)b * Q) * G + g], K) + 1;
            pos += l > 0 ? (int)l : 0;
        }
        s_n = pos;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T * Q; i += blockDim.x) out[((long long)b * T) * Q + i] = 0;
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        const int st = s_start[g];
        const int en = (g + 1 < G) ? s_start[g + 1] : s_n;
        for (int q = 0; q < Q; ++q) {
            const long long c = codes[((long long)b * Q + q) * G + g];
            long long plain = c % K;
            if (plain < 0) plain += K;  // python modulo
            for (int t = st; t < en && t < T; ++t) out[((long long)b * T + t) * Q + q] = plain;
        }
    }
}
int launch_deaggregate(const long long* codes, const long long* len_codes, long long* out, int B, int Q, int G, int T, int K,
                       hipStream_t s) {
    QA_REQUIRE(G <= 1024, "deaggregate: %d groups per item exceed 1024", G);
    hipLaunchKernelGGL(deaggregate_kernel, dim3(B), dim3(64), 0, s, codes, len_codes, out, Q, G, T, K);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// number of codes outside [0, limit): what F.embedding would refuse (Codec.decode, vq/codec.py:183-184)
__global__ __launch_bounds__(256) void codes_check_kernel(const long long* __restrict__ codes, long long n, long long lo, long long limit,
                                                          unsigned long long* __restrict__ bad) {
    unsigned long long local = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long c = codes[i];
        local += (c < lo || c >= limit) ? 1ull : 0ull;
    }
    local += __shfl_xor(local, 32, 64);
    local += __shfl_xor(local, 16, 64);
    local += __shfl_xor(local, 8, 64);
    local += __shfl_xor(local, 4, 64);
    local += __shfl_xor(local, 2, 64);
    local += __shfl_xor(local, 1, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(bad, local);  // integer count: order-independent
}
int launch_codes_check(const long long* codes, long long n, long long limit, unsigned long long* bad, hipStream_t s) {
    QA_HIP(hipMemsetAsync(bad, 0, sizeof(unsigned long long), s));
    return launch_codes_count(codes, n, 0, limit, bad, s);
}
// *bad += number of entries outside [lo, limit): no memset, no copy, no synchronisation (qa_codes_check_async)
int launch_codes_count(const long long* codes, long long n, long long lo, long long limit, unsigned long long* bad, hipStream_t s) {
    if (n <= 0) return QA_OK;
    const unsigned grid = (unsigned)std::min<long long>(ceil_div(n, 256), 1024);
    hipLaunchKernelGGL(codes_check_kernel, dim3(grid), dim3(256), 0, s, codes, n, lo, limit, bad);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// torchaudio.functional.resample's polyphase FIR (transforms.Resample, HCodec-2.0/audio_tokenizer.py:44,51):
//   out[b, q * new + i] = sum_j kernel[i][j] * padded[b, q * orig + j],  padded = wav with `width` zeros in front and
//   width + orig behind, kernel [new][2 * width + orig]; the output is cut to ceil(new * T / orig) samples.
// HBM-bound (one read of the waveform, 1/3 of it written for 48 -> 16 kHz): one thread per output sample, taps from LDS.
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ wav, const float* __restrict__ taps, float* __restrict__ out,
                                                       long long T, long long T_out, int orig, int nw, int width, int ktaps,
                                                       int use_lds) {
    extern __shared__ float s_taps[];
    if (use_lds) {  // small filter banks (48 -> 16 kHz: 1 x 41 taps) live in LDS; large ones (44.1 -> 16 kHz: 160 x 475) stay in L2
        for (int i = threadIdx.x; i < nw * ktaps; i += 256) s_taps[i] = taps[i];
        __syncthreads();
    }
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= T_out) return;
    const int b = blockIdx.y;
    const long long q = o / nw;
    const int ph = (int)(o - q * nw);
    const float* x = wav + (long long)b * T;
    const float* k = (use_lds ? s_taps : taps) + ph * ktaps;
    const long long base = q * orig - width;
    float acc = 0.f;
    for (int j = 0; j < ktaps; ++j) {
        const long long t = base + j;
        const float v = (t >= 0 && t < T) ? x[t] : 0.f;
        acc = fmaf(k[j], v, acc);
    }
    out[(long long)b * T_out + o] = acc;
}
int launch_resample(const float* wav, const float* taps, float* out, int B, long long T, long long T_out, int orig, int nw, int width,
                    int ktaps, hipStream_t s) {
    const size_t bytes = (size_t)nw * ktaps * sizeof(float);
    const int use_lds = bytes <= 48 * 1024;
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)ceil_div(T_out, 256), (unsigned)B), dim3(256), use_lds ? bytes : 0, s, wav, taps, out, T,
                       T_out, orig, nw, width, ktaps, use_lds);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// ------------------------------------------------------------------------------------------------
// RingKVCache.complete() write (mimi/transformer.py:243-250): cache[:, (pos0 + t) % cap] = k[:, t]
__global__ __launch_bounds__(256) void ring_append_kernel(const float* __restrict__ k, const float* __restrict__ v, long long ld,
                                                          float* __restrict__ kc, float* __restrict__ vc, int B, int T, int d,
                                                          int cap, int pos0) {
    const int d4 = d >> 2;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)B * T * d4) return;
    const int c = (int)(gid % d4) * 4;
    const long long row = gid / d4;
    const int b = (int)(row / T), t = (int)(row - (long long)b * T);
    const long long dst = ((long long)b * cap + (pos0 + t) % cap) * d + c;
    *reinterpret_cast<float4*>(kc + dst) = *reinterpret_cast<const float4*>(k + row * ld + c);
    *reinterpret_cast<float4*>(vc + dst) = *reinterpret_cast<const float4*>(v + row * ld + c);
}

int launch_ring_append(const float* k, const float* v, long long ld, float* kc, float* vc, int B, int T, int d, int cap, int pos0,
                       hipStream_t s) {
    QA_REQUIRE(d % 4 == 0 && ld % 4 == 0 && T >= 1 && T <= cap && pos0 >= 0, "ring_append: T=%d cap=%d d=%d (a chunk must fit the ring)", T, cap, d);
    const long long total = (long long)B * T * (d / 4);
    hipLaunchKernelGGL(ring_append_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, k, v, ld, kc, vc, B, T, d, cap, pos0);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

}  // namespace qa
