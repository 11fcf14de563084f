// common.h - shared host/device helpers for libquarkaudio_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "knobs.h"
#include "quarkaudio.h"

namespace qa {

void set_error(const char* fmt, ...);

#define QA_HIP(expr)                                                                                    \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            qa::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);   \
            return QA_ERR_HIP;                                                                          \
        }                                                                                               \
    } while (0)

#define QA_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            qa::set_error(__VA_ARGS__);       \
            return QA_ERR_INVALID;            \
        }                                     \
    } while (0)

#define QA_TRY(expr)               \
    do {                           \
        int s_ = (expr);           \
        if (s_ != QA_OK) return s_; \
    } while (0)

#define QA_LAUNCH_CHECK() QA_HIP(hipGetLastError())

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only: raise a kernel's dynamic-LDS limit once per
// (kernel, device), thread-safe (api.cpp).  Call outside any stream capture.
int raise_dynamic_lds(const void* kernel, int bytes);

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

// ---- activation codes shared by kernels (match qa_conv_args) ----
enum { ACT_NONE = 0, ACT_ELU = 1, ACT_GELU = 2, ACT_SILU = 3, ACT_SNAKE = 4, ACT_TANH = 5 };  // SNAKE needs ConvParams::alpha (conv_gemm epilogue only)
enum { PAD_ZERO = 0, PAD_REFLECT = 1 };

// ELU(alpha=1).  exp(x) - 1 instead of expm1f: ocml's expm1f brings divergent slow paths (and scratch spills) into the GEMM
// staging code; the absolute error of the difference is <= 1 ulp(1) = 6e-8, i.e. fp32 rounding noise of the O(0.1..1)
// activations it feeds.
__device__ __forceinline__ float elu_f(float x) { return x > 0.f ? x : __expf(x) - 1.f; }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case ACT_ELU: return elu_f(v);
        case ACT_GELU: return gelu_erf_f(v);
        case ACT_SILU: return silu_f(v);
        case ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// Source-frame resolution for the padded convolutions.
//   zero:    frames outside [0, L) read as 0 (returns -1)
//   reflect: SConv1d's pad1d (encoder_modules/conv.py:79-96 of the reference): when L <= max_pad the
//            signal is first zero-extended to Lp = max_pad + 1 frames, reflected, then trimmed, so a
//            reflected index may land in the zero extension (returns -1).
__host__ __device__ __forceinline__ int resolve_frame(int r, int L, int Lp, int pad_mode) {
    if (pad_mode == PAD_REFLECT) {
        if (r < 0) r = -r;
        else if (r >= Lp) r = 2 * (Lp - 1) - r;
        return (r >= 0 && r < L) ? r : -1;
    }
    return (r >= 0 && r < L) ? r : -1;
}

// wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Row-norm arithmetic of rownorm_kernel (ew.hip): ONE statement of every step, compiled with contraction off (`fp contract(off)`: the instructions
// carry no `contract` flag, so they stay unfused wherever they are inlined) and fmaf where a fused multiply-add is meant - any kernel that repeats
// these statements on the same lane map (lane l of the row's wave holds columns 4 l + 256 i .. + 3) produces the bits of rownorm_kernel.  r06 used
// that for a norm-fused conv_gemm operand (bit-identical, 4 % SLOWER on H-Codec 1.5: profiles/r06_norm_fused_gemm_ab.txt; removed).
enum { NORM_NONE = 0, NORM_RMS = 1, NORM_LAYER = 2 };  // RMSNorm: transformer.py:77-96 (eps inside the sqrt of mean(x^2)); LayerNorm: biased variance
__device__ __forceinline__ float norm_add(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float norm_sum4(float x, float y, float z, float w) {
#pragma clang fp contract(off)
    return ((x + y) + z) + w;
}
__device__ __forceinline__ float norm_sq4(float x, float y, float z, float w) {
#pragma clang fp contract(off)
    return fmaf(w, w, fmaf(z, z, fmaf(y, y, x * x)));
}
__device__ __forceinline__ float norm_csq4(float x, float y, float z, float w, float mean) {
#pragma clang fp contract(off)
    const float a = x - mean, b = y - mean, c = z - mean, d = w - mean;
    return fmaf(d, d, fmaf(c, c, fmaf(b, b, a * a)));
}
__device__ __forceinline__ float norm_mean(float sum, int C) {
#pragma clang fp contract(off)
    return sum / (float)C;
}
__device__ __forceinline__ float norm_rstd(float sumsq, int C, float eps) {
#pragma clang fp contract(off)
    const float var = sumsq / (float)C;
    return rsqrtf(var + eps);
}
// (v - mean) * rstd * w (+ b): mean = 0 for RMSNorm, b = 0 without a bias
__device__ __forceinline__ float norm_apply(float v, float mean, float rstd, float w, float b) {
#pragma clang fp contract(off)
    const float t = (v - mean) * rstd;
    return fmaf(t, w, b);
}

// ---- kernel launchers (host) ----
struct ConvParams {
    const float* x;
    const float* w;
    const float* bias;
    const float* gamma;
    const float* res;
    const float* gate;
    float* y;
    long long ldx, ldy, ldr, ldg;
    int B, T_in, C_in, T_out, N, K, M;
    int ksize, stride, pad_left, pad_mode, Lp;
    int prologue, act, post_act;
    int algo_n, algo_k;  // un-padded N / K for the algorithmic FLOP count (0 = use N / K)
    int xcd_swizzle;
    int panel;  // > 0: column panels of this many tiles, row tiles fastest inside a panel (QA_GEMM_PANEL)
    int in_rep;  // >1: the input is read as if every frame were repeated in_rep times (x.repeat_interleave, H-Codec 2.0 decoder)
    int vec_epi;  // epilogue may move float4 (set by launch_conv_gemm from N, leading dimensions and pointer alignment)
    unsigned rep_magic, rep_one;  // r / in_rep == __umulhi(r, rep_magic) + r * rep_one  (branch-free; set by launch_conv_gemm)
    // fused interleaved-pair RoPE on output channels n < rope_n (the q and k parts of a fused QKV projection; mimi module/rope.py:13-69):
    // (y[2i], y[2i+1]) <- (y[2i] c - y[2i+1] s, y[2i+1] c + y[2i] s) with (c, s) = rope[(t * rope_hd/2 + i) * 2 + {0,1}], t = rope_pos0 + m % rope_T, i = (n % rope_hd) / 2
    const float* rope;
    int rope_n, rope_hd, rope_T, rope_pos0;
    int dilation;  // tap j reads frame t * stride - pad_left + j * dilation (0 / 1 = dense); zero padding only
    // Snake activation x + sin^2(alpha x) / (alpha + 1e-9) with a per-output-channel alpha (BiCodec wave generator, blocks/layers.py:31-36):
    // `alpha` serves act / post_act == ACT_SNAKE; y2 (optional) receives snake(final value, alpha2) next to y, so that a residual
    // unit's input exists both raw (for the skip) and activated (for its first convolution) without another pass over HBM
    const float* alpha;
    float* y2;
    const float* alpha2;
    long long ldy2;
    // Arg-min epilogue (RVQ distance GEMM, rvq.hip): instead of storing y = x w^T, every group of 32 consecutive output columns of a row
    // is reduced to its nearest code - dist = (am_x2[m] - 2 y[m, n]) + am_e2[n] (core_vq.py:225-229 association), lowest n on ties - and
    // only (dist, n) goes to am_dist / am_idx [M, am_ld]: the [n_vec, K] product matrix never reaches memory.  y may be null.
    const float* am_x2;
    const float* am_e2;
    float* am_dist;
    int* am_idx;
    int am_ld;
};

// Live measurement hook (bench.py): when enabled every conv_gemm launch is bracketed by HIP events on its own stream.
enum { PROF_CFG_128x32 = 0, PROF_CFG_128x64 = 1, PROF_CFG_128x128 = 2, PROF_CFG_64x128 = 3, PROF_CFG_64x64 = 4, PROF_NCFG = 5 };
bool profile_enabled();
// Byte-bound kernels (north_star: "rocprof HBM GB/s"): with qa_profile_begin_ex(2 | ...) their launches are bracketed by HIP events too,
// each with its ALGORITHMIC bytes (every input and output element once), reduced per kind by qa_profile_end_hbm.
enum HbmKind { HK_ROWNORM = 0, HK_DWCONV_LN, HK_GROUPNORM, HK_ROPE, HK_ISTFT_SPEC, HK_ISTFT_OLA, HK_STFT_POST, HK_RVQ_LOOKUP, HK_RVQ_PICK,
               HK_SEANET_FRONT, HK_CONV_IN, HK_NKINDS };
bool profile_hbm_enabled();
struct HbmProf {  // scope guard around ONE launch (or the launches of one logical pass) on stream s
    hipStream_t s;
    bool on;
    HbmProf(int kind, double bytes, hipStream_t stream);
    ~HbmProf();
};
bool serial_mode();  // qa_set_serial / QA_SERIAL=1: no internal stream concurrency (every kernel alone on the device)
void profile_record_begin(int cfg, double flops, double bytes, hipStream_t s, const ConvParams* p = nullptr);
void profile_record_end(hipStream_t s);

int launch_conv_gemm(const ConvParams& p, hipStream_t stream);
int conv_params_from_args(const qa_conv_args& a, ConvParams* p);

}  // namespace qa
