// lstm.hip - the nn.LSTM(d, d, 1, batch_first=True) that every codec transformer layer runs before its QKV
// projection (reference: QuarkAudio-HCodec/HCodec-1.0/vq/encoder_modules/transformer.py:115,133).
//
// The input half (x W_ih^T + b_ih + b_hh) is one big batched GEMM done by conv_gemm; what remains is the strictly
// sequential recurrence  gates_t = xw_t + h_{t-1} W_hh^T.  Round-1 structure: one launch per time step (the kernel
// boundary is the cross-CU synchronisation; MI355X_MICROARCH "boundary" row: ~1.5-1.9 us, cheaper than a software
// grid barrier), d/4 workgroups per step.  A workgroup owns 4 hidden units x 4 gates = 16 rows of W_hh (rows
// pre-permuted to (unit, gate) order at load time) for ALL batch rows, splits K = d over its 8 waves, runs
// v_mfma_f32_16x16x4_f32 with batch as the M dimension, reduces the 8 partial tiles through LDS and applies the
// cell update in the same kernel, so gates never touch HBM.
#include "kernels.h"

namespace qa {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// MT = number of 16-row batch tiles (B <= 16*MT)
template <int MT>
__global__ __launch_bounds__(512) void lstm_step_kernel(const float* __restrict__ xw, const float* __restrict__ w_hh,
                                                        float* __restrict__ h_out, float* __restrict__ c_state, int B,
                                                        int T, int d, int t) {
    __shared__ float part[8][MT][16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 16;  // first permuted gate row of this workgroup
    const int li = lane & 15, kq = lane >> 4;

    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // issue the epilogue's HBM reads (input projection, previous cell state) first: their latency hides under the matvec
    const bool epi = tid < B * 4;
    const int eb = tid >> 2, eu = tid & 3;
    const int unit = blockIdx.x * 4 + eu;
    float4 xg = make_float4(0.f, 0.f, 0.f, 0.f);
    float c_prev = 0.f;
    if (epi) {
        xg = *reinterpret_cast<const float4*>(xw + ((long long)eb * T + t) * 4 * d + (long long)unit * 4);
        if (t > 0) c_prev = c_state[(long long)eb * d + unit];
    }

    if (t > 0) {
        const int kw = d / 8;
        const int k0 = wave * kw;
        const float* wrow = w_hh + (long long)(n0 + li) * d + k0 + 4 * kq;
        const float* hrow[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            int b = m * 16 + li;
            if (b >= B) b = B - 1;  // rows past B are computed on a valid row and discarded
            hrow[m] = h_out + ((long long)b * T + (t - 1)) * d + k0 + 4 * kq;
        }
        for (int g = 0; g < kw; g += 16) {
            const float4 wv = *reinterpret_cast<const float4*>(wrow + g);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float4 hv = *reinterpret_cast<const float4*>(hrow[m] + g);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv.x, wv.x, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv.y, wv.y, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv.z, wv.z, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv.w, wv.w, acc[m], 0, 0, 0);
            }
        }
    }
    // C layout of the 16x16 MFMA: col = lane & 15 (gate row), row = 4 * (lane >> 4) + r (batch row)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][m][4 * kq + r][li] = acc[m][r];
    __syncthreads();

    if (epi) {
        const int b = eb, u = eu;
        float g4[4] = {xg.x, xg.y, xg.z, xg.w};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float s = g4[g];
#pragma unroll
            for (int w = 0; w < 8; ++w) s += part[w][b >> 4][b & 15][u * 4 + g];
            g4[g] = s;
        }
        const float ig = sigmoid_f(g4[0]), fg = sigmoid_f(g4[1]), gg = tanhf(g4[2]), og = sigmoid_f(g4[3]);
        const long long ci = (long long)b * d + unit;
        const float c_new = fg * c_prev + ig * gg;
        c_state[ci] = c_new;
        h_out[((long long)b * T + t) * d + unit] = og * tanhf(c_new);
    }
}

int launch_lstm(const float* xw, const float* w_hh_ug, float* h_out, float* c_state, int B, int T, int d,
                hipStream_t s) {
    QA_REQUIRE(d % 128 == 0, "lstm: hidden size %d must be a multiple of 128", d);
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int bn = std::min(64, B - b0);
        const float* xw_b = xw + (long long)b0 * T * 4 * d;
        float* h_b = h_out + (long long)b0 * T * d;
        float* c_b = c_state + (long long)b0 * d;
        const int mt = (int)ceil_div(bn, 16);
        for (int t = 0; t < T; ++t) {
            switch (mt) {
                case 1: hipLaunchKernelGGL(lstm_step_kernel<1>, dim3(d / 4), dim3(512), 0, s, xw_b, w_hh_ug, h_b, c_b, bn, T, d, t); break;
                case 2: hipLaunchKernelGGL(lstm_step_kernel<2>, dim3(d / 4), dim3(512), 0, s, xw_b, w_hh_ug, h_b, c_b, bn, T, d, t); break;
                case 3: hipLaunchKernelGGL(lstm_step_kernel<3>, dim3(d / 4), dim3(512), 0, s, xw_b, w_hh_ug, h_b, c_b, bn, T, d, t); break;
                default: hipLaunchKernelGGL(lstm_step_kernel<4>, dim3(d / 4), dim3(512), 0, s, xw_b, w_hh_ug, h_b, c_b, bn, T, d, t); break;
            }
        }
        QA_LAUNCH_CHECK();
    }
    return QA_OK;
}

}  // namespace qa
