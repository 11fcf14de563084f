// seanet_front.hip - the HBM-bound front of the SEANet encoder as ONE kernel per stage.
//
// Reference (QuarkAudio-HCodec/HCodec-1.0/vq/encoder_modules/seanet.py:34-76,121-135; conv.py:195-211):
//     x = conv0(wav)                                   SConv1d 1 -> C, k7, reflect pad (3, 3)
//     a = ELU( shortcut_1x1(x) + 1x1( ELU( k3( ELU(x) ) ) ) )      SEANetResnetBlock + the ELU in front of the strided conv
// At C = 32 and 32 x 10 s these are [5.12 M, 32] tensors: run as separate launches the 655 MB activation makes four to five round
// trips through HBM (conv_in write, shortcut read + write, k3 read + write of the 16 -> 32 padded hidden, 1x1 two reads + write).
// Here a workgroup keeps a 128-frame tile (+ halo) of x in LDS, chains the two small contractions on the matrix cores with the
// hidden tile never leaving the CU, and writes `a` once: x never exists in HBM, the hidden width is not padded in memory, the
// block's traffic is 4 B in per frame and 4 C B out.
//
// gfx950 mapping: 256 threads = 4 waves, each owns 32 frames.  Both contractions use v_mfma_f32_32x32x2_f32 in the operand
// convention of conv_gemm.hip (weights = row operand, activations = column operand: a lane ends up with 4 consecutive output
// channels of one frame per register quad).  The k3 taps are plain row offsets into the LDS tile (frame t - 1, t, t + 1), the
// 1x1 pair is one contraction over the concatenated K = [x | hidden].  All weights (<= 20 KB at C = 32) sit in LDS for the life of
// the (persistent) workgroup.  Output rows go through a per-wave LDS staging so that a store instruction writes 1 KB contiguous.
#include <algorithm>
#include <cstdlib>

#include "kernels.h"

namespace qa {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SeanetFrontParams {
    const float* wav;  // [B, L]
    const float *w0, *b0;    // conv0, library layout [7][C], [C]
    const float *w3, *b3;    // k3: [32][3][C] (rows >= hid are zero), [32]
    const float *wsc, *bsc;  // shortcut 1x1: [C][C], [C]
    const float *wpw, *bpw;  // 1x1 after the k3: [C][32] (columns >= hid are zero), [C]
    float* a;                // [B, L, C]
    int B, L, hid8;          // hid8 = hidden width rounded up to 8 (K steps of the second contraction)
    int pl3, Lp3, pl0, Lp0;  // left paddings and short-input lengths of the k3 / k7 reflect pads
};

__device__ __forceinline__ f32x4 elu4(f32x4 v) {
    const f32x4 r = {elu_f(v.x), elu_f(v.y), elu_f(v.z), elu_f(v.w)};
    return r;
}

template <int C>
__global__ __launch_bounds__(256) void seanet_block_kernel(const SeanetFrontParams p) {
    constexpr int ROWS = 128, XR = ROWS + 2, LDX = C + 4, HP = 32;
    constexpr int K3 = 3 * C, LD3 = K3 + 4, KC = C + HP, LDC = KC + 4;  // row strides = 4 mod 32 floats: the 16 lanes of a ds_read_b128 phase hit distinct banks
    constexpr int NT = C / 32, HSLD = C + 4, C4 = C / 4;
    constexpr int WSN = XR + 6;  // wav samples under one x tile (k7)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* XS = smem;               // [XR][LDX]  x tile (raw): the shortcut operand
    float* XE = XS + XR * LDX;      // [XR][LDX]  ELU(x): the k3 operand
    float* W3 = XE + XR * LDX;      // [HP][LD3]
    float* WC = W3 + HP * LD3;      // [C][LDC]   (W_shortcut | W_1x1)
    float* HS = XE;                 // [4][32][HSLD]  per wave: hidden tile, then the output staging - ALIASES the ELU(x) tile, which is
                                    // dead once every wave has finished its k3 contraction (barrier below); keeps the workgroup at 61 KB
    static_assert(4 * 32 * HSLD <= XR * LDX, "hidden / output staging must fit in the ELU(x) tile");
    float* b3s = WC + C * LDC;
    float* bcs = b3s + HP;
    float* b0s = bcs + C;
    float* w0s = b0s + C;           // [7][C]
    float* WS = w0s + 7 * C;        // [2][WSN] wav window of the current / next tile (reflect padding of the k7 resolved at staging)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    for (int e = tid; e < HP * (K3 / 4); e += 256) {
        const int n = e / (K3 / 4), c4 = (e % (K3 / 4)) * 4;
        *reinterpret_cast<f32x4*>(W3 + n * LD3 + c4) = *reinterpret_cast<const f32x4*>(p.w3 + (long long)n * K3 + c4);
    }
    for (int e = tid; e < C * (KC / 4); e += 256) {
        const int n = e / (KC / 4), c4 = (e % (KC / 4)) * 4;
        const f32x4 v = c4 < C ? *reinterpret_cast<const f32x4*>(p.wsc + (long long)n * C + c4)
                               : *reinterpret_cast<const f32x4*>(p.wpw + (long long)n * HP + (c4 - C));
        *reinterpret_cast<f32x4*>(WC + n * LDC + c4) = v;
    }
    for (int e = tid; e < HP; e += 256) b3s[e] = p.b3[e];
    for (int e = tid; e < C; e += 256) {
        bcs[e] = p.bsc[e] + p.bpw[e];
        b0s[e] = p.b0[e];
    }
    for (int e = tid; e < 7 * C; e += 256) w0s[e] = p.w0[e];

    const int L = p.L;
    const int tiles_per_clip = (L + ROWS - 1) / ROWS, n_tiles = p.B * tiles_per_clip;
    float* HSw = HS + wave * 32 * HSLD;
    // sample u of a tile's window is wav[reflect(r0 - pl3 - pl0 + u)]; one sample per thread, fetched one tile ahead
    auto wav_fetch = [&](int tile_) -> float {
        if (tile_ >= n_tiles || tid >= WSN) return 0.f;
        const int b_ = tile_ / tiles_per_clip, r0_ = (tile_ - b_ * tiles_per_clip) * ROWS;
        const int s_ = resolve_frame(r0_ - p.pl3 - p.pl0 + tid, L, p.Lp0, PAD_REFLECT);
        return s_ >= 0 ? p.wav[(long long)b_ * L + s_] : 0.f;
    };
    if (tid < WSN) WS[tid] = wav_fetch(blockIdx.x);
    int cur = 0;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, cur ^= 1) {
        const int b = tile / tiles_per_clip, r0 = (tile - b * tiles_per_clip) * ROWS;
        __syncthreads();  // the previous tile is no longer read (first trip: orders the weight / window stores too)
        const float wnext = wav_fetch(tile + gridDim.x);  // in flight under this whole tile
        const float* ws = WS + cur * WSN;
        // ---- x tile: frames r0 - pl3 .. r0 - pl3 + XR - 1 with the k3's reflect padding resolved per row
        for (int e = tid; e < XR * C4; e += 256) {
            const int i = e / C4, c4 = (e - i * C4) * 4;
            const int q = r0 - p.pl3 + i;
            const int src = resolve_frame(q, L, p.Lp3, PAD_REFLECT);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (src >= 0) {
                {
                    v = *reinterpret_cast<const f32x4*>(b0s + c4);
                    if (src == q) {  // a frame of the clip itself: its 7 samples are ws[i .. i + 6]
#pragma unroll
                        for (int j = 0; j < 7; ++j) {  // same accumulation order as conv_in_kernel (ew.hip): bit-identical x
                            const float xv = ws[i + j];
                            const f32x4 w = *reinterpret_cast<const f32x4*>(w0s + j * C + c4);
                            v.x = fmaf(xv, w.x, v.x); v.y = fmaf(xv, w.y, v.y); v.z = fmaf(xv, w.z, v.z); v.w = fmaf(xv, w.w, v.w);
                        }
                    } else {  // a halo row reflected at a clip edge (at most two rows per clip end): straight from memory
                        const float* wb = p.wav + (long long)b * L;
                        for (int j = 0; j < 7; ++j) {
                            const int s = resolve_frame(src - p.pl0 + j, L, p.Lp0, PAD_REFLECT);
                            const float xv = s >= 0 ? wb[s] : 0.f;
                            const f32x4 w = *reinterpret_cast<const f32x4*>(w0s + j * C + c4);
                            v.x = fmaf(xv, w.x, v.x); v.y = fmaf(xv, w.y, v.y); v.z = fmaf(xv, w.z, v.z); v.w = fmaf(xv, w.w, v.w);
                        }
                    }
                }
            }
            *reinterpret_cast<f32x4*>(XS + i * LDX + c4) = v;
            *reinterpret_cast<f32x4*>(XE + i * LDX + c4) = elu4(v);  // ELU(0) = 0: the zero extension of a short clip stays zero
        }
        __syncthreads();

        // ---- hidden = ELU(k3(ELU(x)) + b3): K = 3 C, taps are row offsets 0, 1, 2 of the tile
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {
            const float* arow = XE + (wave * 32 + fr) * LDX + 4 * fh;
            const float* wrow = W3 + fr * LD3 + 4 * fh;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int c0 = 0; c0 < C; c0 += 8) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(arow + j * LDX + c0);
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + j * C + c0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, av.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, av.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, av.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, av.w, acc, 0, 0, 0);
                }
        }
        __syncthreads();  // every wave is done with the ELU(x) tile: it becomes the hidden tile
        // D layout (operands swapped as in conv_gemm.hip): frame <- lane & 31, channel <- (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = 8 * g + 4 * fh;
            if (n >= p.hid8) continue;  // padded hidden channels: never read by the second contraction
            f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            v += *reinterpret_cast<const f32x4*>(b3s + n);
            *reinterpret_cast<f32x4*>(HSw + fr * HSLD + n) = elu4(v);
        }
        __syncthreads();

        // ---- s = [x | hidden] [W_sc | W_1x1]^T + (b_sc + b_1x1)
        f32x16 acc2[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
        {
            const float* xrow = XS + (wave * 32 + fr + p.pl3) * LDX + 4 * fh;
            const float* wrow = WC + fr * LDC + 4 * fh;
#pragma unroll
            for (int c0 = 0; c0 < C; c0 += 8) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(xrow + c0);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + t * 32 * LDC + c0);
                    acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, av.x, acc2[t], 0, 0, 0);
                    acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, av.y, acc2[t], 0, 0, 0);
                    acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, av.z, acc2[t], 0, 0, 0);
                    acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, av.w, acc2[t], 0, 0, 0);
                }
            }
            const float* hrow = HSw + fr * HSLD + 4 * fh;
            for (int c0 = 0; c0 < p.hid8; c0 += 8) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(hrow + c0);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + t * 32 * LDC + C + c0);
                    acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, av.x, acc2[t], 0, 0, 0);
                    acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, av.y, acc2[t], 0, 0, 0);
                    acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, av.z, acc2[t], 0, 0, 0);
                    acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, av.w, acc2[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();  // every lane of the wave has read its hidden rows: the region becomes the output staging
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = t * 32 + 8 * g + 4 * fh;
                f32x4 v = {acc2[t][4 * g], acc2[t][4 * g + 1], acc2[t][4 * g + 2], acc2[t][4 * g + 3]};
                v += *reinterpret_cast<const f32x4*>(bcs + n);
                *reinterpret_cast<f32x4*>(HSw + fr * HSLD + n) = elu4(v);
            }
        __syncthreads();
        // a store instruction of the wave covers 64 / C4 whole frames = 1 KB contiguous
#pragma unroll
        for (int it = 0; it < 32 * C4 / 64; ++it) {
            const int e = it * 64 + lane, row = e / C4, c4 = (e - row * C4) * 4;
            const int t = r0 + wave * 32 + row;
            if (t < L) *reinterpret_cast<f32x4*>(p.a + ((long long)b * L + t) * C + c4) = *reinterpret_cast<const f32x4*>(HSw + row * HSLD + c4);
        }
        if (tid < WSN) WS[(cur ^ 1) * WSN + tid] = wnext;  // that buffer was last read before this tile's second barrier
    }
}

static size_t seanet_block_lds_bytes(int C) {
    const int XR = 130, LDX = C + 4, HP = 32;
    return sizeof(float) * ((size_t)2 * XR * LDX + (size_t)HP * (3 * C + 4) + (size_t)C * (C + HP + 4) + HP + 2 * C + 7 * C + 2 * (XR + 6));
}

bool seanet_front_supported(int C, int hid, int L) {
    const bool on = knob(K_SEANET_FUSED) != 0;
    // L >= 8: both reflect pads stay in their plain regime (no zero extension of a short clip)
    return on && C == 32 && hid >= 1 && hid <= 32 && L >= 8;
}

// k3 / pw in the padded library layouts of hcodec.cpp (k3 [32][3][C], pw [C][32]).
int launch_seanet_front(const float* wav, const float* w0, const float* b0, const float* w3, const float* b3, const float* wsc,
                        const float* bsc, const float* wpw, const float* bpw, float* a, int B, int L, int C, int hid, int causal,
                        hipStream_t s) {
    QA_REQUIRE(seanet_front_supported(C, hid, L) && wav && w0 && b0 && w3 && b3 && wsc && bsc && wpw && bpw && a, "seanet_front: unsupported "
               "shape C=%d hid=%d L=%d", C, hid, L);
    SeanetFrontParams p{};
    p.wav = wav; p.w0 = w0; p.b0 = b0; p.w3 = w3; p.b3 = b3; p.wsc = wsc; p.bsc = bsc; p.wpw = wpw; p.bpw = bpw; p.a = a;
    p.B = B; p.L = L; p.hid8 = (int)round_up(hid, 8);
    p.pl3 = causal ? 2 : 1;
    p.pl0 = causal ? 6 : 3;
    const int mp3 = causal ? 2 : 1, mp0 = causal ? 6 : 3;
    p.Lp3 = L <= mp3 ? mp3 + 1 : L;
    p.Lp0 = L <= mp0 ? mp0 + 1 : L;
    const size_t lds = seanet_block_lds_bytes(C);
    const long long n_tiles = (long long)B * ceil_div(L, 128);
    const unsigned grid = (unsigned)std::min<long long>(n_tiles, 512);  // persistent: two workgroups per CU, weights staged once each
    QA_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(seanet_block_kernel<32>), (int)lds));
    HbmProf prof_(HK_SEANET_FRONT, 4.0 * ((double)B * L + (double)B * L * C), s);  // wav in, the block output `a` written once
    hipLaunchKernelGGL((seanet_block_kernel<32>), dim3(grid), dim3(256), lds, s, p);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

}  // namespace qa
