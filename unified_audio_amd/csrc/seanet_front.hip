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
    int B, L;
    int pl3, Lp3, pl0, Lp0;  // left paddings and short-input lengths of the k3 / k7 reflect pads
};

__device__ __forceinline__ f32x4 elu4(f32x4 v) {
    const f32x4 r = {elu_f(v.x), elu_f(v.y), elu_f(v.z), elu_f(v.w)};
    return r;
}

// r05: every contraction of the block on v_mfma_f32_16x16x4_f32 with the weights in REGISTERS.
//   * The k3's hidden width is 16: one 16-channel MFMA row block holds it exactly - the round-2 kernel padded it to the 32 rows of the
//     32x32x2 form and spent half of that contraction on zeros (48 MFMAs x 64 cycles per 32-frame wave tile; now 48 x 32).
//   * conv0 (k7, 1 -> 32) is a K = 8 contraction over a sliding window of the wave: A = w0 (k = 7 padded with a zero row), B[k][frame] =
//     the padded signal at frame + k, C-in = the bias - i.e. the SAME fma chain in the same order as conv_in_kernel (ew.hip): x is still
//     bit-identical to the standalone kernel's, and ~110 of a thread's ~400 vector instructions per tile are gone.
//   * All weights live in 52 VGPRs per lane for the life of the persistent workgroup (lane (i = lane & 15, kq = lane >> 4) holds row i of
//     a 16-row block at k = 16 s + 4 kq .. + 3): no weight traffic through LDS; LDS holds the two x tiles and a wave-private hidden tile per wave (49 KB: 3
//     workgroups per CU); TWO workgroup barriers per tile (x tile written / previous tile no longer read) instead of five; outputs are stored from registers.
// Operand convention as before (weights = the MFMA's row operand, activations = its column operand): D gives lane (frame = lane & 15)
// the 4 consecutive channels 4 (lane >> 4) .. + 3 of its 16-channel block: float4 stores everywhere.
// The K slots of a 16-wide chunk are visited in the order k = e + 4 kq (e = the float4 component = one MFMA, kq = the slot inside it):
// a fixed permutation, the same for every frame and batch row.
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int C>
__global__ __launch_bounds__(256) void seanet_block_kernel(const SeanetFrontParams p) {
    static_assert(C == 32, "two 16-channel blocks");
    constexpr int ROWS = 128, XR = ROWS + 2, LDX = C + 4, HLD = 16 + 4;
    constexpr int WSN = XR + 6;       // wav samples under one x tile (k7)
    constexpr int WSP = 9 * 16 + 8;   // padded window length: the 9th frame block of the conv0 phase reads up to sample 151
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* XS = smem;               // [XR][LDX]  x tile (raw): the shortcut operand
    float* XE = XS + XR * LDX;      // [XR][LDX]  ELU(x): the k3 operand
    float* HS = XE + XR * LDX;      // [4][32][HLD]  per WAVE: its 32 x 16 hidden tile - wave-private, so the k3 -> hidden -> 1x1 chain of a wave
                                    // needs no workgroup barrier (the round-2 kernel aliased it onto the ELU(x) tile: two barriers per tile)
    float* WS = HS + 4 * 32 * HLD;  // [2][WSP] wav window of the current / next tile (reflect padding of the k7 resolved at staging)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    // ---- weights -> registers (row li of each 16-row block, k = 16 s + 4 kq .. + 3)
    f32x4v w3r[6];     // k3 [16 hidden][3 taps x 32]: chunk s = tap * 2 + half
    f32x4v wcr[2][3];  // [x | hidden] -> 32 channels: block cb, chunks 0, 1 = W_shortcut columns, chunk 2 = W_1x1 columns 0 .. 15
    float w0r[2][2];   // conv0: block cb, k step st: w0[4 st + kq][16 cb + li] (k = 7: zero)
    f32x4v b0r[2], b3r, bcr[2];  // biases of this lane's 4 output channels (4 kq .. + 3 of each block)
#pragma unroll
    for (int s = 0; s < 6; ++s) w3r[s] = *reinterpret_cast<const f32x4v*>(p.w3 + (long long)li * (3 * C) + 16 * s + 4 * kq);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
        for (int s = 0; s < 2; ++s) wcr[cb][s] = *reinterpret_cast<const f32x4v*>(p.wsc + (long long)(16 * cb + li) * C + 16 * s + 4 * kq);
        wcr[cb][2] = *reinterpret_cast<const f32x4v*>(p.wpw + (long long)(16 * cb + li) * 32 + 4 * kq);
#pragma unroll
        for (int st = 0; st < 2; ++st) w0r[cb][st] = (4 * st + kq) < 7 ? p.w0[(4 * st + kq) * C + 16 * cb + li] : 0.f;
        b0r[cb] = *reinterpret_cast<const f32x4v*>(p.b0 + 16 * cb + 4 * kq);
        bcr[cb] = *reinterpret_cast<const f32x4v*>(p.bsc + 16 * cb + 4 * kq) + *reinterpret_cast<const f32x4v*>(p.bpw + 16 * cb + 4 * kq);
    }
    b3r = *reinterpret_cast<const f32x4v*>(p.b3 + 4 * kq);

    const int L = p.L;
    const int tiles_per_clip = (L + ROWS - 1) / ROWS, n_tiles = p.B * tiles_per_clip;
    float* HSw = HS + wave * 32 * HLD;
    // sample u of a tile's window is wav[reflect(r0 - pl3 - pl0 + u)]; one sample per thread, fetched one tile ahead
    auto wav_fetch = [&](int tile_) -> float {
        if (tile_ >= n_tiles || tid >= WSN) return 0.f;
        const int b_ = tile_ / tiles_per_clip, r0_ = (tile_ - b_ * tiles_per_clip) * ROWS;
        const int s_ = resolve_frame(r0_ - p.pl3 - p.pl0 + tid, L, p.Lp0, PAD_REFLECT);
        return s_ >= 0 ? p.wav[(long long)b_ * L + s_] : 0.f;
    };
    if (tid < WSP) {
        WS[tid] = tid < WSN ? wav_fetch(blockIdx.x) : 0.f;
        WS[WSP + tid] = 0.f;
    }
    int cur = 0;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, cur ^= 1) {
        const int b = tile / tiles_per_clip, r0 = (tile - b * tiles_per_clip) * ROWS;
        __syncthreads();  // the previous tile is no longer read (first trip: orders the window stores too)
        const float wnext = wav_fetch(tile + gridDim.x);  // in flight under this whole tile
        const float* ws = WS + cur * WSP;
        // ---- x tile: rows i = 0 .. XR - 1 are frames q = r0 - pl3 + i with the k3's reflect padding resolved per row: x[src(q)] needs the
        // conv0-padded signal at src - pl0 + k = window sample (src - r0 + pl3) + k.  Frame blocks of 16 rows dealt to the waves
        // (9 blocks cover 144 rows; rows >= XR are not stored).
        for (int fb = wave; fb < 9; fb += 4) {
            const int i = 16 * fb + li;
            const int q = r0 - p.pl3 + i;
            const int src = resolve_frame(q, L, p.Lp3, PAD_REFLECT);
            const int u0 = max(0, min(src - r0 + p.pl3, WSP - 8)) + kq;  // rows whose window falls outside belong to frames past the clip: never stored
            const float s0 = ws[u0], s1 = ws[u0 + 4];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                f32x4v v = b0r[cb];  // C-in = bias, then k = 0 .. 7 in order: conv_in_kernel's fma chain
                v = __builtin_amdgcn_mfma_f32_16x16x4f32(w0r[cb][0], s0, v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x4f32(w0r[cb][1], s1, v, 0, 0, 0);
                if (src < 0) v = (f32x4v){0.f, 0.f, 0.f, 0.f};  // zero extension of a short clip (not reachable for L >= 8; kept for the contract)
                if (i < XR) {
                    *reinterpret_cast<f32x4v*>(XS + i * LDX + 16 * cb + 4 * kq) = v;
                    *reinterpret_cast<f32x4v*>(XE + i * LDX + 16 * cb + 4 * kq) = elu4(v);
                }
            }
        }
        __syncthreads();

        // ---- hidden = ELU(k3(ELU(x)) + b3): K = 3 taps x 32 channels = 6 chunks of 16; taps are row offsets 0, 1, 2 of the tile
        f32x4v h[2];
#pragma unroll
        for (int fbk = 0; fbk < 2; ++fbk) {
            h[fbk] = b3r;
            const float* arow = XE + (wave * 32 + 16 * fbk + li) * LDX + 4 * kq;
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const f32x4v av = *reinterpret_cast<const f32x4v*>(arow + (s >> 1) * LDX + 16 * (s & 1));
                h[fbk] = __builtin_amdgcn_mfma_f32_16x16x4f32(w3r[s].x, av.x, h[fbk], 0, 0, 0);
                h[fbk] = __builtin_amdgcn_mfma_f32_16x16x4f32(w3r[s].y, av.y, h[fbk], 0, 0, 0);
                h[fbk] = __builtin_amdgcn_mfma_f32_16x16x4f32(w3r[s].z, av.z, h[fbk], 0, 0, 0);
                h[fbk] = __builtin_amdgcn_mfma_f32_16x16x4f32(w3r[s].w, av.w, h[fbk], 0, 0, 0);
            }
        }
        // hidden tile: written and read by this wave only (LDS operations of a wave complete in order; the compiler's lgkmcnt waits cover it)
#pragma unroll
        for (int fbk = 0; fbk < 2; ++fbk) *reinterpret_cast<f32x4v*>(HSw + (16 * fbk + li) * HLD + 4 * kq) = elu4(h[fbk]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- a = ELU([x | hidden] [W_sc | W_1x1]^T + (b_sc + b_1x1))
        f32x4v o[2][2];
#pragma unroll
        for (int fbk = 0; fbk < 2; ++fbk) {
            const float* xrow = XS + (wave * 32 + 16 * fbk + li + p.pl3) * LDX + 4 * kq;
            const float* hrow = HSw + (16 * fbk + li) * HLD + 4 * kq;
            const f32x4v a0 = *reinterpret_cast<const f32x4v*>(xrow), a1 = *reinterpret_cast<const f32x4v*>(xrow + 16);
            const f32x4v a2 = *reinterpret_cast<const f32x4v*>(hrow);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                f32x4v v = bcr[cb];
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const f32x4v av = s == 0 ? a0 : (s == 1 ? a1 : a2);
                    v = __builtin_amdgcn_mfma_f32_16x16x4f32(wcr[cb][s].x, av.x, v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_16x16x4f32(wcr[cb][s].y, av.y, v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_16x16x4f32(wcr[cb][s].z, av.z, v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_16x16x4f32(wcr[cb][s].w, av.w, v, 0, 0, 0);
                }
                o[fbk][cb] = v;
            }
        }
        // output: a lane holds 4 consecutive channels of a frame per (frame block, channel block); the four lanes kq = 0 .. 3 of a frame write
        // 64 contiguous bytes, the two channel blocks complete the frame's 128-byte line - straight from registers, no staging, no barrier
#pragma unroll
        for (int fbk = 0; fbk < 2; ++fbk) {
            const int t = r0 + wave * 32 + 16 * fbk + li;
            if (t < L) {
                float* dst = p.a + ((long long)b * L + t) * C + 4 * kq;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) *reinterpret_cast<f32x4v*>(dst + 16 * cb) = elu4(o[fbk][cb]);
            }
        }
        if (tid < WSN) WS[(cur ^ 1) * WSP + tid] = wnext;  // that buffer was last read before this tile's second barrier (the conv0 phase)
    }
}

static size_t seanet_block_lds_bytes(int C) {
    const int XR = 130, LDX = C + 4, WSP = 9 * 16 + 8, HLD = 20;
    return sizeof(float) * ((size_t)2 * XR * LDX + 4 * 32 * HLD + 2 * WSP);
}

bool seanet_front_supported(int C, int hid, int L) {
    const bool on = knob(K_SEANET_FUSED) != 0;
    // L >= 8: both reflect pads stay in their plain regime (no zero extension of a short clip)
    return on && C == 32 && hid >= 1 && hid <= 16 && L >= 8;  // hid <= 16: ONE 16-channel MFMA row block holds the hidden tile
}

// k3 / pw in the padded library layouts of hcodec.cpp (k3 [32][3][C], pw [C][32]).
int launch_seanet_front(const float* wav, const float* w0, const float* b0, const float* w3, const float* b3, const float* wsc,
                        const float* bsc, const float* wpw, const float* bpw, float* a, int B, int L, int C, int hid, int causal,
                        hipStream_t s) {
    QA_REQUIRE(seanet_front_supported(C, hid, L) && wav && w0 && b0 && w3 && b3 && wsc && bsc && wpw && bpw && a, "seanet_front: unsupported "
               "shape C=%d hid=%d L=%d", C, hid, L);
    SeanetFrontParams p{};
    p.wav = wav; p.w0 = w0; p.b0 = b0; p.w3 = w3; p.b3 = b3; p.wsc = wsc; p.bsc = bsc; p.wpw = wpw; p.bpw = bpw; p.a = a;
    p.B = B; p.L = L;
    p.pl3 = causal ? 2 : 1;
    p.pl0 = causal ? 6 : 3;
    // the x-tile rows read the staged waveform window at clamp(src - r0 + pl3): every reflected halo row a tile needs lies inside that
    // window only for a left pad of 1 or 2 frames - any other layout must fail here, not read clamped samples
    QA_REQUIRE(p.pl3 >= 1 && p.pl3 <= 2, "seanet_front: k3 left pad %d outside the staged window's reach", p.pl3);
    const int mp3 = causal ? 2 : 1, mp0 = causal ? 6 : 3;
    p.Lp3 = L <= mp3 ? mp3 + 1 : L;
    p.Lp0 = L <= mp0 ? mp0 + 1 : L;
    const size_t lds = seanet_block_lds_bytes(C);
    const long long n_tiles = (long long)B * ceil_div(L, 128);
    const unsigned grid = (unsigned)std::min<long long>(n_tiles, 768);  // persistent: three workgroups per CU (49 KB of LDS each), weights loaded once each
    QA_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(seanet_block_kernel<32>), (int)lds));
    HbmProf prof_(HK_SEANET_FRONT, 4.0 * ((double)B * L + (double)B * L * C), s);  // wav in, the block output `a` written once
    hipLaunchKernelGGL((seanet_block_kernel<32>), dim3(grid), dim3(256), lds, s, p);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

}  // namespace qa
