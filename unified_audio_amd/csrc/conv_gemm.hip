// conv_gemm.hip - implicit-GEMM Conv1d / Linear over channel-last activations on the fp32 matrix cores.
//
// Replaces every Conv1d / Linear of the reference's codec and LM graphs (SURVEY.md 2.2 K1,K2,K5-K7,K10-K13,K17):
//   y[m, n] = post( res[m,n] + gamma[n] * act( silu(gate[m,n]) * (bias[n] + sum_k A[m,k] * W[n,k]) ) )
// where row m = (b, t) of a [B, T_out] grid and A[m, (j, c)] = pro(x[b, src(t, j), c]) is gathered on the fly:
// in channel-last layout the im2col row of a 1-D convolution is `ksize` contiguous C_in-long segments, so no
// im2col buffer and no padded copy ever exist in HBM (reflect / zero padding are resolved per segment).
//
// gfx950 mapping: 256-thread workgroups (4 wave64), block tile BM x BN x 32, each wave owns a grid of 32x32
// accumulators fed by v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 157 TFLOP/s peak).  Operands are staged
// global -> registers -> LDS (double buffered, one barrier per K step, next tile's global loads issued before
// the current tile's MFMAs).  LDS rows are padded to 36 floats so that the ds_read_b128 fragment loads are
// bank-conflict free; each lane fetches 4 consecutive k per read and the (lane>>5) halves take k-groups
// {0..3},{4..7}: the K order inside a step is permuted identically for A and B, which leaves the sum unchanged.
#include <cstdlib>

#include "common.h"

namespace qa {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BM, int BN, int WM, int WN, bool PRO_ELU, int BK = 32>
__global__ __launch_bounds__(256, (BK == 16 ? 3 : 2)) void conv_gemm_kernel(const ConvParams p) {
    constexpr int LDS = BK + 4;
    constexpr int RPP = 256 / (BK / 4);  // rows staged per pass: 8 (BK=32) or 4 (BK=16) threads cover one row chunk
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int A_IT = BM / RPP, B_IT = BN / RPP;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");

    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDS];
    float* sA = smem;
    float* sB = smem + 2 * BM * LDS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (p.N + BN - 1) / BN;
    // XCD-aware tile order: workgroup b is dispatched to XCD b % 8 (observed, used for speed only).  Give every XCD a
    // contiguous run of tile ids (n fastest inside it) so the A rows an XCD works on stay private to its L2 and the
    // column tiles of W are re-read from that same L2.  Bijective for any grid size.
    int tile = blockIdx.x;
    if (p.xcd_swizzle) {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = tile & 7, local = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int m0 = (tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * BN;

    const int ld_row = tid / (BK / 4);          // 0..RPP-1
    const int ld_c4 = (tid % (BK / 4)) * 4;     // float offset inside the BK-wide K chunk

    // Per-thread A rows: batch base offset and first source frame.  Rows past M are clamped to row M-1 and columns past
    // N to column N-1 (their results are never stored), so every load below is unconditional: hipcc would otherwise
    // branch around each predicated load and drain vmcnt per element.
    long long a_base[A_IT];
    int a_t0[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = min(m0 + ld_row + RPP * i, p.M - 1);
        const int b = m / p.T_out;
        const int t = m - b * p.T_out;
        a_base[i] = (long long)b * p.T_in * p.ldx + ld_c4;
        a_t0[i] = t * p.stride - p.pad_left;
    }
    const float* b_ptr[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int n = min(n0 + ld_row + RPP * i, p.N - 1);
        b_ptr[i] = p.w + (long long)n * p.K + ld_c4;
    }

    const bool reflect = p.pad_mode == PAD_REFLECT;
    const int ldx_i = (int)p.ldx;
    const int nk = p.K / BK;
    const int in_rep = p.in_rep;
    const int t_virtual = p.T_in * (in_rep > 1 ? in_rep : 1);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Staging registers of the NEXT K chunk.  No lambdas / conditionals around them: every iteration loads (the last one
    // re-loads chunk nk-1, harmlessly) so that hipcc keeps them in VGPRs and issues the loads before the MFMAs.  (Keeping
    // the row pointers incrementally under a wave-uniform "tap changed" branch was measured 13 % SLOWER: the branch splits
    // the block and the address VALU no longer interleaves with the MFMAs.)
    float4 a_reg[A_IT], b_reg[B_IT];
    float a_keep[A_IT];  // 0 for frames that fall into zero padding (select on the data at LDS-store time, not on the load)

#define QA_LOAD_GLOBAL(KC)                                                                                     \
    {                                                                                                          \
        const int k0_ = (KC) * BK;                                                                             \
        const int j_ = k0_ / p.C_in;                                                                           \
        const int c_ = k0_ - j_ * p.C_in;                                                                      \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                                     \
            int r_ = a_t0[i] + j_;                                                                             \
            const int rr_ = r_ < 0 ? -r_ : (r_ >= p.Lp ? 2 * (p.Lp - 1) - r_ : r_); /* = resolve_frame() */    \
            r_ = reflect ? rr_ : r_;                                                                           \
            const bool ok_ = r_ >= 0 && r_ < t_virtual;                                                        \
            a_keep[i] = ok_ ? 1.f : 0.f;                                                                       \
            r_ = in_rep > 1 ? r_ / in_rep : r_;                                                                \
            a_reg[i] = *reinterpret_cast<const float4*>(p.x + a_base[i] + (unsigned)((ok_ ? r_ : 0) * ldx_i + c_)); \
        }                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i) b_reg[i] = *reinterpret_cast<const float4*>(b_ptr[i] + k0_); \
    }
#define QA_STORE_LDS(BUF)                                                                                      \
    {                                                                                                          \
        float* a_ = sA + (BUF) * BM * LDS;                                                                     \
        float* b_ = sB + (BUF) * BN * LDS;                                                                     \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                                     \
            float4 v = a_reg[i];                                                                               \
            v.x *= a_keep[i]; v.y *= a_keep[i]; v.z *= a_keep[i]; v.w *= a_keep[i];                            \
            if (PRO_ELU) {                                                                                     \
                v.x = elu_f(v.x); v.y = elu_f(v.y); v.z = elu_f(v.z); v.w = elu_f(v.w);                        \
            }                                                                                                  \
            *reinterpret_cast<float4*>(a_ + (ld_row + RPP * i) * LDS + ld_c4) = v;                             \
        }                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i)                                                       \
            *reinterpret_cast<float4*>(b_ + (ld_row + RPP * i) * LDS + ld_c4) = b_reg[i];                      \
    }

    QA_LOAD_GLOBAL(0)
    QA_STORE_LDS(0)
    __syncthreads();

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 4;
    for (int kc = 0; kc < nk; ++kc) {
        const int cur = kc & 1;
        const int nxt = min(kc + 1, nk - 1);
        QA_LOAD_GLOBAL(nxt)
        const float* a = sA + cur * BM * LDS + (wm * WTM + frag_row) * LDS + frag_k;
        const float* b = sB + cur * BN * LDS + (wn * WTN + frag_row) * LDS + frag_k;
        {
            // fragment double buffering: the ds_read_b128 of k-group kk+1 are issued before the MFMAs of group kk
            float4 af[2][TM], bf[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const float4*>(a + i * 32 * LDS);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const float4*>(b + j * 32 * LDS);
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                const int cb = kk & 1, nb = cb ^ 1;
                if (kk < BK / 8 - 1) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        af[nb][i] = *reinterpret_cast<const float4*>(a + i * 32 * LDS + (kk + 1) * 8);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        bf[nb][j] = *reinterpret_cast<const float4*>(b + j * 32 * LDS + (kk + 1) * 8);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cb][i].x, bf[cb][j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cb][i].y, bf[cb][j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cb][i].z, bf[cb][j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cb][i].w, bf[cb][j].w, acc[i][j], 0, 0, 0);
                    }
            }
        }
        QA_STORE_LDS(cur ^ 1)
        __syncthreads();
    }
#undef QA_LOAD_GLOBAL
#undef QA_STORE_LDS

    // Epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
    const int col_l = lane & 31, row_h = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WTN + j * 32 + col_l;
            if (n >= p.N) continue;
            const float bias = p.bias ? p.bias[n] : 0.f;
            const float gamma = p.gamma ? p.gamma[n] : 1.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bias;
                if (p.gate) v = silu_f(p.gate[(long long)m * p.ldg + n]) * v;
                v = apply_act(v, p.act);
                if (p.gamma) v *= gamma;
                if (p.res) v += p.res[(long long)m * p.ldr + n];
                v = apply_act(v, p.post_act);
                p.y[(long long)m * p.ldy + n] = v;
            }
        }
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg(const ConvParams& p, hipStream_t stream) {
    QA_REQUIRE(p.prologue == ACT_NONE || p.prologue == ACT_ELU, "conv_gemm: prologue %d unsupported", p.prologue);
    const long long tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
    const bool prof = profile_enabled();
    if (prof) {
        const int cfg = BN == 32 ? PROF_CFG_128x32 : (BN == 64 ? PROF_CFG_128x64 : PROF_CFG_128x128);
        const double n = p.algo_n ? p.algo_n : p.N, k = p.algo_k ? p.algo_k : p.K;
        profile_record_begin(cfg, 2.0 * (double)p.M * n * k, stream);
    }
    // Short-K layers (K <= 512: 16 chunks or fewer per tile) are dominated by per-tile fixed costs; the BK = 16 variant needs
    // 41 KB / 31 KB of LDS, so 3-4 workgroups are co-resident per CU and cover each other's prologue / epilogue
    // (measured on M = 9056, K = 512: +10 % for N = 1536, +25 % for N = 2048; neutral from K = 1024 up).
    static const int bk16_max_k = [] {
        const char* e = getenv("QA_GEMM_BK16");
        return e ? atoi(e) : 512;
    }();
    if (BN >= 64 && p.prologue != ACT_ELU && p.K <= bk16_max_k)
        hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, false, (BN >= 64 ? 16 : 32)>), dim3((unsigned)tiles), dim3(256), 0, stream, p);
    else if (p.prologue == ACT_ELU)
        hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, true>), dim3((unsigned)tiles), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, false>), dim3((unsigned)tiles), dim3(256), 0, stream, p);
    if (prof) profile_record_end(stream);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

int launch_conv_gemm(const ConvParams& p, hipStream_t stream) {
    QA_REQUIRE(p.K % 32 == 0 && p.C_in % 32 == 0, "conv_gemm: K=%d / C_in=%d must be multiples of 32", p.K, p.C_in);
    QA_REQUIRE((p.ldx % 4) == 0, "conv_gemm: ldx=%lld must be a multiple of 4 floats", p.ldx);
    QA_REQUIRE((long long)p.T_in * p.ldx < (1LL << 31), "conv_gemm: one batch item spans %lld floats (limit 2^31)",
               (long long)p.T_in * p.ldx);
    QA_REQUIRE(((uintptr_t)p.x % 16) == 0 && ((uintptr_t)p.w % 16) == 0, "conv_gemm: x / w must be 16-byte aligned");
    if (p.M <= 0 || p.N <= 0) return QA_OK;
    QA_REQUIRE(ceil_div(p.M, 64) * ceil_div(p.N, 32) < (1LL << 31), "conv_gemm: grid too large");
    static const int forced = [] {
        const char* e = getenv("QA_GEMM_CFG");
        return e ? atoi(e) : -1;
    }();
    static const int swz = [] {
        const char* e = getenv("QA_GEMM_XCD");
        return e ? atoi(e) : 1;
    }();
    ConvParams q = p;
    q.xcd_swizzle = swz;
    QA_REQUIRE(p.in_rep <= 1 || p.pad_mode == PAD_ZERO, "conv_gemm: in_rep needs zero padding");
    int cfg;
    if (forced >= 0) cfg = forced;
    else if (p.N <= 32) cfg = PROF_CFG_128x32;
    else if (p.N <= 64) cfg = PROF_CFG_128x64;
    else {
        // Tiles of one launch are dealt round-robin over 256 CUs (two co-resident workgroups share a CU's matrix pipes), so
        // the makespan is (tiles on the busiest CU) x (work per tile) / (sustained efficiency of the configuration:
        // measured ~92 TFLOP/s for 128x128 vs ~75 for 128x64 on long-K shapes).
        const long long t128 = ceil_div(p.M, 128) * ceil_div(p.N, 128), t64 = ceil_div(p.M, 128) * ceil_div(p.N, 64);
        const double c128 = (double)ceil_div(t128, 256) * 128 * 128 / 1.0;
        const double c64 = (double)ceil_div(t64, 256) * 128 * 64 / 0.82;
        cfg = c64 < c128 ? PROF_CFG_128x64 : PROF_CFG_128x128;
    }
    switch (cfg) {
        case PROF_CFG_128x32: return launch_cfg<128, 32, 4, 1>(q, stream);
        case PROF_CFG_128x64: return launch_cfg<128, 64, 2, 2>(q, stream);
        default: return launch_cfg<128, 128, 2, 2>(q, stream);
    }
}

int conv_params_from_args(const qa_conv_args& a, ConvParams* out) {
    ConvParams p{};
    QA_REQUIRE(a.x && a.w && a.y, "conv1d_cl: x, w, y must be non-null");
    QA_REQUIRE(a.B > 0 && a.T_in > 0 && a.C_in > 0 && a.N > 0 && a.T_out >= 0, "conv1d_cl: bad shape");
    QA_REQUIRE(a.ksize >= 1 && a.stride >= 1 && a.pad_left >= 0 && a.pad_right >= 0, "conv1d_cl: bad geometry");
    QA_REQUIRE(a.B * a.T_out < (1LL << 31) && a.ksize * a.C_in < (1LL << 31), "conv1d_cl: shape too large");
    p.x = a.x; p.w = a.w; p.bias = a.bias; p.gamma = a.gamma; p.res = a.residual; p.gate = a.gate; p.y = a.y;
    p.ldx = a.ldx ? a.ldx : a.C_in;
    p.ldy = a.ldy ? a.ldy : a.N;
    p.ldr = a.ldr ? a.ldr : a.N;
    p.ldg = a.ldg ? a.ldg : a.N;
    p.B = (int)a.B; p.T_in = (int)a.T_in; p.C_in = (int)a.C_in; p.T_out = (int)a.T_out; p.N = (int)a.N;
    p.K = (int)(a.ksize * a.C_in);
    p.M = (int)(a.B * a.T_out);
    p.ksize = a.ksize; p.stride = a.stride; p.pad_left = a.pad_left; p.pad_mode = a.pad_mode;
    const int max_pad = a.pad_left > a.pad_right ? a.pad_left : a.pad_right;
    p.Lp = (p.T_in <= max_pad) ? max_pad + 1 : p.T_in;
    // every window must stay inside the padded signal
    p.in_rep = a.in_rep > 1 ? a.in_rep : 1;
    QA_REQUIRE(a.T_out == 0 || (a.T_out - 1) * (int64_t)a.stride + a.ksize <= a.pad_left + a.T_in * (int64_t)p.in_rep + a.pad_right,
               "conv1d_cl: T_out=%lld windows do not fit pad_left=%d + T_in=%lld + pad_right=%d", (long long)a.T_out,
               a.pad_left, (long long)a.T_in, a.pad_right);
    p.prologue = a.prologue; p.act = a.act; p.post_act = a.post_act;
    *out = p;
    return QA_OK;
}

}  // namespace qa
