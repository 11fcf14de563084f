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

// tuning knobs of tools/variants.py
#ifndef QA_LOAD_AT  // MFMA group of a K chunk that carries the next chunk's global loads
#define QA_LOAD_AT 0
#endif

#ifdef QA_TIMING  // tuning builds only (tools/variants.py): per-phase shader-cycle totals of the main loop, summed over waves
__device__ unsigned long long g_qa_timing[10];
#define QA_TICK(i)                                              \
    {                                                           \
        const long long now_ = __builtin_readcyclecounter();    \
        tacc[i] += now_ - tlast;                                \
        tlast = now_;                                           \
    }
extern "C" int qa_debug_timing(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qa_timing), sizeof(g_qa_timing)) != hipSuccess) return -1;  // out[10]
    if (reset) {
        unsigned long long z[10] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_qa_timing), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define QA_TICK(i)
#endif

namespace qa {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: stays in VGPRs (float4 arrays were left as scratch allocas)

constexpr int TAP_WIN = 8;  // taps per LDS source-frame table window (ksize <= 8: built once per tile)

__device__ __forceinline__ float snake_f(float x, float a) {
    const float s = sinf(a * x);
    return x + s * s / (a + 1e-9f);
}
__device__ __forceinline__ f32x4 snake4(f32x4 v, const float* alpha, int n) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(alpha + n);
    const f32x4 r = {snake_f(v.x, a.x), snake_f(v.y, a.y), snake_f(v.z, a.z), snake_f(v.w, a.w)};
    return r;
}

// epilogue of 4 consecutive output channels of one row: v = acc + bias -> gate -> act -> gamma -> residual -> post_act
__device__ __forceinline__ f32x4 epilogue4(f32x4 v, const ConvParams& p, long long m, int n, f32x4 bias, f32x4 gamma) {
    v += bias;
    if (p.gate) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(p.gate + m * p.ldg + n);
        v.x *= silu_f(g.x); v.y *= silu_f(g.y); v.z *= silu_f(g.z); v.w *= silu_f(g.w);
    }
    if (p.act == ACT_SNAKE) {
        v = snake4(v, p.alpha, n);
    } else if (p.act != ACT_NONE) {
        v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act); v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act);
    }
    if (p.gamma) v *= gamma;
    if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + m * p.ldr + n);
    if (p.post_act == ACT_SNAKE) {
        v = snake4(v, p.alpha, n);
    } else if (p.post_act != ACT_NONE) {
        v.x = apply_act(v.x, p.post_act); v.y = apply_act(v.y, p.post_act); v.z = apply_act(v.z, p.post_act);
        v.w = apply_act(v.w, p.post_act);
    }
    if (p.y2) *reinterpret_cast<f32x4*>(p.y2 + m * p.ldy2 + n) = snake4(v, p.alpha2, n);
    if (p.rope && n < p.rope_n) {  // interleaved pairs (2i, 2i+1): both members of a pair sit in this float4 (n % 4 == 0)
        const int t = p.rope_pos0 + (int)(m % p.rope_T), i = (n % p.rope_hd) >> 1;
        const f32x4 cs = *reinterpret_cast<const f32x4*>(p.rope + ((long long)t * (p.rope_hd >> 1) + i) * 2);  // c_i, s_i, c_i+1, s_i+1
        const f32x4 r = {v.x * cs.x - v.y * cs.y, v.y * cs.x + v.x * cs.y, v.z * cs.z - v.w * cs.w, v.w * cs.z + v.z * cs.w};
        v = r;
    }
    return v;
}

// LINEAR: ksize 1, stride 1, no padding, no input repetition (every Linear / 1x1 of the graphs: most of the FLOPs).  Row m of A is
// x + m * ldx, so the per-chunk source-frame lookup (LDS read, clamp, select, 64-bit address build) and the padding multiply
// disappear from the main loop - about 20 of its ~50 non-MFMA instructions, each of which costs ~30 cycles beside the co-resident
// workgroup's MFMAs.
template <int BM, int BN, int WM, int WN, bool PRO_ELU, int BK = 32, bool LINEAR = false>
__global__ __launch_bounds__(256, (BM == 64 && (BK == 16 || BN == 64) ? 4 : (BK == 16 ? 3 : 2))) void conv_gemm_kernel(const ConvParams p_in) {
    ConvParams p = p_in;
    constexpr int LDS = BK + 4;
    constexpr int RPP = 256 / (BK / 4);  // rows staged per pass: 8 (BK=32) or 4 (BK=16) threads cover one row chunk
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int A_IT = BM / RPP, B_IT = BN / RPP;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");

    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDS];
    __shared__ int s_tap[BM * TAP_WIN];  // source-frame offset (floats, relative to the clip) of (row, tap); -1 = zero padding
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
    int tm_i = tile / tiles_n, tn_i = tile % tiles_n;
    if (p.panel > 0 && tiles_n > p.panel) {
        // 2-D blocking of the tile order (QA_GEMM_PANEL = PW): column panels of PW tiles, row tiles fastest inside a panel, so that the
        // ~96 tiles an XCD has in flight form a ~(96 / PW) x PW block - the bytes they pull through the XCD's L2 scale with
        // rows + columns of the block instead of with one of them
        const int tiles_m = (p.M + BM - 1) / BM;
        const int per_panel = tiles_m * p.panel;
        const int panel = tile / per_panel, r = tile - panel * per_panel;
        const int pw = min(p.panel, tiles_n - panel * p.panel);
        tm_i = r / pw;
        tn_i = panel * p.panel + (r - tm_i * pw);
    }
    const int m0 = tm_i * BM;
    const int n0 = tn_i * BN;

    const int ld_row = tid / (BK / 4);          // 0..RPP-1
    const int ld_c4 = (tid % (BK / 4)) * 4;     // float offset inside the BK-wide K chunk

    // Source-frame table.  The im2col row of output frame t is the ksize frames src(t, j); padding (reflect with the
    // short-input rule of pad1d, or zeros) and repeat_interleave are resolved HERE, once per (row, tap), so that the main
    // loop's address arithmetic is one LDS read per staged row: vector ALU instructions issued beside the other wave's
    // MFMAs cost ~30 cycles each (measured), and the per-chunk resolve used to take as long as the MFMA phase itself.
    // Rows past M are clamped to row M-1 and columns past N to column N-1 (their results are never stored), so every
    // load in the main loop is unconditional.
    const bool reflect = p.pad_mode == PAD_REFLECT;
    const int ldx_i = (int)p.ldx;
    const int t_virtual = p.T_in * (p.in_rep > 1 ? p.in_rep : 1);
    const int dil = p.dilation > 1 ? p.dilation : 1;
    auto build_taps = [&](int jbase) {
        for (int e = tid; e < BM * TAP_WIN; e += 256) {
            const int row = e / TAP_WIN, j = jbase + (e % TAP_WIN);
            const int m = min(m0 + row, p.M - 1);
            const int t = m % p.T_out;
            int r = t * p.stride - p.pad_left + j * dil;
            const int rr = r < 0 ? -r : (r >= p.Lp ? 2 * (p.Lp - 1) - r : r);  // = resolve_frame()
            r = reflect ? rr : r;
            const bool ok = r >= 0 && r < t_virtual && j < p.ksize;
            const unsigned ru = ok ? (unsigned)r : 0u;
            const unsigned src = __umulhi(ru, p.rep_magic) + ru * p.rep_one;  // = r / in_rep
            s_tap[e] = ok ? (int)(src * (unsigned)ldx_i) : -1;
        }
    };
    if (!LINEAR) build_taps(0);
    int jbase = 0;

    const float* a_ptr[A_IT];  // clip base + this thread's column offset inside a chunk
    int a_tab[A_IT];           // byte offset of the row's table line
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = min(m0 + ld_row + RPP * i, p.M - 1);
        a_ptr[i] = LINEAR ? p.x + (long long)m * p.ldx + ld_c4 : p.x + (long long)(m / p.T_out) * p.T_in * p.ldx + ld_c4;
        a_tab[i] = (ld_row + RPP * i) * TAP_WIN;
    }
    const float* b_ptr[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int n = min(n0 + ld_row + RPP * i, p.N - 1);
        b_ptr[i] = p.w + (long long)n * p.K + ld_c4;
    }
    const int nk = p.K / BK;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Staging registers of the NEXT K chunk (native vector type: float4 arrays were left as scratch allocas).  Every
    // iteration loads (the last one re-loads chunk nk-1, harmlessly).
    f32x4 a_reg[A_IT], b_reg[B_IT];
    float a_keep[A_IT];  // 0 for frames that fall into zero padding (select on the data at LDS-store time, not on the load)

#define QA_LOAD_GLOBAL(KC)                                                                                     \
    {                                                                                                          \
        const int k0_ = (KC) * BK;                                                                             \
        if (LINEAR) {                                                                                          \
            _Pragma("unroll") for (int i = 0; i < A_IT; ++i) a_reg[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + k0_); \
        } else {                                                                                               \
            const int j_ = k0_ / p.C_in;                                                                       \
            const int c_ = k0_ - j_ * p.C_in;                                                                  \
            if (j_ >= jbase + TAP_WIN) { /* block-uniform; only for ksize > TAP_WIN */                         \
                jbase = j_;                                                                                    \
                build_taps(jbase);                                                                             \
                __syncthreads();                                                                               \
            }                                                                                                  \
            _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                                 \
                const int off_ = s_tap[a_tab[i] + (j_ - jbase)];                                               \
                a_keep[i] = off_ >= 0 ? 1.f : 0.f;                                                             \
                a_reg[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + (unsigned)(max(off_, 0) + c_));          \
            }                                                                                                  \
        }                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i) b_reg[i] = *reinterpret_cast<const f32x4*>(b_ptr[i] + k0_); \
    }
#define QA_STORE_LDS(BUF)                                                                                      \
    {                                                                                                          \
        float* a_ = sA + (BUF) * BM * LDS;                                                                     \
        float* b_ = sB + (BUF) * BN * LDS;                                                                     \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                                     \
            f32x4 v = LINEAR ? a_reg[i] : a_reg[i] * a_keep[i];                                                \
            if (PRO_ELU) {                                                                                     \
                v.x = elu_f(v.x); v.y = elu_f(v.y); v.z = elu_f(v.z); v.w = elu_f(v.w);                        \
            }                                                                                                  \
            *reinterpret_cast<f32x4*>(a_ + (ld_row + RPP * i) * LDS + ld_c4) = v;                              \
        }                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i)                                                       \
            *reinterpret_cast<f32x4*>(b_ + (ld_row + RPP * i) * LDS + ld_c4) = b_reg[i];                       \
    }

    __syncthreads();  // tap table visible
    QA_LOAD_GLOBAL(0)
    QA_STORE_LDS(0)
    __syncthreads();

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 4;
#ifdef QA_TIMING
    long long tacc[5] = {0, 0, 0, 0, 0};
    long long tlast = __builtin_readcyclecounter();
    const long long tbegin = tlast;
    const unsigned long long rbegin = __builtin_amdgcn_s_memrealtime();  // constant 100 MHz
#endif
    // One K chunk = BK/8 groups of 4*TM*TN MFMAs.  The next chunk's address arithmetic + global loads ride in group 0 and
    // its LDS stores in the last group, pinned there by sched_barriers: a wave's non-MFMA instructions must sit BETWEEN ITS
    // OWN MFMAs.  Bunched before / after the MFMA phase they have to issue while the co-resident workgroup's wave owns the
    // SIMD with back-to-back 64-cycle MFMAs, and then get roughly one issue slot per MFMA (measured: ~55 cycles per
    // instruction, the "address phase" lasted as long as the whole MFMA phase).
    constexpr int NKK = BK / 8;
    // One barrier per chunk, placed right behind the stores of the NEXT chunk's tile and in front of the LAST MFMA group of this one (r06): a wave leaves the
    // barrier with that group's operands already in registers, issues the next chunk's first fragment reads and covers their LDS latency with 4 TM TN MFMAs.
    // At the chunk's END (rounds 1 - 5) the barrier was followed by reads -> wait -> first MFMA: a bubble per chunk and wave that the co-resident waves did
    // not always fill - 131.1 -> 135.4 TFLOP/s at 16 000 x 3 072 x 1 024, H-Codec 1.5 129.8 -> 128.0 ms, 2.0 470.2 -> 461.8 ms, bit-identical
    // (profiles/r06_gemm_early_barrier_ab.txt).  Safe with two buffers: a wave's reads of buffer `cur` all precede its arrival at this chunk's barrier, and
    // nobody writes `cur` before passing it.  Operands swapped on purpose: the W fragment is the MFMA's row operand and the activation fragment its column
    // operand, so D = (A W^T)^T and every lane ends up with 4 CONSECUTIVE output channels of one output row per register quad -> the epilogue moves float4.
    static_assert(NKK % 2 == 0, "the fragment buffers alternate: the next chunk's first group lands in slot 0");
    // Where the NEXT tile's global loads are issued.  64 x 64 tiles (the N = 512 layers of the aggregator stacks): chunk kc + 2's loads right behind chunk kc's
    // barrier - the staging registers are free again there, and the loads get a whole chunk of MFMAs to arrive (+5 ... 7 % on 9 056 x 512 x 512 / 2 048).  Every
    // larger tile: chunk kc + 1's loads in MFMA group QA_LOAD_AT of chunk kc - behind the barrier they cost the 128-row tiles 4 % (all waves of a workgroup
    // issue them in one burst together with the fragment reads; profiles/r06_gemm_early_barrier_ab.txt).
    constexpr bool LOAD_AHEAD = BM == 64 && BN == 64;
    if (LOAD_AHEAD) QA_LOAD_GLOBAL(min(1, nk - 1))
    f32x4 af[2][TM], bf[2][TN];
    {
        const float* a0 = sA + (wm * WTM + frag_row) * LDS + frag_k;
        const float* b0 = sB + (wn * WTN + frag_row) * LDS + frag_k;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const f32x4*>(a0 + i * 32 * LDS);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const f32x4*>(b0 + j * 32 * LDS);
    }
    for (int kc = 0; kc < nk; ++kc) {
        const int cur = kc & 1;
        const int nxt = min(kc + 1, nk - 1);
        const float* a = sA + cur * BM * LDS + (wm * WTM + frag_row) * LDS + frag_k;
        const float* b = sB + cur * BN * LDS + (wn * WTN + frag_row) * LDS + frag_k;
        const float* an = sA + (cur ^ 1) * BM * LDS + (wm * WTM + frag_row) * LDS + frag_k;
        const float* bn = sB + (cur ^ 1) * BN * LDS + (wn * WTN + frag_row) * LDS + frag_k;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const int cb = kk & 1, nb = cb ^ 1;
            if (kk < NKK - 1) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[nb][i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDS + (kk + 1) * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[nb][j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDS + (kk + 1) * 8);
            }
            if (!LOAD_AHEAD && kk == QA_LOAD_AT) QA_LOAD_GLOBAL(nxt)
            if (kk == NKK - 1) {
                QA_STORE_LDS(cur ^ 1)
                __syncthreads();
                if (LOAD_AHEAD) QA_LOAD_GLOBAL(min(kc + 2, nk - 1))
#pragma unroll
                for (int i = 0; i < TM; ++i) af[nb][i] = *reinterpret_cast<const f32x4*>(an + i * 32 * LDS);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[nb][j] = *reinterpret_cast<const f32x4*>(bn + j * 32 * LDS);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[cb][j].x, af[cb][i].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[cb][j].y, af[cb][i].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[cb][j].z, af[cb][i].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[cb][j].w, af[cb][i].w, acc[i][j], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#ifdef QA_TIMING
    const long long tloop = __builtin_readcyclecounter();
#endif
#undef QA_LOAD_GLOBAL
#undef QA_STORE_LDS

    // Epilogue.  D layout of the 32x32 MFMA with swapped operands: output row m <- lane & 31, output channel
    // n <- (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5): registers 4g..4g+3 are channels 8g + 4h .. +3 of row m.
    // The accumulators take one trip through LDS (the operand buffers are free now) so that (a) global traffic is whole
    // 512-byte row segments per wave, float4 per lane, for the store AND for the fused residual / gate loads, and (b) the
    // bias / gate / activation / gamma / residual code exists once, in a rolled loop - fully unrolled over the 64
    // accumulator registers it was ~20k instructions per kernel and the epilogue alone cost 20 K-chunks of time.
    constexpr int EP_LD = BN + 4;          // staging row stride (floats)
    constexpr int EP_ROWS = WM * 32;       // rows staged per pass
    constexpr int EP_C4 = BN / 4;          // float4 per row
    static_assert(EP_ROWS * EP_LD <= 2 * (BM + BN) * LDS, "epilogue staging must fit in the operand buffers");
    static_assert(256 % EP_C4 == 0, "a thread keeps its column group across iterations");
    static_assert(EP_C4 % 8 == 0, "arg-min epilogue: the 8 lanes of a 32-column group are consecutive lanes of one wave");
    float* stage = smem;
    const int row_l = lane & 31, col_h = 4 * (lane >> 5);
    const int ep_c4 = tid % EP_C4, ep_r0 = tid / EP_C4;
    const int ep_n = n0 + 4 * ep_c4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f}, one4 = {1.f, 1.f, 1.f, 1.f};
    f32x4 bias4 = zero4, gamma4 = one4;
    if (p.vec_epi && ep_n < p.N) {
        if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + ep_n);
        if (p.gamma) gamma4 = *reinterpret_cast<const f32x4*>(p.gamma + ep_n);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        __syncthreads();  // operand buffers (i = 0) / previous pass (i > 0) no longer read
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                *reinterpret_cast<f32x4*>(stage + (wm * 32 + row_l) * EP_LD + wn * WTN + j * 32 + 8 * g + col_h) = v;
            }
        __syncthreads();
#pragma unroll 1
        for (int r = ep_r0; r < EP_ROWS; r += 256 / EP_C4) {
            const long long m = m0 + (r >> 5) * WTM + i * 32 + (r & 31);
            if (p.am_dist) {  // launch-uniform: the RVQ arg-min epilogue (ConvParams::am_*); every lane takes part in the shuffles
                float best = INFINITY;
                int bi = 0x7fffffff;
                if (m < p.M && ep_n < p.N) {
                    const f32x4 sv = *reinterpret_cast<const f32x4*>(stage + r * EP_LD + 4 * ep_c4);
                    const f32x4 e4 = *reinterpret_cast<const f32x4*>(p.am_e2 + ep_n);
                    const float xx = p.am_x2[m];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {  // codes ascend: strict '<' keeps the first minimum
                        const float dist = (xx - 2.f * sv[e]) + e4[e];
                        if (dist < best) {
                            best = dist;
                            bi = ep_n + e;
                        }
                    }
                }
#pragma unroll
                for (int o = 1; o < 8; o <<= 1) {  // the 8 lanes of a 32-column group are consecutive lanes of one wave (EP_C4 % 8 == 0)
                    const float d2 = __shfl_xor(best, o, 64);
                    const int i2 = __shfl_xor(bi, o, 64);
                    if (d2 < best || (d2 == best && i2 < bi)) {
                        best = d2;
                        bi = i2;
                    }
                }
                if ((ep_c4 & 7) == 0 && m < p.M && ep_n < p.N) {
                    p.am_dist[m * p.am_ld + (ep_n >> 5)] = best;
                    p.am_idx[m * p.am_ld + (ep_n >> 5)] = bi;
                }
                continue;
            }
            if (m >= p.M || ep_n >= p.N) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(stage + r * EP_LD + 4 * ep_c4);
            if (p.vec_epi) {  // N % 4 == 0, every leading dimension and pointer 16-byte aligned
                v = epilogue4(v, p, m, ep_n, bias4, gamma4);
                *reinterpret_cast<f32x4*>(p.y + m * p.ldy + ep_n) = v;
            } else {
                for (int e = 0; e < 4 && ep_n + e < p.N; ++e) {
                    const int n = ep_n + e;
                    float u = v[e] + (p.bias ? p.bias[n] : 0.f);
                    if (p.gate) u = silu_f(p.gate[m * p.ldg + n]) * u;
                    u = apply_act(u, p.act);
                    if (p.gamma) u *= p.gamma[n];
                    if (p.res) u += p.res[m * p.ldr + n];
                    u = apply_act(u, p.post_act);
                    p.y[m * p.ldy + n] = u;
                }
            }
        }
    }
#ifdef QA_TIMING
    if (lane == 0) {
        const long long tend = __builtin_readcyclecounter();
        for (int i = 0; i < 4; ++i) atomicAdd(&g_qa_timing[i], (unsigned long long)tacc[i]);
        atomicAdd(&g_qa_timing[4], (unsigned long long)(tend - tloop));   // epilogue
        atomicAdd(&g_qa_timing[5], (unsigned long long)(tend - tbegin));  // main loop + epilogue
        atomicAdd(&g_qa_timing[6], 1ULL);                                 // waves
        atomicAdd(&g_qa_timing[7], (unsigned long long)nk);               // chunks
        atomicAdd(&g_qa_timing[8], __builtin_amdgcn_s_memrealtime() - rbegin);  // same interval as [5] in 100 MHz ticks
    }
#endif
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg(const ConvParams& p, hipStream_t stream) {
    QA_REQUIRE(p.prologue == ACT_NONE || p.prologue == ACT_ELU, "conv_gemm: prologue %d unsupported", p.prologue);
    const long long tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
    const bool prof = profile_enabled();
    if (prof) {
        const int cfg = BM == 64 ? (BN == 64 ? PROF_CFG_64x64 : PROF_CFG_64x128) : (BN == 32 ? PROF_CFG_128x32 : (BN == 64 ? PROF_CFG_128x64 : PROF_CFG_128x128));
        const double n = p.algo_n ? p.algo_n : p.N, k = p.algo_k ? p.algo_k : p.K;
        // algorithmic bytes: every input frame, weight and output element once (+ fused residual / gate reads)
        const double out_elems = p.am_dist ? 2.0 * (double)p.M * p.am_ld + p.M + n  // arg-min epilogue: (dist, idx) per 32 columns + |r|^2 + |e|^2
                                            : (double)p.M * n * (1.0 + (p.res ? 1.0 : 0.0) + (p.gate ? 1.0 : 0.0));
        const double elems = (double)p.B * p.T_in * p.C_in + n * k + out_elems;
        profile_record_begin(cfg, 2.0 * (double)p.M * n * k, 4.0 * elems, stream, &p);
    }
    // BK = 16 chunks need 45 KB / 35 KB of LDS, so 3-4 workgroups are co-resident per CU (BK = 32: 2) and cover each
    // other's barriers, prologues and epilogues: +10..25 % on the K = 512 layers of the aggregator stacks, +3..5 % on
    // K = 768..3072 (per-shape sweep, tools/gemm_bench.py with QA_GEMM_BK16=0 / default).  QA_GEMM_BK16 = largest K that
    // takes the BK = 16 variant.
    const long long bk16_max_k = knob(K_GEMM_BK16);
    // ... but only when the launch has enough tiles for that co-residency: with about one workgroup per CU nobody covers the
    // exposed latency of the next chunk's global loads, which a BK = 16 chunk's 0.45 us of MFMAs is too short to hide (measured: the
    // 144-tile RVQ distance GEMM 1056 x 1024 x 512 ran 37 us, 2.5 x its MFMA time).  QA_GEMM_BK16_MIN_TILES = fewest tiles that take BK = 16.
    const long long bk16_min_tiles = knob(K_GEMM_BK16_MIN_TILES);
    const bool linear_on = knob(K_GEMM_LINEAR) != 0;
    const bool linear = linear_on && p.ksize == 1 && p.stride == 1 && p.pad_left == 0 && p.in_rep <= 1 && p.T_in == p.T_out &&
                        (p.dilation <= 1);
    const bool bk16 = BN >= 64 && p.prologue != ACT_ELU && ((p.K <= bk16_max_k && tiles >= bk16_min_tiles) || p.C_in % 32 != 0);
    if (bk16 && linear)
        hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, false, (BN >= 64 ? 16 : 32), true>), dim3((unsigned)tiles), dim3(256), 0, stream, p);
    else if (bk16)
        hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, false, (BN >= 64 ? 16 : 32)>), dim3((unsigned)tiles), dim3(256), 0, stream, p);
    else if (p.prologue == ACT_ELU)
        hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, true>), dim3((unsigned)tiles), dim3(256), 0, stream, p);
    else if (linear)
        hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, false, 32, true>), dim3((unsigned)tiles), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, false>), dim3((unsigned)tiles), dim3(256), 0, stream, p);
    if (prof) profile_record_end(stream);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

int launch_conv_gemm(const ConvParams& p, hipStream_t stream) {
    // a K chunk never straddles two taps: C_in must be a multiple of the chunk width (16 for the BK = 16 variants, which
    // exist for N > 32 without the ELU prologue - e.g. the 48-channel groups of the SSL positional convolution; 32 otherwise)
    QA_REQUIRE(p.K % 16 == 0 && p.C_in % 16 == 0, "conv_gemm: K=%d / C_in=%d must be multiples of 16", p.K, p.C_in);
    QA_REQUIRE(p.C_in % 32 == 0 || (p.N > 32 && p.prologue != ACT_ELU),
               "conv_gemm: C_in=%d is not a multiple of 32 (only supported for N > 32 without the ELU prologue)", p.C_in);
    QA_REQUIRE((p.ldx % 4) == 0, "conv_gemm: ldx=%lld must be a multiple of 4 floats", p.ldx);
    QA_REQUIRE((long long)p.T_in * p.ldx < (1LL << 31), "conv_gemm: one batch item spans %lld floats (limit 2^31)",
               (long long)p.T_in * p.ldx);
    QA_REQUIRE(((uintptr_t)p.x % 16) == 0 && ((uintptr_t)p.w % 16) == 0, "conv_gemm: x / w must be 16-byte aligned");
    if (p.M <= 0 || p.N <= 0) return QA_OK;
    QA_REQUIRE(ceil_div(p.M, 64) * ceil_div(p.N, 32) < (1LL << 31), "conv_gemm: grid too large");
    const int forced = (int)knob(K_GEMM_CFG);
    const int swz = (int)knob(K_GEMM_XCD);
    ConvParams q = p;
    q.xcd_swizzle = swz;
    q.panel = (int)knob(K_GEMM_PANEL);
    // frame / in_rep by multiply-high: exact for frame * in_rep < 2^32 (frames of one clip are < 2^31 / ldx)
    auto al16 = [](const void* ptr) { return ((uintptr_t)ptr % 16) == 0; };
    q.vec_epi = p.N % 4 == 0 && p.ldy % 4 == 0 && al16(p.y) && (!p.bias || al16(p.bias)) && (!p.gamma || al16(p.gamma)) &&
                (!p.res || (p.ldr % 4 == 0 && al16(p.res))) && (!p.gate || (p.ldg % 4 == 0 && al16(p.gate)));
    q.rep_one = p.in_rep > 1 ? 0u : 1u;
    q.rep_magic = p.in_rep > 1 ? (unsigned)(((1ULL << 32) + p.in_rep - 1) / p.in_rep) : 0u;
    QA_REQUIRE(p.in_rep <= 1 || p.pad_mode == PAD_ZERO, "conv_gemm: in_rep needs zero padding");
    const bool snake = p.act == ACT_SNAKE || p.post_act == ACT_SNAKE;
    QA_REQUIRE((!snake && !p.y2) || (q.vec_epi && (!snake || (p.alpha && al16(p.alpha))) &&
                                     (!p.y2 || (p.alpha2 && al16(p.alpha2) && al16(p.y2) && p.ldy2 % 4 == 0))),
               "conv_gemm: Snake activation / second output need the float4 epilogue and 16-byte aligned alpha vectors");
    QA_REQUIRE(p.dilation <= 1 || (p.pad_mode == PAD_ZERO && p.in_rep <= 1), "conv_gemm: dilation needs zero padding");
    QA_REQUIRE(!p.am_dist || (q.vec_epi && p.am_idx && p.am_x2 && p.am_e2 && al16(p.am_e2) && p.am_ld >= (p.N + 31) / 32 && !p.y2),
               "conv_gemm: the arg-min epilogue needs N %% 4 == 0, x2 / e2 / dist / idx and am_ld >= ceil(N / 32)");
    QA_REQUIRE(!p.rope || (q.vec_epi && p.rope_hd % 4 == 0 && p.rope_n % 4 == 0 && p.rope_T > 0 && al16(p.rope)),
               "conv_gemm: fused RoPE needs the float4 epilogue (N, strides, pointers multiples of 4 / 16 B)");
    int cfg;
    if (forced >= 0) cfg = forced;
    else if (p.N <= 32) cfg = PROF_CFG_128x32;
    else if (p.N <= 64) cfg = PROF_CFG_128x64;
    else {
        // Tiles of one launch are dealt round-robin over 256 CUs (the co-resident workgroups of a CU share its matrix pipes), so the
        // makespan is (tiles on the busiest CU) x (work per tile) / (sustained efficiency of the tile).  Round 4 adds the 64-row
        // tiles: the aggregator stacks of H-Codec 1.5 (M = 9056 = 70.75 x 128) and the 4000-row SSL / BiCodec layers lose up to a
        // third of the machine to tile quantisation with 128-row tiles (852 tiles of 128 x 128 = 3.33 per CU: a fourth round for a
        // third of the CUs).  Efficiencies from the square 8192 x 4096 x 4096 problem, where every tile divides the grid evenly
        // (profiles/r04_gemm_tile_sweep.txt: 133.1 / 121.5 / 121.6 / 115.9 TFLOP/s); ties go to the larger tile (less L2 traffic).
        // Every configuration accumulates an output element over k in the same order, so this choice - which depends on M, i.e.
        // on the batch size - never changes a bit (tests/test_kernels_gpu.py::test_conv_gemm_tile_configurations_are_bit_identical).
        struct Cand { int cfg, bm, bn; double eff; };
        static const Cand cands[] = {{PROF_CFG_128x128, 128, 128, 1.0}, {PROF_CFG_64x128, 64, 128, 0.914}, {PROF_CFG_128x64, 128, 64, 0.913},
                                     {PROF_CFG_64x64, 64, 64, 0.871}};
        double best = 0.0;
        cfg = PROF_CFG_128x128;
        for (const Cand& c : cands) {
            const long long tiles = ceil_div(p.M, c.bm) * ceil_div(p.N, c.bn);
            // a launch that gives a CU at most ONE workgroup has nobody to cover that workgroup's barriers, prologue and epilogue: the
            // 256-tile 4032 x 512 x 512 launch runs 8 % faster as 504 tiles of 64 x 64 although those need two rounds (same sweep)
            const double cost = (double)ceil_div(tiles, 256) * c.bm * c.bn / c.eff * (tiles <= 256 ? 1.25 : 1.0);
            if (best == 0.0 || cost < best * 0.995) {
                best = cost;
                cfg = c.cfg;
            }
        }
    }
    switch (cfg) {
        case PROF_CFG_128x32: return launch_cfg<128, 32, 4, 1>(q, stream);
        case PROF_CFG_128x64: return launch_cfg<128, 64, 2, 2>(q, stream);
        case PROF_CFG_64x128: return launch_cfg<64, 128, 1, 4>(q, stream);
        case PROF_CFG_64x64: return launch_cfg<64, 64, 2, 2>(q, stream);
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
