// split_gemm.hip - the experiment VERDICT r04 item 1 asks for, as a standalone program: an fp32 GEMM whose operands are split into three
// bf16 planes (x = h + m + l exactly: 3 x 8 significand bits cover fp32's 24) and multiplied as 9 plane products on the bf16 matrix pipe
// (v_mfma_f32_32x32x16_bf16, 16 x the rate of v_mfma_f32_32x32x2_f32), accumulated in fp32.  Every single product of two bf16 values is
// exact in fp32 (16 significand bits), so the result differs from the fp32 fma chain only in where the roundings of the SUM fall.
//   part 1 (numerics): error of the 9-product form against an fp64 reference, beside the error of the fp32 fmaf chain the product kernel
//                      computes today, on uniform and on wide-dynamic-range operands; a probe of the MFMA's internal adder.
//   part 2 (speed):    one 128 x 128 x 16 tile kernel (LINEAR shapes only) on the shapes that carry H-Codec 1.5's FLOPs.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/split_gemm.hip -o tools/micro/split_gemm
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(e)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (e);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);  \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

// ---- the split: truncation to the top 16 bits, twice; the third remainder has <= 8 significant bits and IS a bf16 ----
__device__ __host__ inline void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    unsigned xb;
    memcpy(&xb, &x, 4);
    const unsigned hb = xb & 0xffff0000u;
    float hf;
    memcpy(&hf, &hb, 4);
    const float r = x - hf;  // exact
    unsigned rb;
    memcpy(&rb, &r, 4);
    const unsigned mb = rb & 0xffff0000u;
    float mf;
    memcpy(&mf, &mb, 4);
    const float r2 = r - mf;  // exact, <= 8 significant bits
    unsigned lb;
    memcpy(&lb, &r2, 4);
    h = hb >> 16;
    m = mb >> 16;
    l = lb >> 16;
}

// 8 consecutive fp32 -> three u32x4 of 8 bf16 each.  v_perm_b32 packs the two high halves of a pair in one instruction.
__device__ __forceinline__ void split8(const f32x4 lo, const f32x4 hi, u32x4& ph, u32x4& pm, u32x4& pl) {
    float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    unsigned xb[8], rb[8], r2b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        xb[i] = __builtin_bit_cast(unsigned, x[i]);
        const float r = x[i] - __builtin_bit_cast(float, xb[i] & 0xffff0000u);
        rb[i] = __builtin_bit_cast(unsigned, r);
        const float r2 = r - __builtin_bit_cast(float, rb[i] & 0xffff0000u);
        r2b[i] = __builtin_bit_cast(unsigned, r2);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ph[i] = __builtin_amdgcn_perm(xb[2 * i + 1], xb[2 * i], 0x07060302u);
        pm[i] = __builtin_amdgcn_perm(rb[2 * i + 1], rb[2 * i], 0x07060302u);
        pl[i] = __builtin_amdgcn_perm(r2b[2 * i + 1], r2b[2 * i], 0x07060302u);
    }
}

// weights, once: W [N, K] fp32 -> Wp [N][K/8][3 planes][8] bf16 (48 contiguous bytes per (row, k-group))
__global__ void prepack_kernel(const float* w, u32x4* wp, long long n_groups) {
    const long long g = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(w + g * 8), hi = *reinterpret_cast<const f32x4*>(w + g * 8 + 4);
    u32x4 ph, pm, pl;
    split8(lo, hi, ph, pm, pl);
    wp[g * 3 + 0] = ph;
    wp[g * 3 + 1] = pm;
    wp[g * 3 + 2] = pl;
}

// ---- probe of the matrix pipe's adder: D = sum_k a_k b_k + C for hand-made operands (one lane's worth of interest) ----
__global__ void probe_kernel(const unsigned short* a16, const unsigned short* b16, const float* c, float* d) {
    // A[i][k] = a16[k] for every i, B[k][j] = b16[k] for every j, C = c[0] everywhere: every D element is the same dot product
    const int lane = threadIdx.x;
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = __builtin_bit_cast(__bf16, a16[(lane >> 5) * 8 + j]);
        b[j] = __builtin_bit_cast(__bf16, b16[(lane >> 5) * 8 + j]);
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = c[0];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (lane == 0) d[0] = acc[0];
}

// ---- the GEMM: y[m, n] = sum_k x[m, k] w[n, k] (+ bias[n]), x fp32 [M, K] split on the way into LDS, w pre-split ----
// LDS image of one operand plane of a K = 16 chunk: rows of 16 bf16 = two 16-byte slots; slot (row, half) sits at
// row * 2 + (half ^ ((row >> 3) & 1)): the 16 lanes one ds_read_b128 group serves then cover all 64 banks.
#ifndef SG_VARIANT  // diagnostic builds (timing only, results wrong): 1 no global loads in the loop, 2 no split (raw bits stored), 4 no LDS stores,
#define SG_VARIANT 0  // 8 no barrier in the loop, 16 fragments read once before the loop
#endif
#ifndef SG_STORE_AT
#define SG_STORE_AT 4  // MFMA group (of 9) in front of which the next chunk's split + LDS stores are placed
#endif
template <int BM, int BN>
__global__ __launch_bounds__(256, 3) void split_gemm_kernel(const float* __restrict__ x, const u32x4* __restrict__ wp, const float* __restrict__ bias,
                                                            float* __restrict__ y, int M, int N, int K, int panel) {
    constexpr int WM = 2, WN = 2;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr int A_IT = BM / 128, B_IT = BN / 128;  // 256 threads cover 128 rows x 2 halves per pass
    static_assert(A_IT >= 1 && B_IT >= 1, "tile rows");
    constexpr int A_SLOTS = BM * 2, B_SLOTS = BN * 2;             // 16-byte slots per plane
    constexpr int STAGE = 3 * (A_SLOTS + B_SLOTS);                // slots per stage
    __shared__ u32x4 smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    int tile = blockIdx.x;
    {  // XCD-aware order + column panels, as conv_gemm.hip
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = tile & 7, local = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    int tm_i = tile / tiles_n, tn_i = tile % tiles_n;
    if (panel > 0 && tiles_n > panel) {
        const int per_panel = tiles_m * panel;
        const int pn = tile / per_panel, r = tile - pn * per_panel;
        const int pw = min(panel, tiles_n - pn * panel);
        tm_i = r / pw;
        tn_i = pn * panel + (r - tm_i * pw);
    }
    const int m0 = tm_i * BM, n0 = tn_i * BN;

    const int ld_row = tid >> 1, ld_half = tid & 1;
    const int ld_slot = ld_row * 2 + (ld_half ^ ((ld_row >> 3) & 1));
    const float* a_ptr[A_IT];
    const u32x4* b_ptr[B_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) a_ptr[i] = x + (long long)min(m0 + ld_row + 128 * i, M - 1) * K + ld_half * 8;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) b_ptr[i] = wp + ((long long)min(n0 + ld_row + 128 * i, N - 1) * (K / 8) + ld_half) * 3;
    const int nk = K / 16;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 a_lo[A_IT], a_hi[A_IT];
    u32x4 b_reg[B_IT][3];
#define SG_LOAD(KC)                                                                        \
    {                                                                                      \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                 \
            a_lo[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + (KC) * 16);               \
            a_hi[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + (KC) * 16 + 4);           \
        }                                                                                  \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i) {                                 \
            _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) b_reg[i][pl] = b_ptr[i][(KC) * 6 + pl]; \
        }                                                                                  \
    }
#define SG_STORE(BUF)                                                                      \
    {                                                                                      \
        u32x4* sa_ = smem + (BUF) * STAGE;                                                 \
        u32x4* sb_ = sa_ + 3 * A_SLOTS;                                                    \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                 \
            u32x4 ph_, pm_, pl_;                                                           \
            if (SG_VARIANT & 2) {                                                          \
                ph_ = __builtin_bit_cast(u32x4, a_lo[i]); pm_ = __builtin_bit_cast(u32x4, a_hi[i]); pl_ = ph_; \
            } else                                                                         \
                split8(a_lo[i], a_hi[i], ph_, pm_, pl_);                                   \
            sa_[0 * A_SLOTS + ld_slot + 256 * i] = ph_;                                    \
            sa_[1 * A_SLOTS + ld_slot + 256 * i] = pm_;                                    \
            sa_[2 * A_SLOTS + ld_slot + 256 * i] = pl_;                                    \
        }                                                                                  \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i) {                                 \
            _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) sb_[pl * B_SLOTS + ld_slot + 256 * i] = b_reg[i][pl]; \
        }                                                                                  \
    }

    SG_LOAD(0)
    SG_STORE(0)
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    // plane pairs (activation plane, weight plane), smallest products first: 0 = h, 1 = m, 2 = l
    constexpr int PA[9] = {2, 2, 1, 1, 2, 0, 1, 0, 0};
    constexpr int PB[9] = {2, 1, 2, 1, 0, 2, 0, 1, 0};
    bf16x8 af[3][TM], bfr[3][TN];
#define SG_FRAGS(CUR)                                                                                          \
    {                                                                                                          \
        const u32x4* sa = smem + (CUR) * STAGE;                                                                \
        const u32x4* sb = sa + 3 * A_SLOTS;                                                                    \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) {                                                     \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                   \
                const int row = wm * WTM + i * 32 + fr;                                                        \
                af[pl][i] = __builtin_bit_cast(bf16x8, sa[pl * A_SLOTS + row * 2 + (fh ^ ((row >> 3) & 1))]);  \
            }                                                                                                  \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                   \
                const int row = wn * WTN + j * 32 + fr;                                                        \
                bfr[pl][j] = __builtin_bit_cast(bf16x8, sb[pl * B_SLOTS + row * 2 + (fh ^ ((row >> 3) & 1))]); \
            }                                                                                                  \
        }                                                                                                      \
    }
    if (SG_VARIANT & 16) SG_FRAGS(0)
    for (int kc = 0; kc < nk; ++kc) {
        const int cur = kc & 1;
        const int nxt = min(kc + 1, nk - 1);
        if (!(SG_VARIANT & 16)) SG_FRAGS(cur)
        if (!(SG_VARIANT & 1)) SG_LOAD(nxt)
        __builtin_amdgcn_sched_barrier(0);
        // region 1: MFMA groups 0 .. SG_STORE_AT-1 alone (the next chunk's global loads are in flight underneath them)
#pragma unroll
        for (int g = 0; g < SG_STORE_AT; ++g)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[PB[g]][j], af[PA[g]][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // region 2: the split of the next chunk's activations (44 VALU per staged row group) and the six LDS stores, dealt out BETWEEN this
        // wave's own remaining MFMAs: 3 VALU per MFMA slot, then the stores
        if (!(SG_VARIANT & 4)) SG_STORE(cur ^ 1)
#pragma unroll
        for (int g = SG_STORE_AT; g < 9; ++g)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[PB[g]][j], af[PA[g]][i], acc[i][j], 0, 0, 0);
        {
            constexpr int NM = (9 - SG_STORE_AT) * TM * TN;          // MFMAs of region 2
            constexpr int NV = 46 * A_IT, NW = 3 * (A_IT + B_IT);     // VALU / DS-write instructions to place
            constexpr int VM = NM - NW / 2 - 1;                       // MFMA slots that carry VALU
            constexpr int VPER = (NV + VM - 1) / VM;
#pragma unroll
            for (int q = 0; q < VM; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, VPER, 0);
            }
#pragma unroll
            for (int q = 0; q < NW / 2; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
            }
        }
        if (!(SG_VARIANT & 8)) __syncthreads();
    }
#undef SG_LOAD
#undef SG_STORE

    // epilogue through LDS (as conv_gemm.hip): whole 512-byte row segments per wave
    constexpr int EP_LD = BN + 4, EP_ROWS = WM * 32, EP_C4 = BN / 4;
    static_assert(EP_ROWS * EP_LD * 4 <= 2 * STAGE * 16, "epilogue staging fits");
    float* stage = reinterpret_cast<float*>(smem);
    const int row_l = lane & 31, col_h = 4 * (lane >> 5);
    const int ep_c4 = tid % EP_C4, ep_r0 = tid / EP_C4;
    const int ep_n = n0 + 4 * ep_c4;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (bias && ep_n < N) bias4 = *reinterpret_cast<const f32x4*>(bias + ep_n);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        __syncthreads();
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
            if (m >= M || ep_n >= N) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(stage + r * EP_LD + 4 * ep_c4);
            v += bias4;
            *reinterpret_cast<f32x4*>(y + m * N + ep_n) = v;
        }
    }
}

// ---- version 2: NO ds_write at all.  Both operands reach LDS by LDS-DMA (global_load_lds_dwordx4: wave-uniform LDS base + lane x 16 B,
// per-lane global address, so the swizzle lives in the SOURCE address): the pre-split weight planes as they are, the activations as raw fp32;
// the activation split happens on the fragment a wave has just read (2 x ds_read_b128 = 8 consecutive k of one row -> three bf16x8).
// r05 diagnosis (profiles/r05_split_gemm_micro.txt): with register staging the six ds_write_b128 per thread and chunk cost 1068 -> 707 us on
// 16000 x 4096 x 1024 - the VGPR -> LDS store path, not the matrix pipe, bounded version 1.
//   A stage: rows of 16 fp32 = four 16-byte slots, slot (row, q) at row * 4 + (q ^ ((row >> 2) & 3))
//   B stage: per plane rows of 16 bf16 = two slots, slot (row, half) at row * 2 + (half ^ ((row >> 3) & 1))
#define SG_LDS(p) ((__attribute__((address_space(3))) void*)(p))
#define SG_GLB(p) ((const __attribute__((address_space(1))) void*)(p))
template <int BM, int BN>
__global__ __launch_bounds__(256, 3) void split_gemm_dma_kernel(const float* __restrict__ x, const u32x4* __restrict__ wp, const float* __restrict__ bias,
                                                                float* __restrict__ y, int M, int N, int K, int panel) {
    constexpr int WM = 2, WN = 2;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr int A_SLOTS = BM * 4, B_SLOTS = BN * 2;        // 16-byte slots of the fp32 activation stage / of ONE weight plane
    constexpr int STAGE = A_SLOTS + 3 * B_SLOTS;
    constexpr int A_DMA = A_SLOTS / 256, B_DMA = 3 * B_SLOTS / 256;  // wave-instructions (64 slots each) per wave and chunk
    static_assert(A_SLOTS % 256 == 0 && (3 * B_SLOTS) % 256 == 0, "whole DMA instructions per wave");
    __shared__ u32x4 smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    int tile = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = tile & 7, local = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    int tm_i = tile / tiles_n, tn_i = tile % tiles_n;
    if (panel > 0 && tiles_n > panel) {
        const int per_panel = tiles_m * panel;
        const int pn = tile / per_panel, r = tile - pn * per_panel;
        const int pw = min(panel, tiles_n - pn * panel);
        tm_i = r / pw;
        tn_i = pn * panel + (r - tm_i * pw);
    }
    const int m0 = tm_i * BM, n0 = tn_i * BN;

    // per-lane DMA sources: wave w fills slots [64 (w A_DMA + i), +64) of the activation stage and [64 (w B_DMA + i), +64) of the weight stage
    const float* a_src[A_DMA];
    const u32x4* b_src[B_DMA];
#pragma unroll
    for (int i = 0; i < A_DMA; ++i) {
        const int s = 64 * (wave * A_DMA + i) + lane, row = s >> 2, q = (s & 3) ^ ((row >> 2) & 3);
        a_src[i] = x + (long long)min(m0 + row, M - 1) * K + q * 4;
    }
#pragma unroll
    for (int i = 0; i < B_DMA; ++i) {
        const int s = 64 * (wave * B_DMA + i) + lane, pl = s / B_SLOTS, sp = s % B_SLOTS, row = sp >> 1, half = (sp & 1) ^ ((row >> 3) & 1);
        b_src[i] = wp + ((long long)min(n0 + row, N - 1) * (K / 8) + half) * 3 + pl;
    }
    const int nk = K / 16;
#define SG_DMA(KC, BUF)                                                                                                                   \
    {                                                                                                                                     \
        u32x4* st_ = smem + (BUF) * STAGE;                                                                                                \
        _Pragma("unroll") for (int i = 0; i < A_DMA; ++i)                                                                                 \
            __builtin_amdgcn_global_load_lds(SG_GLB(a_src[i] + (KC) * 16), SG_LDS(st_ + 64 * (wave * A_DMA + i)), 16, 0, 0);              \
        _Pragma("unroll") for (int i = 0; i < B_DMA; ++i)                                                                                 \
            __builtin_amdgcn_global_load_lds(SG_GLB(b_src[i] + (KC) * 6), SG_LDS(st_ + A_SLOTS + 64 * (wave * B_DMA + i)), 16, 0, 0);     \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    SG_DMA(0, 0)
    __syncthreads();  // hipcc's barrier drains vmcnt(0) while an LDS-DMA is in flight: chunk 0 has landed for everybody

    const int fr = lane & 31, fh = lane >> 5;
    constexpr int PA[9] = {2, 2, 1, 1, 2, 0, 1, 0, 0};
    constexpr int PB[9] = {2, 1, 2, 1, 0, 2, 0, 1, 0};
    for (int kc = 0; kc < nk; ++kc) {
        const int cur = kc & 1;
        if (kc + 1 < nk) SG_DMA(kc + 1, cur ^ 1)
        const u32x4* sa = smem + cur * STAGE;
        const u32x4* sb = sa + A_SLOTS;
        u32x4 araw[TM][2];
        bf16x8 bfr[3][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wm * WTM + i * 32 + fr;
#pragma unroll
            for (int q = 0; q < 2; ++q) araw[i][q] = sa[row * 4 + ((2 * fh + q) ^ ((row >> 2) & 3))];
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * WTN + j * 32 + fr;
                bfr[pl][j] = __builtin_bit_cast(bf16x8, sb[pl * B_SLOTS + row * 2 + (fh ^ ((row >> 3) & 1))]);
            }
        bf16x8 af[3][TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            u32x4 ph, pm, pl;
            split8(__builtin_bit_cast(f32x4, araw[i][0]), __builtin_bit_cast(f32x4, araw[i][1]), ph, pm, pl);
            af[0][i] = __builtin_bit_cast(bf16x8, ph);
            af[1][i] = __builtin_bit_cast(bf16x8, pm);
            af[2][i] = __builtin_bit_cast(bf16x8, pl);
        }
        // products in an order that lets the MFMAs of row block 0 start while row block 1 is still being split
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 9; ++g)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[PB[g]][j], af[PA[g]][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);  // keep the barrier (and the vmcnt(0) hipcc puts in front of it) BEHIND the chunk's MFMAs
        __syncthreads();
    }
#undef SG_DMA

    constexpr int EP_LD = BN + 4, EP_ROWS = WM * 32, EP_C4 = BN / 4;
    static_assert(EP_ROWS * EP_LD * 4 <= 2 * STAGE * 16, "epilogue staging fits");
    float* stage = reinterpret_cast<float*>(smem);
    const int row_l = lane & 31, col_h = 4 * (lane >> 5);
    const int ep_c4 = tid % EP_C4, ep_r0 = tid / EP_C4;
    const int ep_n = n0 + 4 * ep_c4;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (bias && ep_n < N) bias4 = *reinterpret_cast<const f32x4*>(bias + ep_n);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        __syncthreads();
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
            if (m >= M || ep_n >= N) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(stage + r * EP_LD + 4 * ep_c4);
            v += bias4;
            *reinterpret_cast<f32x4*>(y + m * N + ep_n) = v;
        }
    }
}

// ---- version 3: version 2 + a software pipeline.  r05 finding: versions 1 and 2 run at the SAME 123 - 125 TFLOP/s whether the operands are
// staged through registers or by DMA - what a workgroup loses per chunk is the serial tail  [wait for the stage] -> barrier -> [fragment
// reads] -> first MFMA, and the co-resident workgroups (sharing the matrix pipe MFMA by MFMA) fall into lockstep, so nobody covers it.
// Here the tail is taken out of the chunk: three LDS stages, the DMA of chunk kc+2 and the fragment reads + activation split of chunk
// kc+1 are issued underneath the MFMAs of chunk kc (two register sets of fragments, loop unrolled by two), so a chunk ends with a bare
// vmcnt(0) + barrier and the next chunk's first MFMA has its operands in registers.
#ifndef SG_LEAD
#define SG_LEAD 8   // MFMAs issued before the first split instruction (covers the LDS latency of the raw activation fragments)
#endif
#ifndef SG_VPM
#define SG_VPM 4    // split VALU instructions per MFMA slot after that
#endif
__device__ unsigned long long g_clk[2];  // [0] shader cycles, [1] 100 MHz ticks, summed over workgroups (main loop of wave 0)
template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void split_gemm_pipe_kernel(const float* __restrict__ x, const u32x4* __restrict__ wp, const float* __restrict__ bias,
                                                                 float* __restrict__ y, int M, int N, int K, int panel) {
    constexpr int WM = 2, WN = 2;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr int A_SLOTS = BM * 4, B_SLOTS = BN * 2;
    constexpr int STAGE = A_SLOTS + 3 * B_SLOTS;
    constexpr int A_DMA = A_SLOTS / 256, B_DMA = 3 * B_SLOTS / 256;
    static_assert(A_SLOTS % 256 == 0 && (3 * B_SLOTS) % 256 == 0, "whole DMA instructions per wave");
    __shared__ u32x4 smem[3 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    int tile = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = tile & 7, local = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    int tm_i = tile / tiles_n, tn_i = tile % tiles_n;
    if (panel > 0 && tiles_n > panel) {
        const int per_panel = tiles_m * panel;
        const int pn = tile / per_panel, r = tile - pn * per_panel;
        const int pw = min(panel, tiles_n - pn * panel);
        tm_i = r / pw;
        tn_i = pn * panel + (r - tm_i * pw);
    }
    const int m0 = tm_i * BM, n0 = tn_i * BN;

    const float* a_src[A_DMA];
    const u32x4* b_src[B_DMA];
#pragma unroll
    for (int i = 0; i < A_DMA; ++i) {
        const int s = 64 * (wave * A_DMA + i) + lane, row = s >> 2, q = (s & 3) ^ ((row >> 2) & 3);
        a_src[i] = x + (long long)min(m0 + row, M - 1) * K + q * 4;
    }
#pragma unroll
    for (int i = 0; i < B_DMA; ++i) {
        const int s = 64 * (wave * B_DMA + i) + lane, pl = s / B_SLOTS, sp = s % B_SLOTS, row = sp >> 1, half = (sp & 1) ^ ((row >> 3) & 1);
        b_src[i] = wp + ((long long)min(n0 + row, N - 1) * (K / 8) + half) * 3 + pl;
    }
    const int nk = K / 16;  // even (K % 32 == 0)
#define SG_DMA(KC, ST)                                                                                                                    \
    {                                                                                                                                     \
        u32x4* st_ = smem + (ST) * STAGE;                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < A_DMA; ++i)                                                                                 \
            __builtin_amdgcn_global_load_lds(SG_GLB(a_src[i] + (long long)(KC) * 16), SG_LDS(st_ + 64 * (wave * A_DMA + i)), 16, 0, 0);   \
        _Pragma("unroll") for (int i = 0; i < B_DMA; ++i)                                                                                 \
            __builtin_amdgcn_global_load_lds(SG_GLB(b_src[i] + (long long)(KC) * 6), SG_LDS(st_ + A_SLOTS + 64 * (wave * B_DMA + i)), 16, 0, 0); \
    }
    const int fr = lane & 31, fh = lane >> 5;
    int a_off[TM][2], b_off[TN];  // slot offsets of this lane's fragments inside a stage
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wm * WTM + i * 32 + fr;
#pragma unroll
        for (int q = 0; q < 2; ++q) a_off[i][q] = row * 4 + ((2 * fh + q) ^ ((row >> 2) & 3));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn * WTN + j * 32 + fr;
        b_off[j] = A_SLOTS + row * 2 + (fh ^ ((row >> 3) & 1));
    }
#define SG_FRAG_READ(ST, ARAW, BFR)                                                                                     \
    {                                                                                                                   \
        const u32x4* st_ = smem + (ST) * STAGE;                                                                         \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                                \
            _Pragma("unroll") for (int q = 0; q < 2; ++q) ARAW[i][q] = st_[a_off[i][q]];                                \
        }                                                                                                               \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) {                                                              \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) BFR[pl][j] = __builtin_bit_cast(bf16x8, st_[pl * B_SLOTS + b_off[j]]); \
        }                                                                                                               \
    }
#define SG_SPLIT(ARAW, AF)                                                                                              \
    {                                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                                \
            u32x4 ph_, pm_, pl_;                                                                                        \
            split8(__builtin_bit_cast(f32x4, ARAW[i][0]), __builtin_bit_cast(f32x4, ARAW[i][1]), ph_, pm_, pl_);        \
            AF[0][i] = __builtin_bit_cast(bf16x8, ph_);                                                                 \
            AF[1][i] = __builtin_bit_cast(bf16x8, pm_);                                                                 \
            AF[2][i] = __builtin_bit_cast(bf16x8, pl_);                                                                 \
        }                                                                                                               \
    }
#define SG_MFMA(AF, BFR)                                                                                                \
    {                                                                                                                   \
        _Pragma("unroll") for (int g = 0; g < 9; ++g) {                                                                 \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                            \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                          \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BFR[PB[g]][j], AF[PA[g]][i], acc[i][j], 0, 0, 0); \
            }                                                                                                           \
        }                                                                                                               \
    }
#define SG_PATTERN()                                                                                                    \
    {                                                                                                                   \
        __builtin_amdgcn_sched_group_barrier(0x008, SG_LEAD, 0);                                                        \
        _Pragma("unroll") for (int q_ = 0; q_ < 9 * TM * TN - SG_LEAD; ++q_) {                                          \
            __builtin_amdgcn_sched_group_barrier(0x002, SG_VPM, 0);                                                     \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                          \
        }                                                                                                               \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    constexpr int PA[9] = {2, 2, 1, 1, 2, 0, 1, 0, 0};
    constexpr int PB[9] = {2, 1, 2, 1, 0, 2, 0, 1, 0};

    SG_DMA(0, 0)
    SG_DMA(1, 1)
    __syncthreads();  // hipcc drains vmcnt(0) in front of a barrier while an LDS-DMA is in flight: chunks 0 and 1 have landed
    u32x4 araw[TM][2];
    bf16x8 af0[3][TM], bf0[3][TN], af1[3][TM], bf1[3][TN];
    SG_FRAG_READ(0, araw, bf0)
    SG_SPLIT(araw, af0)
    int st_next = 1, st_dma = 2;
    const long long c0 = __builtin_readcyclecounter();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int kc = 0; kc < nk; kc += 2) {
        SG_DMA(min(kc + 2, nk - 1), st_dma)  // unconditional (the tail re-loads the last chunk into a free stage): the body stays ONE basic block
        SG_FRAG_READ(st_next, araw, bf1)
        __builtin_amdgcn_sched_barrier(0);
        SG_MFMA(af0, bf0)
        SG_SPLIT(araw, af1)
        SG_PATTERN()
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        st_next = st_dma;
        st_dma = st_dma == 2 ? 0 : st_dma + 1;
        SG_DMA(min(kc + 3, nk - 1), st_dma)
        SG_FRAG_READ(st_next, araw, bf0)
        __builtin_amdgcn_sched_barrier(0);
        SG_MFMA(af1, bf1)
        SG_SPLIT(araw, af0)
        SG_PATTERN()
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        st_next = st_dma;
        st_dma = st_dma == 2 ? 0 : st_dma + 1;
    }
#undef SG_DMA
    if (tid == 0) {
        atomicAdd(&g_clk[0], (unsigned long long)(__builtin_readcyclecounter() - c0));
        atomicAdd(&g_clk[1], __builtin_amdgcn_s_memrealtime() - r0);
    }

    constexpr int EP_LD = BN + 4, EP_ROWS = WM * 32, EP_C4 = BN / 4;
    static_assert(EP_ROWS * EP_LD * 4 <= 3 * STAGE * 16, "epilogue staging fits");
    float* stage = reinterpret_cast<float*>(smem);
    const int row_l = lane & 31, col_h = 4 * (lane >> 5);
    const int ep_c4 = tid % EP_C4, ep_r0 = tid / EP_C4;
    const int ep_n = n0 + 4 * ep_c4;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (bias && ep_n < N) bias4 = *reinterpret_cast<const f32x4*>(bias + ep_n);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        __syncthreads();
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
            if (m >= M || ep_n >= N) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(stage + r * EP_LD + 4 * ep_c4);
            v += bias4;
            *reinterpret_cast<f32x4*>(y + m * N + ep_n) = v;
        }
    }
}

// pure-register bf16 MFMA loop with random operand bits: the matrix pipe's sustained rate (clock under load) on this box
__global__ __launch_bounds__(256) void peak_kernel(const u32x4* seed, float* out, int iters) {
    bf16x8 a[3], b[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, seed[(threadIdx.x * 6 + i) & 1023]);
        b[i] = __builtin_bit_cast(bf16x8, seed[(threadIdx.x * 6 + 3 + i) & 1023]);
    }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 9; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[g % 3], a[g / 3], acc[i], 0, 0, 0);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

static double now_ms(hipEvent_t e0, hipEvent_t e1) {
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

static void host_split(float x, float& h, float& m, float& l) {
    unsigned a, b, c;
    split3(x, a, b, c);
    a <<= 16; b <<= 16; c <<= 16;
    memcpy(&h, &a, 4); memcpy(&m, &b, 4); memcpy(&l, &c, 4);
}

int main(int argc, char** argv) {
    const bool speed_only = argc > 1 && !strcmp(argv[1], "speed");
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    {
        std::vector<unsigned> hs(4096);
        std::mt19937 rng(7);
        for (auto& v : hs) v = (rng() & 0x7fff7fffu) | 0x30003000u;  // bf16 pairs of moderate magnitude, random significands
        for (auto& v : hs) v &= 0xbfffbfffu;
        u32x4* ds;
        float* dout;
        CK(hipMalloc(&ds, 16384)); CK(hipMalloc(&dout, 1024 * 256 * 4));
        CK(hipMemcpy(ds, hs.data(), 16384, hipMemcpyHostToDevice));
        for (int wg_per_cu = 1; wg_per_cu <= 3; wg_per_cu += 2) {
            const int blocks = 256 * wg_per_cu, iters = 20000;
            peak_kernel<<<blocks, 256>>>(ds, dout, 100);
            CK(hipEventRecord(e0));
            peak_kernel<<<blocks, 256>>>(ds, dout, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            const double ms = now_ms(e0, e1);
            const double flop = (double)blocks * 4 * iters * 36 * 32768.0;
            printf("== bf16 32x32x16 MFMA register loop, %d workgroup(s) per CU: %.1f ms, %.0f TFLOP/s bf16 = %.1f fp32-equivalent (9 products)\n", wg_per_cu, ms, flop / ms / 1e9,
                   flop / ms / 1e9 / 9.0);
        }
    }
    if (!speed_only) {
        // ---------- probe: what does the pipe's adder keep? ----------
        printf("== probe of v_mfma_f32_32x32x16_bf16's adder (D = sum_k a_k b_k + C, one instruction)\n");
        unsigned short *da, *db;
        float *dc, *dd;
        CK(hipMalloc(&da, 32)); CK(hipMalloc(&db, 32)); CK(hipMalloc(&dc, 4)); CK(hipMalloc(&dd, 4));
        auto bf = [](float v) { unsigned u; memcpy(&u, &v, 4); return (unsigned short)(u >> 16); };
        struct Case { const char* name; float a[16], b[16], c; double exact; };
        std::vector<Case> cases;
        {
            Case c{"C = 2^24, one product 1.0 (exact 16777217: fp32 rounds to even 16777216)", {}, {}, 16777216.f, 16777217.0};
            c.a[0] = 1.f; c.b[0] = 1.f; cases.push_back(c);
            Case d{"C = 2^24, sixteen products 0.25 (sum 4: exact 16777220; per-product rounding would give 16777216)", {}, {}, 16777216.f, 16777220.0};
            for (int k = 0; k < 16; ++k) { d.a[k] = 0.5f; d.b[k] = 0.5f; } cases.push_back(d);
            Case e{"C = 0, products 1 + 15 x 2^-26 (exact 1 + 15 x 2^-26 -> fp32 1.0000002384 = 1 + 2^-22 if summed exactly)", {}, {}, 0.f, 1.0 + 15.0 * ldexp(1.0, -26)};
            e.a[0] = 1.f; e.b[0] = 1.f; for (int k = 1; k < 16; ++k) { e.a[k] = ldexpf(1.f, -13); e.b[k] = ldexpf(1.f, -13); } cases.push_back(e);
            Case f{"C = 1, products 8 x 2^-25 in k < 8 and 8 x 2^-25 in k >= 8 (exact 1 + 2^-21)", {}, {}, 1.f, 1.0 + 16.0 * ldexp(1.0, -25)};
            for (int k = 0; k < 16; ++k) { f.a[k] = ldexpf(1.f, -12); f.b[k] = ldexpf(1.f, -13); } cases.push_back(f);
            Case g{"C = -2^24 + products 2^24 + 1 + 2^-8 (cancellation: exact 1.00390625)", {}, {}, -16777216.f, 1.00390625};
            g.a[0] = 4096.f; g.b[0] = 4096.f; g.a[1] = 1.f; g.b[1] = 1.f; g.a[2] = 0.0625f; g.b[2] = 0.0625f; cases.push_back(g);
            Case h{"C = 0, products 2^20, 1.5, -2^20 (exact 1.5)", {}, {}, 0.f, 1.5};
            h.a[0] = 1024.f; h.b[0] = 1024.f; h.a[1] = 1.5f; h.b[1] = 1.f; h.a[2] = -1024.f; h.b[2] = 1024.f; cases.push_back(h);
            Case i{"C = 0, products 2^30, 1.5, -2^30 (exact 1.5; a 24-bit aligned adder would lose it)", {}, {}, 0.f, 1.5};
            i.a[0] = 32768.f; i.b[0] = 32768.f; i.a[1] = 1.5f; i.b[1] = 1.f; i.a[2] = -32768.f; i.b[2] = 32768.f; cases.push_back(i);
            Case j{"C = 0, products 2^30 (k=0), 1.5 (k=8), -2^30 (k=1) (same across the k halves)", {}, {}, 0.f, 1.5};
            j.a[0] = 32768.f; j.b[0] = 32768.f; j.a[8] = 1.5f; j.b[8] = 1.f; j.a[1] = -32768.f; j.b[1] = 32768.f; cases.push_back(j);
        }
        for (auto& c : cases) {
            unsigned short ha[16], hb[16];
            for (int k = 0; k < 16; ++k) { ha[k] = bf(c.a[k]); hb[k] = bf(c.b[k]); }
            CK(hipMemcpy(da, ha, 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb, 32, hipMemcpyHostToDevice));
            CK(hipMemcpy(dc, &c.c, 4, hipMemcpyHostToDevice));
            probe_kernel<<<1, 64>>>(da, db, dc, dd);
            float d;
            CK(hipMemcpy(&d, dd, 4, hipMemcpyDeviceToHost));
            printf("  %-100s -> %.10g (exact %.10g, fp32(exact) %.10g)\n", c.name, (double)d, c.exact, (double)(float)c.exact);
        }
    }

    struct Shape { const char* name; int M, N, K; };
    std::vector<Shape> acc_shapes = {{"acc.uniform", 256, 512, 1024}, {"acc.k4096", 256, 512, 4096}};
    std::vector<Shape> speed_shapes = {{"stride test 16000x4096x1056", 16000, 4096, 1056}, {"stride test 16000x4096x992", 16000, 4096, 992},
                                       {"stride test 8000x3072x1056", 8000, 3072, 1056}, {"stride test 9056x2048x544", 9056, 2048, 544},
                                       {"dec.ffn1 16000x4096x1024", 16000, 4096, 1024}, {"bt.in_proj 8000x3072x1024", 8000, 3072, 1024},
                                       {"dec.qkv 16000x2304x1024", 16000, 2304, 1024},  {"dec.w2 16000x1024x4096", 16000, 1024, 4096},
                                       {"bt.lin2 8000x1024x2048", 8000, 1024, 2048},     {"bt.o 8000x1024x1024", 8000, 1024, 1024},
                                       {"mimi.lin1 9056x2048x512", 9056, 2048, 512},     {"mimi.lin2 9056x512x2048", 9056, 512, 2048},
                                       {"mimi.in 9056x1536x512", 9056, 1536, 512},       {"mimi.out 9056x512x512", 9056, 512, 512},
                                       {"cal 8192x4096x4096", 8192, 4096, 4096}};

    auto run = [&](const Shape& s, int dist, bool check, int reps) {
        const long long M = s.M, N = s.N, K = s.K;
        std::vector<float> hx(M * K), hw(N * K), hb(N);
        std::mt19937_64 rng(1234 + M + N + K + dist);
        std::uniform_real_distribution<float> uni(-1.f, 1.f);
        std::normal_distribution<float> nrm(0.f, 1.f);
        std::uniform_int_distribution<int> ex(-12, 12);
        const bool zero = getenv("SG_ZERO") != nullptr;
        for (auto& v : hx) v = dist == 0 ? uni(rng) : nrm(rng) * ldexpf(1.f, ex(rng));
        if (zero) { for (auto& v : hx) v = 0.f; }
        for (auto& v : hw) v = dist == 0 ? uni(rng) : nrm(rng) * 0.05f * ldexpf(1.f, ex(rng) / 2);
        for (auto& v : hb) v = uni(rng);
        if (zero) { for (auto& v : hw) v = 0.f; }
        float *dx, *dw, *db, *dy;
        u32x4* dwp;
        CK(hipMalloc(&dx, M * K * 4)); CK(hipMalloc(&dw, N * K * 4)); CK(hipMalloc(&db, N * 4)); CK(hipMalloc(&dy, M * N * 4));
        CK(hipMalloc(&dwp, N * K * 6));
        CK(hipMemcpy(dx, hx.data(), M * K * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dw, hw.data(), N * K * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
        const long long ng = N * K / 8;
        prepack_kernel<<<(unsigned)((ng + 255) / 256), 256>>>(dw, dwp, ng);
        CK(hipGetLastError());
        const unsigned tiles = (unsigned)(((M + 127) / 128) * ((N + 127) / 128));
        const bool dma = !getenv("SG_V1");
        const bool pipe = !getenv("SG_V1") && !getenv("SG_V2");
        auto launch = [&]() {
            if (pipe) split_gemm_pipe_kernel<128, 128><<<tiles, 256>>>(dx, dwp, db, dy, (int)M, (int)N, (int)K, 8);
            else if (dma) split_gemm_dma_kernel<128, 128><<<tiles, 256>>>(dx, dwp, db, dy, (int)M, (int)N, (int)K, 8);
            else split_gemm_kernel<128, 128><<<tiles, 256>>>(dx, dwp, db, dy, (int)M, (int)N, (int)K, 8);
        };
        launch();
        CK(hipDeviceSynchronize());
        if (check) {
            std::vector<float> hy(M * N);
            CK(hipMemcpy(hy.data(), dy, M * N * 4, hipMemcpyDeviceToHost));
            // references on sampled rows: fp64, the fp32 fmaf chain in k order (what v_mfma_f32_32x32x2_f32 computes, bitwise), and the
            // 9-product sum evaluated in fp64 (must equal the fp64 reference: the split is exact)
            double e_split = 0, e_chain = 0, e_split_max = 0, e_chain_max = 0, den = 0, split_exact_dev = 0;
            double rel_rms_s = 0, rel_rms_c = 0, ref_rms = 0;
            long long cnt = 0;
            for (long long m = 0; m < M; m += 4) {
                for (long long n = 0; n < N; ++n) {
                    double r = hb[n], sabs = 0, r9 = hb[n];
                    float c = 0.f;
                    for (long long k = 0; k < K; ++k) {
                        const float a = hx[m * K + k], b = hw[n * K + k];
                        r += (double)a * b;
                        sabs += fabs((double)a * b);
                        c = fmaf(a, b, c);
                        if (n < 8) {
                            float ah, am, al, bh, bm, bl;
                            host_split(a, ah, am, al); host_split(b, bh, bm, bl);
                            r9 += ((double)ah + am + al) * ((double)bh + bm + bl);
                        }
                    }
                    c += hb[n];
                    if (n < 8) split_exact_dev = fmax(split_exact_dev, fabs(r9 - r));
                    const double es = fabs((double)hy[m * N + n] - r) / sabs, ec = fabs((double)c - r) / sabs;
                    e_split += es; e_chain += ec;
                    e_split_max = fmax(e_split_max, es); e_chain_max = fmax(e_chain_max, ec);
                    rel_rms_s += ((double)hy[m * N + n] - r) * ((double)hy[m * N + n] - r);
                    rel_rms_c += ((double)c - r) * ((double)c - r);
                    ref_rms += r * r;
                    ++cnt;
                }
            }
            (void)den;
            printf("  %-14s dist=%d M=%lld N=%lld K=%lld : |err| / sum|a b|  split9 mean %.3e max %.3e | fp32 chain mean %.3e max %.3e | rel RMS split9 %.3e chain %.3e | "
                   "fp64(9 products) - fp64 = %.1e\n",
                   s.name, dist, M, N, K, e_split / cnt, e_split_max, e_chain / cnt, e_chain_max, sqrt(rel_rms_s / ref_rms), sqrt(rel_rms_c / ref_rms), split_exact_dev);
            // determinism: three launches, same bits
            std::vector<float> hy2(M * N);
            bool same = true;
            for (int t = 0; t < 2; ++t) {
                launch();
                CK(hipMemcpy(hy2.data(), dy, M * N * 4, hipMemcpyDeviceToHost));
                same = same && !memcmp(hy.data(), hy2.data(), M * N * 4);
            }
            printf("  %-14s three launches bit-identical: %s\n", s.name, same ? "yes" : "NO");
        } else {
            for (int i = 0; i < 2; ++i) launch();
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            const double ms = now_ms(e0, e1) / reps;
            unsigned long long hc[2] = {0, 0}, z[2] = {0, 0};
            CK(hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), 16));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, 16));
            printf("  %-28s %8.1f us  %7.1f TFLOP/s (fp32-equivalent 2MNK)   tiles %u   shader clock in the main loop %.2f GHz, %.0f cycles per tile-chunk-workgroup\n", s.name, ms * 1e3,
                   2.0 * M * N * K / ms / 1e9, tiles, hc[1] ? (double)hc[0] / (double)hc[1] * 0.1 : 0.0, hc[0] / ((reps + 2.0) * tiles * (K / 16.0)));
        }
        CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(db)); CK(hipFree(dy)); CK(hipFree(dwp));
    };
    if (!speed_only) {
        printf("== numerics: 9 bf16 plane products vs the fp32 fma chain, both against fp64 (rows sampled every 4th)\n");
        for (auto& s : acc_shapes)
            for (int dist = 0; dist < 2; ++dist) run(s, dist, true, 0);
    }
    printf("== speed: %s<128,128>, SG_STORE_AT=%d SG_VARIANT=%d\n", getenv("SG_V1") ? "split_gemm_kernel (register staging)" : getenv("SG_V2") ? "split_gemm_dma_kernel" : "split_gemm_pipe_kernel", SG_STORE_AT, SG_VARIANT);
    for (auto& s : speed_shapes) run(s, 0, false, 10);
    return 0;
}
