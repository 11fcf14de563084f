// lstm.hip - the nn.LSTM(d, d, 1, batch_first=True) that every codec transformer layer runs before its QKV
// projection (reference: QuarkAudio-HCodec/HCodec-1.0/vq/encoder_modules/transformer.py:115,133).
//
// The input half (x W_ih^T + b_ih + b_hh) is one big batched GEMM done by conv_gemm; what remains is the strictly
// sequential recurrence  gates_t = xw_t + h_{t-1} W_hh^T.  Default structure (d < 1536): one launch per time step - every step is an
// all-to-all seam (each gate needs the whole h_{t-1} of its batch row), the kernel boundary IS the cross-CU synchronisation and at
// ~1.2-1.9 us (MI355X_MICROARCH price list, "boundary") it is cheaper than an in-launch exchange of a 64-128 KB state across
// 256 CUs ("barrier-xcd" 4.1 us + the fresh read) as long as the per-step stream of W_hh is short (d <= 1024: measured below).
// What the step must not do is serialise its memory round trips: every load of a wave's K
// share is issued before its first MFMA (NI template), and the T launches of a call are replayed from a cached hipGraph so the
// host never limits the 3-4 us step.  d/4 workgroups per step.  A workgroup owns 4 hidden units x 4 gates = 16 rows of W_hh (rows
// pre-permuted to (unit, gate) order at load time) for ALL batch rows, splits K = d over its 8 waves, runs
// v_mfma_f32_16x16x4_f32 with batch as the M dimension, reduces the 8 partial tiles through LDS and applies the
// cell update in the same kernel, so gates never touch HBM.
//
// XCD-local variant (lstm_xcd_kernel, further down; default for d = 512 / 768, QA_LSTM_XCD): every XCD keeps a copy of W_hh in the
// registers of its 32 CUs and runs its share of the batch rows alone - 2.3 / 2.8 us per step against 6.2 / 7.5 us.
//
// Persistent variant (lstm_persistent_kernel, measured in tools/micro/lstm_persistent.hip: 14.7 -> 8.7 us per step at d = 1536 / B = 16,
// 10.05 -> 9.1 at d = 1024 / B = 32, equal at 768, slower at 512): ONE launch for all T steps, d / U workgroups (<= CU count, all
// co-resident), each holding its 4U rows of W_hh in its waves' REGISTERS for the whole call; h_t crosses CUs through write-through
// (sc1) stores and sc1 loads (no fences: MI355X_MICROARCH.md "Valid forms"), every step ends in an XCD-hierarchical counter
// barrier with BOUNDED spins.  What it removes is the per-step weight stream (37.7 MB at d = 1536), what it adds is the in-launch
// barrier (~4 us) - so it is the default only where the stream dominates (d >= 1536: H-Codec 2.0); QA_LSTM_PERSISTENT=0 / 1 forces
// it off / on for every supported width.  It needs every workgroup resident at once: two such kernels sharing the device (two
// handles driven concurrently on two streams) starve each other until the spin bound trips; that sets an error word in pinned host
// memory.  The model graph that launched the kernel waits for its stream before returning, reads the word (lstm_persistent_collect) and
// re-runs the call on the per-step kernels (hcodec.cpp run_graph_checked), so the failing call itself returns valid results; the device
// then stops choosing the persistent kernel by itself.  Run concurrent handles with QA_LSTM_PERSISTENT=0 to avoid the time-out altogether.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "kernels.h"

namespace qa {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// MT = number of 16-row batch tiles (B <= 16*MT); NI = 16-wide K steps per wave (d / 128) when known at compile time, 0 = loop
template <int MT, int NI>
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
        if (NI > 0) {  // all of this wave's W_hh and h loads in flight before the first MFMA: one L2 round trip per step
            float4 wv[NI > 0 ? NI : 1], hv[MT][NI > 0 ? NI : 1];
#pragma unroll
            for (int i = 0; i < NI; ++i) wv[i] = *reinterpret_cast<const float4*>(wrow + i * 16);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int i = 0; i < NI; ++i) hv[m][i] = *reinterpret_cast<const float4*>(hrow[m] + i * 16);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[m][i].x, wv[i].x, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[m][i].y, wv[i].y, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[m][i].z, wv[i].z, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[m][i].w, wv[i].w, acc[m], 0, 0, 0);
                }
        } else {
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

template <int MT>
static void launch_step(int ni, dim3 grid, hipStream_t s, const float* xw, const float* w, float* h, float* c, int bn, int T, int d, int t) {
#define QA_LS(NI) hipLaunchKernelGGL((lstm_step_kernel<MT, NI>), grid, dim3(512), 0, s, xw, w, h, c, bn, T, d, t)
    switch (ni) {
        case 1: QA_LS(1); break;
        case 2: QA_LS(2); break;
        case 4: QA_LS(4); break;    // d = 512  (H-Codec encoder)
        case 6: QA_LS(6); break;    // d = 768  (H-Codec 1.0 decoder)
        case 8: QA_LS(8); break;    // d = 1024 (H-Codec 1.5 decoder)
        case 12: QA_LS(12); break;  // d = 1536 (H-Codec 2.0)
        default: QA_LS(0); break;
    }
#undef QA_LS
}

// One chain of T dependent step launches for batch rows [0, bn) of the given buffers.
static void lstm_chain(const float* xw_b, const float* w_hh_ug, float* h_b, float* c_b, int bn, int T, int d, int t, hipStream_t s) {
    const int ni = (d % 128 == 0) ? d / 128 : 0;
    const int mt = (int)ceil_div(bn, 16);
    const dim3 grid(d / 4);
    switch (mt) {
        case 1: launch_step<1>(ni, grid, s, xw_b, w_hh_ug, h_b, c_b, bn, T, d, t); break;
        case 2: launch_step<2>(ni, grid, s, xw_b, w_hh_ug, h_b, c_b, bn, T, d, t); break;
        case 3: launch_step<3>(ni, grid, s, xw_b, w_hh_ug, h_b, c_b, bn, T, d, t); break;
        default: launch_step<4>(ni, grid, s, xw_b, w_hh_ug, h_b, c_b, bn, T, d, t); break;
    }
}

// The step chains of a call: batches above 64 rows as consecutive chains of 64 (MT <= 4).  (r02: two concurrent half-batch chains on two
// streams were measured and lost, 147.4 vs 144.9 ms per H-Codec 1.5 step - DESIGN.md "measured and rejected".)
static int lstm_launch_steps(const float* xw, const float* w_hh_ug, float* h_out, float* c_state, int B, int T, int d, hipStream_t s) {
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int bn = std::min(64, B - b0);
        for (int t = 0; t < T; ++t)
            lstm_chain(xw + (long long)b0 * T * 4 * d, w_hh_ug, h_out + (long long)b0 * T * d, c_state + (long long)b0 * d, bn, T, d, t, s);
        QA_LAUNCH_CHECK();
    }
    return QA_OK;
}

// ------------------------------------------------------------------------------------------------ persistent recurrence
namespace {
// sync words, one per 128-byte line: grp_cnt[8] | top_cnt | top_gen | grp_gen[8] | err
enum { SY_GRP_CNT = 0, SY_TOP_CNT = 8, SY_TOP_GEN = 9, SY_GRP_GEN = 10, SY_ERR = 18, SY_WORDS = 19, SY_STRIDE = 32 };
// spin bound of the barrier polls: QA_LSTM_SPIN_LIMIT, default 2^21 x (s_sleep + one L2 round trip) ~ seconds, then the barrier is declared broken
constexpr int LSTM_SYNC_RING = 64;  // launches in flight on one device before a sync block is re-armed (several handles share the ring)
}  // namespace

#define QA_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ bool lstm_spin_until(unsigned* word, unsigned want, unsigned* err, unsigned* err_host, unsigned limit) {
    for (unsigned spins = 0; spins < limit; ++spins) {
        if (__hip_atomic_load(word, QA_RLX) >= want) return true;
        if ((spins & 1023u) == 1023u && __hip_atomic_load(err, QA_RLX) != 0u) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store(err, 1u, QA_RLX);
    __hip_atomic_store(err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return false;
}

// 16-byte load that bypasses this CU's L1 (L2 / memory-side served).  A relaxed agent-scope __hip_atomic_load stops at 8 bytes and
// hipcc waits for each one before issuing the next; an asm load is invisible to its scoreboard, so the caller waits by hand
// (asm "s_waitcnt vmcnt(0)" with the destination as an in/out operand, which also pins every use behind the wait).
__device__ __forceinline__ f32x4 lstm_load_sc1_b128(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// one lane per workgroup; epoch = steps completed (1-based).  The last arriver of a group (blockIdx % ngrp: the XCD a workgroup is
// observed to land on - a speed assumption only) forwards to the top counter, the last group publishes top_gen, every group's
// forwarder republishes it as its group's generation word, which is what the group's pollers read.
__device__ __forceinline__ void lstm_barrier_arrive(unsigned* sy, int g, unsigned epoch, unsigned per_grp, unsigned ngrp, unsigned* err_host,
                                                    unsigned limit) {
    const unsigned old = __hip_atomic_fetch_add(sy + (SY_GRP_CNT + g) * SY_STRIDE, 1u, QA_RLX);
    if (old + 1u == per_grp * epoch) {
        const unsigned o2 = __hip_atomic_fetch_add(sy + SY_TOP_CNT * SY_STRIDE, 1u, QA_RLX);
        if (o2 + 1u == ngrp * epoch) __hip_atomic_store(sy + SY_TOP_GEN * SY_STRIDE, epoch, QA_RLX);
        else if (!lstm_spin_until(sy + SY_TOP_GEN * SY_STRIDE, epoch, sy + SY_ERR * SY_STRIDE, err_host, limit)) return;
        __hip_atomic_store(sy + (SY_GRP_GEN + g) * SY_STRIDE, epoch, QA_RLX);
    }
}

// MT batch tiles of 16 rows, NT column tiles of 16 W_hh rows (a workgroup owns R = 4 U <= 16 NT rows), NI = d / 128 K steps per wave
template <int MT, int NT, int NI>
__global__ __launch_bounds__(512) void lstm_persistent_kernel(const float* __restrict__ xw, const float* __restrict__ w_hh, float* h_out,
                                                              float* __restrict__ c_state, int B, int T, int d, int U, unsigned* sy,
                                                              int ngrp, int per_grp, unsigned* err_host, unsigned spin_limit) {
    __shared__ float part[4][MT * NT][16][17];
    __shared__ int s_ok;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int R = 4 * U, n0 = blockIdx.x * R;
    const int grp = blockIdx.x % ngrp;
    const int kw = d / 8, k0 = wave * kw;
    // this wave's share of the workgroup's W_hh rows, resident in registers for all T steps
    float4 wv[NT][NI];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int r = min(n * 16 + li, R - 1);  // columns past R repeat the last row and are never read back
        const float* wrow = w_hh + (long long)(n0 + r) * d + k0 + 4 * kq;
#pragma unroll
        for (int i = 0; i < NI; ++i) wv[n][i] = *reinterpret_cast<const float4*>(wrow + i * 16);
    }
    const bool epi = tid < B * U;
    const int eb = epi ? tid / U : 0, eu = epi ? tid - eb * U : 0;
    const int unit = blockIdx.x * U + eu;
    float c_reg = 0.f;
    for (int t = 0; t < T; ++t) {
        float4 xg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (epi) xg = *reinterpret_cast<const float4*>(xw + ((long long)eb * T + t) * 4 * d + (long long)unit * 4);
        f32x4 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (t > 0) {
            if (tid == 0) s_ok = lstm_spin_until(sy + (SY_GRP_GEN + grp) * SY_STRIDE, (unsigned)t, sy + SY_ERR * SY_STRIDE, err_host, spin_limit) ? 1 : 0;
            __syncthreads();
            if (!s_ok) break;  // uniform: a broken barrier ends the call for everybody (the error words are set)
            asm volatile("" ::: "memory");
            // h_{t-1}: written by sc1 (write-through) stores on other CUs, read with sc1 loads, every load in flight at once
            f32x4 hv[MT][NI];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                int b = m * 16 + li;
                if (b >= B) b = B - 1;
                const float* hrow = h_out + ((long long)b * T + (t - 1)) * d + k0 + 4 * kq;
#pragma unroll
                for (int i = 0; i < NI; ++i) hv[m][i] = lstm_load_sc1_b128(hrow + i * 16);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int i = 0; i < NI; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv[m][i])::"memory");
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[m][i].x, wv[n][i].x, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[m][i].y, wv[n][i].y, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[m][i].z, wv[n][i].z, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[m][i].w, wv[n][i].w, acc[m][n], 0, 0, 0);
                    }
        }
        // K-split reduction in two rounds (4 slots of LDS): waves 4..7 park their tiles, waves 0..3 add them to their own
        if (wave >= 4) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) part[wave - 4][m * NT + n][4 * kq + r][li] = acc[m][n][r];
        }
        __syncthreads();
        if (wave < 4) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) part[wave][m * NT + n][4 * kq + r][li] += acc[m][n][r];
        }
        __syncthreads();
        if (epi) {
            float g4[4] = {xg.x, xg.y, xg.z, xg.w};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = eu * 4 + g;
                float s = g4[g];
#pragma unroll
                for (int w = 0; w < 4; ++w) s += part[w][(eb >> 4) * NT + (col >> 4)][eb & 15][col & 15];
                g4[g] = s;
            }
            const float ig = sigmoid_f(g4[0]), fg = sigmoid_f(g4[1]), gg = tanhf(g4[2]), og = sigmoid_f(g4[3]);
            c_reg = fg * c_reg + ig * gg;
            // write-through store: the value has left this XCD's L2 once vmcnt drains
            __hip_atomic_store(reinterpret_cast<unsigned*>(h_out + ((long long)eb * T + t) * d + unit), __float_as_uint(og * tanhf(c_reg)), QA_RLX);
        }
        if (t + 1 < T) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the workgroup reports the step done
            __syncthreads();
            if (tid == 0) lstm_barrier_arrive(sy, grp, (unsigned)(t + 1), (unsigned)per_grp, (unsigned)ngrp, err_host, spin_limit);
        }
    }
    if (epi) c_state[(long long)eb * d + unit] = c_reg;
}

// ------------------------------------------------------------------------------------------------ XCD-local recurrence (QA_LSTM_XCD)
// The batch rows of an LSTM are independent recurrences and a d <= 768 W_hh (4.2 / 9.4 MB) fits the REGISTERS of the 32 CUs of ONE
// XCD.  So: one launch for all T steps, one workgroup per CU, and the workgroups that land on XCD x (s_getreg HW_REG_XCC_ID - read
// from the hardware, never assumed from blockIdx) form TEAM x.  Every team keeps its own copy of the whole W_hh in registers (a
// workgroup: U = d / 32 hidden units = 4 U gate rows) and runs the sequences b = x, x + 8, x + 16, ... of the call all by itself: no
// exchange between XCDs at all, a step's h_t travels through the team's own L2, and the per-step barrier has 32 participants on one
// counter in that L2 (~1 us) instead of 256 over the fabric (~4 us).  Matrix work on v_mfma_f32_4x4x1_16b_f32: block = hidden unit
// (A: lane & 3 = gate row of the unit, resident weights), B: lane & 3 = sequence of the team (h_{t-1} from LDS); lane (unit, q) ends
// with the unit's 4 gate pre-activations of sequence q.  K is split over the waves; partials meet in LDS in a fixed order
// (deterministic; a row's arithmetic does not depend on the batch).
//   QA_LSTM_XCD = 1: agent-scope forms (sc1 write-through h stores, agent atomics) - valid wherever the workgroups sit;
//   QA_LSTM_XCD = 2: XCD-local forms (plain h stores that stay in the team's L2, workgroup-scope atomic executed in that L2) -
//                    relies on what defines a team: its members share one L2.  Loads of h / the counter bypass L1 (sc1) in both.
// Needs 32 resident workgroups per XCD: a surplus or missing member trips the bounded spins -> error word -> the calling model
// graph re-runs the call on the per-step kernels (run_graph_checked), exactly like lstm_persistent_kernel.
enum { SX_SLOT = 0, SX_CNT = 8 };  // sync lines: team slot counters [8], team arrive counters [8], SY_ERR

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

template <int D, int NW>
__global__ __launch_bounds__(NW * 64, 1) void lstm_xcd_kernel(const float* __restrict__ xw, const float* __restrict__ w_hh, float* h_out,
                                                               float* __restrict__ c_state, int B, int T, unsigned* sy, int per_team,
                                                               int n_teams, unsigned* err_host, unsigned spin_limit, int fast) {
    constexpr int U = D / 32, R = 4 * U, RG = (R + 63) / 64, KS = NW / RG, KW = D / KS, NQ = 4, LDH = D + 4;
    static_assert(NW % RG == 0 && D % KS == 0 && KW % 4 == 0 && NQ * D / 4 == NW * 64, "lstm_xcd: shape does not tile");
    __shared__ __attribute__((aligned(16))) float s_h[NQ][LDH];           // h_{t-1} of the team's sequences (rows 16 B apart in bank phase)
    __shared__ __attribute__((aligned(16))) float s_part[KS][RG * 64][4];  // [K slice][unit * 4 + q][gate]
    __shared__ int s_ctl[3];
    extern __shared__ float s_pad[];  // unused: sized by the host so that ONE workgroup fits a CU
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rg = wave % RG, ks = wave / RG;
    if (tid == 0) {
        const unsigned x = xcc_id();
        s_ctl[0] = (int)x;
        s_ctl[1] = x < 8u ? (int)__hip_atomic_fetch_add(sy + (SX_SLOT + x) * SY_STRIDE, 1u, QA_RLX) : per_team;
    }
    __syncthreads();
    const int team = s_ctl[0], slot = s_ctl[1];
    unsigned* err = sy + SY_ERR * SY_STRIDE;
    if (slot >= per_team || team >= n_teams) {  // a 33rd workgroup on this XCD (so another team is short of one): nobody can finish
        if (tid == 0) {
            __hip_atomic_store(err, 1u, QA_RLX);
            __hip_atomic_store(err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const int nq = min(NQ, (B - team + n_teams - 1) / n_teams);  // this team's sequences: b = team + n_teams * q
    if (nq <= 0) return;
    unsigned* cnt = sy + (SX_CNT + team) * SY_STRIDE;

    // resident weights: lane = gate row (unit-major, gate-minor: the load-time permutation) of row group rg, K slice ks
    const int lrow = rg * 64 + lane;
    float wreg[KW];
    {
        const float* wrow = w_hh + ((long long)slot * R + min(lrow, R - 1)) * D + ks * KW;
#pragma unroll
        for (int j = 0; j < KW / 4; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(wrow + 4 * j);
            wreg[4 * j] = v.x; wreg[4 * j + 1] = v.y; wreg[4 * j + 2] = v.z; wreg[4 * j + 3] = v.w;
        }
    }
    // epilogue lane (waves of K slice 0): hidden unit of the team slot, sequence q
    const int q = lane & 3, ul = rg * 16 + (lane >> 2);
    const bool epi = ks == 0 && ul < U && q < nq;
    const int unit = slot * U + min(ul, U - 1);
    const int eb = team + n_teams * min(q, nq - 1);
    // staging share of this thread: one float4 of h_{t-1}
    const int sq = tid / (D / 4), sc4 = tid % (D / 4);
    const int sb = team + n_teams * min(sq, nq - 1);
    float c_reg = 0.f;
    for (int t = 0; t < T; ++t) {
        float4 xg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (epi) xg = *reinterpret_cast<const float4*>(xw + ((long long)eb * T + t) * 4 * D + (long long)unit * 4);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (t > 0) {
            if (tid == 0) s_ctl[2] = lstm_spin_until(cnt, (unsigned)(per_team * t), err, err_host, spin_limit) ? 1 : 0;
            __syncthreads();
            if (!s_ctl[2]) break;  // uniform: a broken barrier ends the call for the whole workgroup (the error words are set)
            asm volatile("" ::: "memory");
            f32x4 hv = lstm_load_sc1_b128(h_out + ((long long)sb * T + (t - 1)) * D + 4 * sc4);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv)::"memory");
            *reinterpret_cast<f32x4*>(&s_h[sq][4 * sc4]) = hv;
            __syncthreads();
            const float* hq = &s_h[q][ks * KW];
#pragma unroll
            for (int j = 0; j < KW / 4; ++j) {
                const float4 hb = *reinterpret_cast<const float4*>(hq + 4 * j);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * j], hb.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * j + 1], hb.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * j + 2], hb.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * j + 3], hb.w, acc, 0, 0, 0);
            }
            *reinterpret_cast<f32x4*>(&s_part[ks][rg * 64 + lane][0]) = acc;
            __syncthreads();
        }
        if (epi) {
            f32x4 g = {xg.x, xg.y, xg.z, xg.w};
            if (t > 0) {
#pragma unroll
                for (int k = 0; k < KS; ++k) g += *reinterpret_cast<const f32x4*>(&s_part[k][rg * 64 + lane][0]);
            }
            const float ig = sigmoid_f(g[0]), fg = sigmoid_f(g[1]), gg = tanhf(g[2]), og = sigmoid_f(g[3]);
            c_reg = fg * c_reg + ig * gg;
            unsigned* dst = reinterpret_cast<unsigned*>(h_out + ((long long)eb * T + t) * D + unit);
            const unsigned hbits = __float_as_uint(og * tanhf(c_reg));
            if (fast) __hip_atomic_store(dst, hbits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // plain store: stays in this XCD's L2
            else __hip_atomic_store(dst, hbits, QA_RLX);                                               // sc1: write-through
        }
        if (t + 1 < T) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the storing waves drain before the workgroup reports the step done
            __syncthreads();
            if (tid == 0) {
                if (fast) (void)__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else (void)__hip_atomic_fetch_add(cnt, 1u, QA_RLX);
            }
        }
    }
    if (epi) c_state[(long long)eb * D + unit] = c_reg;
}

// ------------------------------------------------------------------------------------------------ team recurrence for wider layers
// QA_LSTM_TEAM (default 1 since r04: parity green on MI355X, 5.4 us per step against 9.7 for the per-step kernel at d = 1024,
// profiles/r04_lstm_team_ab.txt).  The XCD-local idea for widths whose W_hh does not fit ONE XCD's registers: a team is PT
// workgroups (64 at d = 1024: two XCDs' worth of register files per copy of W_hh, four teams; the 128-workgroup / two-team form for d = 1536 lost to lstm_persistent_kernel in r05), team =
// blockIdx % n_teams, slot = blockIdx / n_teams - nothing depends on where a workgroup lands, because every hand-off uses the
// agent-scope forms (sc1 write-through stores, drained vmcnt, agent atomic on the team's counter, sc1 loads).  A team runs
// 4 SG sequences (two MFMA chains per wave at SG = 2: the resident A operand is used twice), so B = 32 at d = 1024 is ONE launch.
// Per step: one PT-member counter + 4 SG x d x 4 bytes of h per workgroup + KW x SG 4x4x1 MFMAs per wave (MFMA floor at
// d = 1024 / B = 32: 2 us per step; the per-step kernel takes 9.7 us).
template <int D, int NW, int PT, int SG>
__global__ __launch_bounds__(NW * 64, 1) void lstm_team_kernel(const float* __restrict__ xw, const float* __restrict__ w_hh, float* h_out,
                                                                float* __restrict__ c_state, int B, int T, unsigned* sy, int n_teams, int fault,
                                                                unsigned* err_host, unsigned spin_limit) {
    constexpr int U = D / PT, R = 4 * U, KS = NW, KW = D / KS, NQ = 4 * SG, LDH = D + 4, F4 = NQ * D / 4 / (NW * 64);
    static_assert(R <= 64 && D % PT == 0 && D % KS == 0 && KW % 4 == 0 && (NQ * D / 4) % (NW * 64) == 0, "lstm_team: shape does not tile");
    __shared__ __attribute__((aligned(16))) float s_part[KS][64][SG][4];  // [K slice][unit * 4 + q][sequence group][gate]
    __shared__ int s_ok;
    // h_{t-1} of the team's sequences lives in the DYNAMIC segment (49 KB at d = 1536: static LDS stops at 64 KB); the host sizes that
    // segment beyond it so that ONE workgroup fits a CU
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    float (*s_h)[LDH] = reinterpret_cast<float (*)[LDH]>(s_dyn);
    const int tid = threadIdx.x, lane = tid & 63, ks = tid >> 6;
    const int team = blockIdx.x % n_teams, slot = blockIdx.x / n_teams;
    unsigned* err = sy + SY_ERR * SY_STRIDE;
    const int nq = min(NQ, (B - team + n_teams - 1) / n_teams);  // this team's sequences: b = team + n_teams * q
    if (nq <= 0) return;
    unsigned* cnt = sy + (SX_CNT + team) * SY_STRIDE;
    const unsigned members = (unsigned)(PT + fault);  // QA_LSTM_FAULT: wait for a member that does not exist

    float wreg[KW];
    {
        const float* wrow = w_hh + ((long long)slot * R + min(lane, R - 1)) * D + ks * KW;
#pragma unroll
        for (int j = 0; j < KW / 4; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(wrow + 4 * j);
            wreg[4 * j] = v.x; wreg[4 * j + 1] = v.y; wreg[4 * j + 2] = v.z; wreg[4 * j + 3] = v.w;
        }
    }
    const int q = lane & 3, ul = lane >> 2;
    const bool epi_wave = ks == 0 && ul < U;
    const int unit = slot * U + min(ul, U - 1);
    float c_reg[SG];
#pragma unroll
    for (int g = 0; g < SG; ++g) c_reg[g] = 0.f;
    for (int t = 0; t < T; ++t) {
        float4 xg[SG];
#pragma unroll
        for (int g = 0; g < SG; ++g) {
            xg[g] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (epi_wave && q + 4 * g < nq)
                xg[g] = *reinterpret_cast<const float4*>(xw + ((long long)(team + n_teams * (q + 4 * g)) * T + t) * 4 * D + (long long)unit * 4);
        }
        f32x4 acc[SG];
#pragma unroll
        for (int g = 0; g < SG; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (t > 0) {
            if (tid == 0) s_ok = lstm_spin_until(cnt, members * (unsigned)t, err, err_host, spin_limit) ? 1 : 0;
            __syncthreads();
            if (!s_ok) break;
            asm volatile("" ::: "memory");
            {
                f32x4 hv[F4];
#pragma unroll
                for (int i = 0; i < F4; ++i) {
                    const int f = tid + i * NW * 64, sq = f / (D / 4), sc4 = f % (D / 4);
                    hv[i] = lstm_load_sc1_b128(h_out + ((long long)(team + n_teams * min(sq, nq - 1)) * T + (t - 1)) * D + 4 * sc4);
                }
#pragma unroll
                for (int i = 0; i < F4; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv[i])::"memory");
#pragma unroll
                for (int i = 0; i < F4; ++i) {
                    const int f = tid + i * NW * 64;
                    *reinterpret_cast<f32x4*>(&s_h[f / (D / 4)][4 * (f % (D / 4))]) = hv[i];
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < KW / 4; ++j) {
#pragma unroll
                for (int g = 0; g < SG; ++g) {
                    const float4 hb = *reinterpret_cast<const float4*>(&s_h[q + 4 * g][ks * KW + 4 * j]);
                    acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * j], hb.x, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * j + 1], hb.y, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * j + 2], hb.z, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * j + 3], hb.w, acc[g], 0, 0, 0);
                }
            }
#pragma unroll
            for (int g = 0; g < SG; ++g) *reinterpret_cast<f32x4*>(&s_part[ks][lane][g][0]) = acc[g];
            __syncthreads();
        }
        if (epi_wave) {
#pragma unroll
            for (int g = 0; g < SG; ++g) {
                if (q + 4 * g >= nq) continue;
                f32x4 gt = {xg[g].x, xg[g].y, xg[g].z, xg[g].w};
                if (t > 0) {
#pragma unroll
                    for (int k = 0; k < KS; ++k) gt += *reinterpret_cast<const f32x4*>(&s_part[k][lane][g][0]);
                }
                const float ig = sigmoid_f(gt[0]), fg = sigmoid_f(gt[1]), gg = tanhf(gt[2]), og = sigmoid_f(gt[3]);
                c_reg[g] = fg * c_reg[g] + ig * gg;
                const int b = team + n_teams * (q + 4 * g);
                __hip_atomic_store(reinterpret_cast<unsigned*>(h_out + ((long long)b * T + t) * D + unit), __float_as_uint(og * tanhf(c_reg[g])), QA_RLX);
            }
        }
        if (t + 1 < T) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) (void)__hip_atomic_fetch_add(cnt, 1u, QA_RLX);
        }
    }
    if (epi_wave) {
#pragma unroll
        for (int g = 0; g < SG; ++g)
            if (q + 4 * g < nq) c_state[(long long)(team + n_teams * (q + 4 * g)) * D + unit] = c_reg[g];
    }
}

namespace {
// One model-graph call's view of the in-launch recurrences it issued (the error word and the launch counter used to be
// per DEVICE, so of two handles sharing a device the one that synchronised first collected - and cleared - the other's failure).
// A call takes a ticket (lstm_call_begin): its own mapped pinned error word out of the device's pool and its own launch count;
// every in-launch recurrence the calling thread issues until lstm_call_end reports into that word and nowhere else.
struct LstmCall {
    unsigned* err_host = nullptr;
    unsigned* err_dev = nullptr;
    unsigned launches = 0;  // in-launch recurrences this call issued
    unsigned pending = 0;   // ... of which not yet behind a host synchronisation of the call's stream
    bool failed = false;    // an observed synchronisation found the error word set
    bool fault_injected = false;  // QA_LSTM_FAULT was set when one of its recurrences was LAUNCHED (tests): its failure says nothing about the device
    bool counted = false;   // this call is one of LstmPersistentDev::inflight_calls (it has an in-launch recurrence behind no host sync yet)
    int dev = -1, slot = -1;
};
constexpr int LSTM_ERR_POOL = 64;  // error words (= concurrent model-graph calls) per device; further calls share the device word
struct LstmPersistentDev {
    unsigned* sync = nullptr;      // LSTM_SYNC_RING blocks of SY_WORDS * SY_STRIDE words (device)
    unsigned* err_host = nullptr;  // pinned, mapped: LSTM_ERR_POOL + 1 words; [LSTM_ERR_POOL] = the device word of ticket-less callers
    unsigned* err_dev = nullptr;   // device alias of err_host
    unsigned long long pool_busy = 0;  // bit i: pool word i belongs to a live call
    int cus = 0, next = 0;
    unsigned long long launches = 0;  // persistent launches so far on this device (diagnostics)
    bool degraded = false;            // a barrier timed out on this device: the auto mode stops choosing the persistent kernel
    // r05, co-residency ticket: model-graph calls that have launched an in-launch recurrence and not yet synchronised behind it.  The
    // in-launch kernels need (nearly) the whole device resident at once, so a launch that finds ANOTHER call's recurrence possibly still
    // running takes the per-step kernels at once (`diverted`) instead of starving that kernel's barrier and paying the 2^21-poll time-out
    // plus a re-run (VERDICT r04 item 8 / ADVICE r04).  Conservative: a call counts until its own host synchronisation.
    int inflight_calls = 0;
    unsigned long long diverted = 0;
};
LstmPersistentDev g_lstm_p[16];
std::mutex g_lstm_mu;  // guards the persistent-device table and the step-graph cache
thread_local bool t_lstm_per_step = false;  // lstm_force_per_step(): the re-run of a call whose persistent recurrence failed
thread_local LstmCall* t_lstm_call = nullptr;  // the calling thread's open model-graph call (lstm_call_begin .. lstm_call_end)

// -1 auto (widths whose per-step weight stream dominates: d >= 1536), 0 off, 1 on for every supported width
int lstm_persistent_mode() { return (int)knob(K_LSTM_PERSISTENT); }

// the error word a launch of the calling thread reports into, and the book-keeping of one launch
unsigned* lstm_err_word(LstmPersistentDev& P, int dev) {
    if (t_lstm_call && t_lstm_call->dev == dev && t_lstm_call->err_dev) return t_lstm_call->err_dev;
    return P.err_dev + LSTM_ERR_POOL;
}
void lstm_count_launch(LstmPersistentDev& P, int dev) {
    ++P.launches;
    if (t_lstm_call && t_lstm_call->dev == dev) {
        ++t_lstm_call->launches;
        ++t_lstm_call->pending;
        if (knob(K_LSTM_FAULT) != 0) t_lstm_call->fault_injected = true;
        if (!t_lstm_call->counted) {
            t_lstm_call->counted = true;
            ++P.inflight_calls;
        }
    }
}
// another model-graph call of this device has an in-launch recurrence that may still be running (g_lstm_mu held)
bool lstm_other_in_flight(const LstmPersistentDev& P, int dev) {
    const int own = (t_lstm_call && t_lstm_call->dev == dev && t_lstm_call->counted) ? 1 : 0;
    return P.inflight_calls - own > 0;
}
void lstm_call_settled(LstmPersistentDev& P, LstmCall* c) {  // the call's stream was synchronised on the host: nothing of it runs any more
    c->pending = 0;
    if (c->counted) {
        c->counted = false;
        --P.inflight_calls;
    }
}
}  // namespace

static int lstm_err_pool_init(LstmPersistentDev& P) {
    if (P.err_host) return QA_OK;
    QA_HIP(hipHostMalloc(reinterpret_cast<void**>(&P.err_host), sizeof(unsigned) * (LSTM_ERR_POOL + 1), hipHostMallocMapped));
    for (int i = 0; i <= LSTM_ERR_POOL; ++i) P.err_host[i] = 0u;
    QA_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&P.err_dev), P.err_host, 0));
    return QA_OK;
}

// sync ring + error word of a device (first use), and the backstop for a caller that did not collect an earlier failure
static int lstm_persistent_prepare(LstmPersistentDev& P, int dev) {
    if (!P.sync) {
        QA_HIP(hipDeviceGetAttribute(&P.cus, hipDeviceAttributeMultiprocessorCount, dev));
        QA_HIP(hipMalloc(reinterpret_cast<void**>(&P.sync), sizeof(unsigned) * LSTM_SYNC_RING * SY_WORDS * SY_STRIDE));
    }
    QA_TRY(lstm_err_pool_init(P));
    // backstop for launches made OUTSIDE a model-graph call (no ticket: they report into the device word and nobody collects it)
    volatile unsigned* dw = P.err_host + LSTM_ERR_POOL;
    if (*dw != 0u) {
        *dw = 0u;
        P.degraded = true;
        set_error("lstm: the grid barrier of an earlier persistent LSTM call on device %d timed out (its outputs are invalid): the kernel "
                  "needs every workgroup resident at once - another persistent kernel was sharing the device; set QA_LSTM_PERSISTENT=0", dev);
        return QA_ERR_HIP;
    }
    return QA_OK;
}

// returns QA_OK and sets *done = true when the persistent kernel took the call; *done = false: shape / device not eligible
static int launch_lstm_persistent(const float* xw, const float* w_hh_ug, float* h_out, float* c_state, int B, int T, int d, hipStream_t s,
                                  int dev, bool* done) {
    *done = false;
    const int mode = lstm_persistent_mode();
    LstmPersistentDev& P = g_lstm_p[dev];
    if (t_lstm_per_step || mode == 0 || (mode < 0 && (d < 1536 || P.degraded))) return QA_OK;
    if (!(d == 1536 || d == 1024 || d == 768 || d == 512) || T < 2) return QA_OK;
    QA_TRY(lstm_persistent_prepare(P, dev));
    // U hidden units per workgroup: the fewest that keep d / U workgroups (a multiple of the 8 barrier groups) <= CU count
    int U = 1;
    while (U <= 8 && (d % U || d / U > P.cus || (d / U) % 8)) ++U;
    const int NT = (4 * U + 15) / 16;
    if (U > 8 || NT > 2 || (NT == 1 && d == 1536)) return QA_OK;
    // QA_LSTM_FAULT (tests): the barrier waits for one workgroup more than exists, i.e. what a starved launch looks like
    const int nwg = d / U, ngrp = 8, per_grp = nwg / ngrp + (knob(K_LSTM_FAULT) ? 1 : 0);
    const unsigned spin_limit = (unsigned)std::max<long long>(64, std::min<long long>(knob(K_LSTM_SPIN_LIMIT), 1LL << 30));
    unsigned* const err_word = lstm_err_word(P, dev);
    for (int b0 = 0; b0 < B; b0 += 32) {
        const int bn = std::min(32, B - b0);
        const float* xw_b = xw + (long long)b0 * T * 4 * d;
        float* h_b = h_out + (long long)b0 * T * d;
        float* c_b = c_state + (long long)b0 * d;
        unsigned* sy = P.sync + (size_t)P.next * SY_WORDS * SY_STRIDE;
        P.next = (P.next + 1) % LSTM_SYNC_RING;
        QA_HIP(hipMemsetAsync(sy, 0, sizeof(unsigned) * SY_WORDS * SY_STRIDE, s));  // every polled word, before EVERY launch
#define QA_LP(MT, NT_, NI) \
    hipLaunchKernelGGL((lstm_persistent_kernel<MT, NT_, NI>), dim3(nwg), dim3(512), 0, s, xw_b, w_hh_ug, h_b, c_b, bn, T, d, U, sy, ngrp, per_grp, err_word, spin_limit)
        const bool two = bn > 16;
        if (d == 1536) { if (two) QA_LP(2, 2, 12); else QA_LP(1, 2, 12); }
        else if (d == 1024 && NT == 2) { if (two) QA_LP(2, 2, 8); else QA_LP(1, 2, 8); }
        else if (d == 1024) { if (two) QA_LP(2, 1, 8); else QA_LP(1, 1, 8); }
        else if (d == 768 && NT == 2) { if (two) QA_LP(2, 2, 6); else QA_LP(1, 2, 6); }
        else if (d == 768) { if (two) QA_LP(2, 1, 6); else QA_LP(1, 1, 6); }
        else if (NT == 2) { if (two) QA_LP(2, 2, 4); else QA_LP(1, 2, 4); }
        else { if (two) QA_LP(2, 1, 4); else QA_LP(1, 1, 4); }
#undef QA_LP
        QA_LAUNCH_CHECK();
        lstm_count_launch(P, dev);
    }
    *done = true;
    return QA_OK;
}

// QA_LSTM_XCD: the XCD-local recurrence for the widths whose W_hh fits one XCD's registers (d = 512 / 768); *done as above
static int launch_lstm_xcd(const float* xw, const float* w_hh_ug, float* h_out, float* c_state, int B, int T, int d, hipStream_t s, int dev,
                           bool* done) {
    *done = false;
    const int mode = (int)knob(K_LSTM_XCD);
    LstmPersistentDev& P = g_lstm_p[dev];
    if (t_lstm_per_step || mode <= 0 || P.degraded || !(d == 512 || d == 768) || T < 2) return QA_OK;
    QA_TRY(lstm_persistent_prepare(P, dev));
    const int per_team = 32, n_teams = P.cus / per_team;
    if (n_teams < 1 || n_teams > 8 || P.cus % per_team) return QA_OK;
    const unsigned spin_limit = (unsigned)std::max<long long>(64, std::min<long long>(knob(K_LSTM_SPIN_LIMIT), 1LL << 30));
    const int pad = 96 * 1024;  // dynamic LDS nobody touches: ONE workgroup per CU, so the grid spreads 32 per XCD
    const int fault = knob(K_LSTM_FAULT) ? 1 : 0;
    if (d == 512) QA_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(lstm_xcd_kernel<512, 8>), pad));
    else QA_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(lstm_xcd_kernel<768, 12>), pad));
    for (int b0 = 0; b0 < B; b0 += 4 * n_teams) {
        const int bn = std::min(4 * n_teams, B - b0);
        const float* xw_b = xw + (long long)b0 * T * 4 * d;
        float* h_b = h_out + (long long)b0 * T * d;
        float* c_b = c_state + (long long)b0 * d;
        unsigned* sy = P.sync + (size_t)P.next * SY_WORDS * SY_STRIDE;
        P.next = (P.next + 1) % LSTM_SYNC_RING;
        QA_HIP(hipMemsetAsync(sy, 0, sizeof(unsigned) * SY_WORDS * SY_STRIDE, s));
        const dim3 grid((unsigned)(n_teams * per_team));
        if (d == 512)
            hipLaunchKernelGGL((lstm_xcd_kernel<512, 8>), grid, dim3(512), pad, s, xw_b, w_hh_ug, h_b, c_b, bn, T, sy, per_team + fault, n_teams,
                               lstm_err_word(P, dev), spin_limit, mode >= 2 ? 1 : 0);
        else
            hipLaunchKernelGGL((lstm_xcd_kernel<768, 12>), grid, dim3(768), pad, s, xw_b, w_hh_ug, h_b, c_b, bn, T, sy, per_team + fault, n_teams,
                               lstm_err_word(P, dev), spin_limit, mode >= 2 ? 1 : 0);
        QA_LAUNCH_CHECK();
        lstm_count_launch(P, dev);
    }
    *done = true;
    return QA_OK;
}

// QA_LSTM_TEAM: d = 1024 on 4 teams of 64 workgroups - see lstm_team_kernel; *done as above
template <int D, int PT, int NW>
static int launch_lstm_team_t(LstmPersistentDev& P, int dev, const float* xw, const float* w_hh_ug, float* h_out, float* c_state, int B, int T,
                              hipStream_t s) {
    constexpr int SG = 2;
    const int n_teams = P.cus / PT;
    const unsigned spin_limit = (unsigned)std::max<long long>(64, std::min<long long>(knob(K_LSTM_SPIN_LIMIT), 1LL << 30));
    const int dyn = 96 * 1024;  // s_h (4 SG sequences x (D + 4) floats: 33 / 49 KB) + padding: with the static 16 KB one workgroup per CU
    static_assert(4 * SG * (D + 4) * 4 <= 96 * 1024, "lstm_team: h staging does not fit the dynamic segment");
    const int fault = knob(K_LSTM_FAULT) ? 1 : 0;
    QA_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(lstm_team_kernel<D, NW, PT, SG>), dyn));
    const int per_launch = 4 * SG * n_teams;
    for (int b0 = 0; b0 < B; b0 += per_launch) {
        const int bn = std::min(per_launch, B - b0);
        unsigned* sy = P.sync + (size_t)P.next * SY_WORDS * SY_STRIDE;
        P.next = (P.next + 1) % LSTM_SYNC_RING;
        QA_HIP(hipMemsetAsync(sy, 0, sizeof(unsigned) * SY_WORDS * SY_STRIDE, s));
        hipLaunchKernelGGL((lstm_team_kernel<D, NW, PT, SG>), dim3((unsigned)(n_teams * PT)), dim3(NW * 64), dyn, s, xw + (long long)b0 * T * 4 * D, w_hh_ug,
                           h_out + (long long)b0 * T * D, c_state + (long long)b0 * D, bn, T, sy, n_teams, fault, lstm_err_word(P, dev), spin_limit);
        QA_LAUNCH_CHECK();
        lstm_count_launch(P, dev);
    }
    return QA_OK;
}

static int launch_lstm_team(const float* xw, const float* w_hh_ug, float* h_out, float* c_state, int B, int T, int d, hipStream_t s, int dev,
                            bool* done) {
    *done = false;
    LstmPersistentDev& P = g_lstm_p[dev];
    // d = 1024 (H-Codec 1.5 decoder): 4 teams of 64 workgroups x 8 waves, 16 hidden units per workgroup in 128 VGPRs per lane.
    // d = 1536 (H-Codec 2.0) on 2 teams of 128 workgroups x 12 waves (again 128 resident weights per lane; h staged through registers or by
    // LDS-DMA) was built and measured in r05: parity green, ~10 us per step against 8.5 for lstm_persistent_kernel (a 128-member arrival
    // counter alone is ~1.5 us of serialised atomics; profiles/r05_lstm_team1536_ab.txt) - removed again, d = 1536 stays on that kernel
    if (t_lstm_per_step || knob(K_LSTM_TEAM) <= 0 || d != 1024 || P.degraded || T < 2) return QA_OK;
    QA_TRY(lstm_persistent_prepare(P, dev));
    if (P.cus != 256) return QA_OK;  // the team size is laid out for 256 CUs
    QA_TRY((launch_lstm_team_t<1024, 64, 8>(P, dev, xw, w_hh_ug, h_out, c_state, B, T, s)));
    *done = true;
    return QA_OK;
}

// ---- what the model graphs (hcodec.cpp) do about a timed-out barrier: a call opens a ticket (lstm_call_begin), and if it launched an
// in-launch recurrence it waits for its stream before returning (lstm_call_end), reads ITS error word, and - if a barrier broke - runs
// itself again on the per-step kernels, so the call that HIT the failure still returns valid results (the error used to
// surface one call late, or never; ADVICE r03: per call, not per device - two handles on two threads cannot collect each other's word).
int lstm_call_begin(int dev, void** ticket) {
    *ticket = nullptr;
    if (dev < 0 || dev >= 16) return QA_OK;
    std::lock_guard<std::mutex> lock(g_lstm_mu);
    LstmPersistentDev& P = g_lstm_p[dev];
    QA_TRY(lstm_err_pool_init(P));
    LstmCall* c = new LstmCall();
    c->dev = dev;
    for (int i = 0; i < LSTM_ERR_POOL; ++i)  // more than LSTM_ERR_POOL concurrent calls: the rest share the device word
        if (!(P.pool_busy >> i & 1ull)) {
            P.pool_busy |= 1ull << i;
            c->slot = i;
            c->err_host = P.err_host + i;
            c->err_dev = P.err_dev + i;
            *c->err_host = 0u;
            break;
        }
    t_lstm_call = c;
    *ticket = c;
    return QA_OK;
}

// The model graph just synchronised the call's stream on the host for a reason of its own (H-Codec 1.5 reads the data-dependent group
// count back, hcodec.cpp read_scalar): every recurrence launched so far has finished, so its error word can be read NOW and the call
// need not synchronise again at its end unless it launches another one - encode of H-Codec 1.5 returns asynchronously again (its
// LSTMs sit in front of the alignment), which hides the host's preparation of the next call behind ~40 ms of aggregator kernels.
void lstm_call_note_sync() {
    LstmCall* c = t_lstm_call;
    if (!c || !c->pending) return;
    std::lock_guard<std::mutex> lock(g_lstm_mu);
    LstmPersistentDev& P = g_lstm_p[c->dev];
    volatile unsigned* w = c->err_host ? c->err_host : (P.err_host ? P.err_host + LSTM_ERR_POOL : nullptr);
    if (w && *w != 0u) {
        *w = 0u;
        c->failed = true;
    }
    lstm_call_settled(P, c);
}

int lstm_call_end(void* ticket, hipStream_t s, bool* failed) {
    *failed = false;
    LstmCall* c = static_cast<LstmCall*>(ticket);
    if (!c) return QA_OK;
    if (t_lstm_call == c) t_lstm_call = nullptr;
    int st = QA_OK;
    if (c->pending) {  // only a call with a recurrence still in flight pays the host synchronisation
        if (hipStreamSynchronize(s) != hipSuccess) {
            set_error("lstm: hipStreamSynchronize failed while collecting an in-launch recurrence");
            st = QA_ERR_HIP;
        }
    }
    std::lock_guard<std::mutex> lock(g_lstm_mu);
    LstmPersistentDev& P = g_lstm_p[c->dev];
    // a ticket without a pool word (pool exhausted) reported into the device word (lstm_err_word)
    volatile unsigned* w = c->err_host ? c->err_host : (P.err_host ? P.err_host + LSTM_ERR_POOL : nullptr);
    if (st == QA_OK && c->launches && ((w && *w != 0u) || c->failed)) {
        if (w) *w = 0u;
        if (!c->fault_injected) P.degraded = true;  // an injected fault (tests; recorded at LAUNCH time) says nothing about the device
        *failed = true;
    }
    if (st == QA_OK) {
        lstm_call_settled(P, c);
        if (c->slot >= 0) P.pool_busy &= ~(1ull << c->slot);
    }
    // a FAILED synchronisation: the recurrence may still be running and could write a word the next call would be handed -
    // the slot stays busy (leaked: 64 per device) and the call stays counted as in flight, so later launches keep to the per-step kernels.
    // that fallback is permanent for the process - make it visible: the device is marked degraded (qa_debug_lstm_stats out[3],
    // and the auto modes stop choosing the in-launch kernels by that flag rather than by a silently stuck counter)
    if (st != QA_OK) {
        P.degraded = true;
        set_error("lstm: hipStreamSynchronize failed while collecting an in-launch recurrence; device %d keeps to the per-step kernels from here on "
                  "(qa_debug_lstm_stats reports degraded = 1)", c->dev);
    }
    delete c;
    return st;
}

void lstm_force_per_step(bool on) { t_lstm_per_step = on; }

}  // namespace qa
// diagnostics (tests): out[0] in-launch recurrences launched so far on `device`, out[1] calls with one possibly still in flight,
// out[2] launches diverted to the per-step kernels because another call's recurrence was in flight, out[3] degraded flag
extern "C" int qa_debug_lstm_stats(int32_t device, int64_t* out) {
    if (!out || device < 0 || device >= 16) {
        qa::set_error("qa_debug_lstm_stats: bad argument");
        return QA_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lock(qa::g_lstm_mu);
    const qa::LstmPersistentDev& P = qa::g_lstm_p[device];
    out[0] = (int64_t)P.launches;
    out[1] = P.inflight_calls;
    out[2] = (int64_t)P.diverted;
    out[3] = P.degraded ? 1 : 0;
    return QA_OK;
}
namespace qa {

// The T step launches of one call as a hipGraph: captured once per (buffers, shape) - the model graphs re-use the same arena
// addresses call after call - and replayed, so the host issues one graph launch instead of T kernel launches (eager launches go
// host-bound below ~3.5 us per kernel).  QA_LSTM_GRAPH=0 keeps the eager launches.
namespace {
struct LstmGraph {
    const void *xw, *w, *h, *c;
    int B, T, d, device;
    hipGraph_t graph;
    hipGraphExec_t exec;
    unsigned long long stamp;
};
std::vector<LstmGraph> g_lstm_graphs;
hipStream_t g_lstm_cap[16] = {};
unsigned long long g_lstm_clock = 0;
constexpr size_t LSTM_GRAPH_CACHE = 24;
}  // namespace

int launch_lstm(const float* xw, const float* w_hh_ug, float* h_out, float* c_state, int B, int T, int d,
                hipStream_t s, bool eager) {
    QA_REQUIRE(d % 128 == 0, "lstm: hidden size %d must be a multiple of 128", d);
    const bool use_graph = knob(K_LSTM_GRAPH) != 0;
    int dev = 0;
    QA_HIP(hipGetDevice(&dev));
    QA_REQUIRE(dev >= 0 && dev < 16, "lstm: device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_lstm_mu);
    if (eager) return lstm_launch_steps(xw, w_hh_ug, h_out, c_state, B, T, d, s);
    if (lstm_other_in_flight(g_lstm_p[dev], dev)) {
        ++g_lstm_p[dev].diverted;  // another handle's in-launch recurrence may be running: no second whole-device kernel beside it
    } else {
        bool done = false;
        QA_TRY(launch_lstm_xcd(xw, w_hh_ug, h_out, c_state, B, T, d, s, dev, &done));
        if (done) return QA_OK;
        QA_TRY(launch_lstm_team(xw, w_hh_ug, h_out, c_state, B, T, d, s, dev, &done));
        if (done) return QA_OK;
        QA_TRY(launch_lstm_persistent(xw, w_hh_ug, h_out, c_state, B, T, d, s, dev, &done));
        if (done) return QA_OK;
    }
    if (!use_graph || T < 8) return lstm_launch_steps(xw, w_hh_ug, h_out, c_state, B, T, d, s);
    LstmGraph* hit = nullptr;
    for (LstmGraph& g : g_lstm_graphs)
        if (g.xw == xw && g.w == w_hh_ug && g.h == h_out && g.c == c_state && g.B == B && g.T == T && g.d == d && g.device == dev)
            hit = &g;
    if (!hit) {
        if (!g_lstm_cap[dev]) QA_HIP(hipStreamCreateWithFlags(&g_lstm_cap[dev], hipStreamNonBlocking));
        QA_HIP(hipStreamBeginCapture(g_lstm_cap[dev], hipStreamCaptureModeThreadLocal));
        const int st = lstm_launch_steps(xw, w_hh_ug, h_out, c_state, B, T, d, g_lstm_cap[dev]);
        hipGraph_t graph = nullptr;
        const hipError_t e = hipStreamEndCapture(g_lstm_cap[dev], &graph);
        if (st != QA_OK) {
            if (graph) (void)hipGraphDestroy(graph);
            return st;
        }
        QA_HIP(e);
        hipGraphExec_t exec = nullptr;
        QA_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        if (g_lstm_graphs.size() >= LSTM_GRAPH_CACHE) {  // evict the least recently used entry
            size_t lru = 0;
            for (size_t i = 1; i < g_lstm_graphs.size(); ++i)
                if (g_lstm_graphs[i].stamp < g_lstm_graphs[lru].stamp) lru = i;
            (void)hipGraphExecDestroy(g_lstm_graphs[lru].exec);
            (void)hipGraphDestroy(g_lstm_graphs[lru].graph);
            g_lstm_graphs.erase(g_lstm_graphs.begin() + (long)lru);
        }
        g_lstm_graphs.push_back(LstmGraph{xw, w_hh_ug, h_out, c_state, B, T, d, dev, graph, exec, 0});
        hit = &g_lstm_graphs.back();
    }
    hit->stamp = ++g_lstm_clock;
    QA_HIP(hipGraphLaunch(hit->exec, s));
    return QA_OK;
}

}  // namespace qa
