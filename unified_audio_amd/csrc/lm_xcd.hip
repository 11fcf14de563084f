// lm_xcd.hip - the UniSE greedy decode loop as ONE persistent launch per phase, one decode chain PER XCD (QA_LM_XCD, default 0:
// written at the end of round 3 after the GPU budget was spent - NOT YET RUN; its parity test is gated by QA_TEST_EXPERIMENTAL).
//
// Why: the launch-per-stage step is bound by its 62 dependent kernel boundaries (~3.3 us each + the fresh-
// data pull), not by bytes.  The XCD-local LSTM recurrence (lstm.hip lstm_xcd_kernel) showed what a dependent stage costs when the
// dependency stays inside one XCD: a 32-member counter barrier in ~1 us.  The sequences of a batch are independent through the
// whole step (GEMVs per row, attention per sequence), so they are dealt to the XCDs - team x = the workgroups that find themselves
// on XCD x (s_getreg HW_REG_XCC_ID), one per CU - and every team runs ITS sequences (b = x, x + 8, ...; at most 4) through all
// layers and all steps by itself: 5 team barriers per layer + 1 for the head instead of kernel boundaries, nothing crosses an XCD.
// The price: every team streams all weights (217 MB per step; the 8 teams run in near lockstep, so the other seven read them out of
// the Infinity Cache), i.e. the step is bound by ONE XCD's port: ~256 MB per step at B = 16 (weights + 2 sequences' K/V).
//
// Stage map of a layer (weights pre-laid tile-major per workgroup slot by lm.cpp, RMSNorm gains folded as in the other paths):
//   S1  RMSNorm + QKV rows of the slot (8 q rotary pairs, 8 k rotary pairs, 16 v rows) + RoPE + cache append       -> barrier
//   S2  single-query attention: slot = (sequence, head, key split) over the cache, partial record [o | m | l]     -> barrier
//   S3  merge of the partials + o_proj rows of the slot (16) + residual                                            -> barrier
//   S4  RMSNorm + gate / up rows of the slot's 64 activation columns + SwiGLU                                      -> barrier
//   S5  down_proj rows of the slot (16, K = 2048) + residual                                                       -> barrier
//   head: RMSNorm + the slot's rows of the active vocabulary slice, per-slot arg-max                               -> barrier,
//         then EVERY workgroup folds the 32 slot maxima of its team's sequences itself (first maximum wins) - no further barrier.
// All GEMVs run on v_mfma_f32_4x4x1_16b_f32 (A = 4 weight rows per block streamed with coalesced 16-byte loads, B = the team's <= 4
// activation rows from LDS): every FMA useful at M = 4.  Team hand-offs use the forms validated in lstm_xcd_kernel: write-through
// (sc1) stores + drained vmcnt + agent atomic + sc1 loads; activations that change every stage are always read with sc1 loads
// (a CU's L1 is never refreshed by another CU's stores).  Bounded spins -> error word -> qa_lm_generate reports the failure.
#include <algorithm>

#include "kernels.h"
#include "lm_xcd.h"

namespace qa {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int XD = 512, XH = 8, XHD = 64, XI = 2048;  // the UniSE Llama body this kernel is laid out for (lm_xcd_supported)
constexpr int LDX = XI + 4;                           // LDS row stride of the staged activations (floats)
constexpr int REC = XHD + 4;                          // attention partial record [o (64) | m | l | pad2]
constexpr int SX_STRIDE = 32;                         // sync words, one per 128-byte line: slot[8] | cnt[8] | err
enum { SXL_SLOT = 0, SXL_CNT = 8, SXL_ERR = 16 };
}  // namespace

#define QX_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ unsigned lmx_xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

__device__ __forceinline__ bool lmx_spin_until(unsigned* word, unsigned want, unsigned* err, unsigned* err_host, unsigned limit) {
    for (unsigned spins = 0; spins < limit; ++spins) {
        if (__hip_atomic_load(word, QX_RLX) >= want) return true;
        if ((spins & 1023u) == 1023u && __hip_atomic_load(err, QX_RLX) != 0u) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store(err, 1u, QX_RLX);
    __hip_atomic_store(err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return false;
}

// hand-off accessors: data another CU of the team wrote (or will read) during this launch
__device__ __forceinline__ f32x4 lmx_ld(const float* p) {  // 16-byte load that bypasses this CU's L1; the CALLER waits (lmx_wait)
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
#define lmx_wait(v) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)::"memory")
__device__ __forceinline__ void lmx_st4(float* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void lmx_st1(float* p, float v) { __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), QX_RLX); }
__device__ __forceinline__ f32x4 lmx_ldw(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }  // streamed once

struct LmxSync {
    unsigned *cnt, *err, *err_host;
    unsigned limit, members, epoch;
};

// A team barrier in two halves, so that loads which do not depend on the team (the NEXT stage's weights) can be issued in between:
// lmx_arrive - every wave's hand-off stores are drained and the workgroup reports the stage done; lmx_wait_team - nobody continues
// before the whole team has arrived.
__device__ __forceinline__ void lmx_arrive(LmxSync& ts, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++ts.epoch;
    if (tid == 0) (void)__hip_atomic_fetch_add(ts.cnt, 1u, QX_RLX);
}
__device__ __forceinline__ bool lmx_wait_team(LmxSync& ts, int* s_ok, int tid) {
    if (tid == 0) *s_ok = lmx_spin_until(ts.cnt, ts.members * ts.epoch, ts.err, ts.err_host, ts.limit) ? 1 : 0;
    __syncthreads();
    asm volatile("" ::: "memory");
    return *s_ok != 0;
}

// 16 output rows x K inputs for the team's sequences.  Lane = (block b = lane >> 2, i = lane & 3): row block rb = b & 3 (rows
// 4 rb + i as the MFMA's A operand), K phase kp = b >> 2; wave w and phase kp own the K slice (4 w + kp) of K / 32 values; the B
// operand is sequence i's slice of the staged activations.  Returns, valid in lanes < 16 (kp = 0): lane (rb, q) -> rows 4 rb + r.
template <int K>
__device__ __forceinline__ void lmx_ld16(const float* __restrict__ wt, int wave, int lane, f32x4* w) {  // w[K / 128]
    constexpr int NJ = K / 128;
    const float* wp = wt + ((size_t)wave * NJ * 64 + lane) * 4;
#pragma unroll
    for (int j = 0; j < NJ; ++j) w[j] = lmx_ldw(wp + (size_t)j * 256);
}
template <int K>
__device__ __forceinline__ f32x4 lmx_mm16(const f32x4* w, const float (*s_x)[LDX], int wave, int lane) {
    constexpr int KW = K / 32, NJ = KW / 4;
    const float* xp = &s_x[lane & 3][(wave * 4 + (lane >> 4)) * KW];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const f32x4 hb = *reinterpret_cast<const f32x4*>(xp + 4 * j);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w[j].x, hb.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w[j].y, hb.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w[j].z, hb.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w[j].w, hb.w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        acc[r] += __shfl_xor(acc[r], 16, 64);
        acc[r] += __shfl_xor(acc[r], 32, 64);
    }
    return acc;
}

// 128 output rows (two groups of 64: wave w -> group w & 1, K slice w >> 1 of 128 values) x 512 inputs.  Lane = row of the group
// (A operand), B = sequence (lane & 3).  Returns lane (block, q) -> rows 64 g + 4 block + r, partial over the wave's K slice.
__device__ __forceinline__ void lmx_ld128(const float* __restrict__ wt, int wave, int lane, f32x4* w) {  // w[32]
    const int g = wave & 1, ks = wave >> 1;
    const float* wp = wt + (((size_t)g * 4 + ks) * 32 * 64 + lane) * 4;
#pragma unroll
    for (int j = 0; j < 32; ++j) w[j] = lmx_ldw(wp + (size_t)j * 256);
}
__device__ __forceinline__ f32x4 lmx_mm128(const f32x4* w, const float (*s_x)[LDX], int wave, int lane) {
    const float* xp = &s_x[lane & 3][(wave >> 1) * 128];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const f32x4 hb = *reinterpret_cast<const f32x4*>(xp + 4 * j);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w[j].x, hb.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w[j].y, hb.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w[j].z, hb.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w[j].w, hb.w, acc, 0, 0, 0);
    }
    return acc;
}

// PF = the NEXT stage's weights are loaded between a workgroup's arrival at a team barrier and its wait for the team (they do not
// depend on the team's activations), so a stage starts with its weight stream already in flight: QA_LM_XCD=2 (1: PF = false)
template <bool PF>
__global__ __launch_bounds__(512, 1) void lm_xcd_decode_kernel(const LmXcdArgs a) {
    __shared__ __attribute__((aligned(16))) float s_x[4][LDX];          // staged activations of the team's sequences
    __shared__ __attribute__((aligned(16))) float s_pb[3][8][16][4];    // 16-row GEMVs: [part][wave][row block * 4 + q][r]
    __shared__ __attribute__((aligned(16))) float s_pa[4][128][4];      // 128-row GEMVs: [K slice][64 g + lane][r]
    __shared__ __attribute__((aligned(16))) float s_o[8][XHD];          // attention: per-wave states
    __shared__ float s_m[8], s_l[8], s_rs[4], s_bv[4][32];
    __shared__ int s_bi[4][32], s_tok[4], s_ctl[3];
    extern __shared__ float s_pad[];  // unused: sized by the host so that ONE workgroup fits a CU
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- team formation (as lstm_xcd_kernel)
    if (tid == 0) {
        const unsigned x = lmx_xcc_id();
        s_ctl[0] = (int)x;
        s_ctl[1] = x < 8u ? (int)__hip_atomic_fetch_add(a.sy + (SXL_SLOT + x) * SX_STRIDE, 1u, QX_RLX) : 1 << 20;
    }
    __syncthreads();
    const int team = s_ctl[0], slot = s_ctl[1];
    unsigned* err = a.sy + SXL_ERR * SX_STRIDE;
    if (slot >= 32 || team >= 8) {
        if (tid == 0) {
            __hip_atomic_store(err, 1u, QX_RLX);
            __hip_atomic_store(a.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const int nq = min(4, (a.B - team + 7) / 8);  // this team's sequences: b = team + 8 q
    if (nq <= 0) return;
    LmxSync ts{a.sy + (SXL_CNT + team) * SX_STRIDE, err, a.err_host, a.spin_limit, (unsigned)(32 + a.fault), 0u};
    int* s_ok = &s_ctl[2];
    const int S = nq == 1 ? 4 : (nq == 2 ? 2 : 1);  // key splits of the attention: nq * 8 * S <= 32 work items
    const float scale = 0.125f;                     // 1 / sqrt(64)
    const float eps = a.rms_eps;

    if (tid < 4) s_tok[tid] = (int)a.tok_init;
    __syncthreads();
    f32x4 wr[32];  // the weights of the next GEMV stage (PF: loaded one team barrier early)
    const auto load_s1 = [&](const LmXcdLayer& L) {
#pragma unroll
        for (int part = 0; part < 3; ++part) lmx_ld16<XD>(L.qkv + ((size_t)slot * 3 + part) * 16 * XD, wave, lane, wr + 4 * part);
    };
    if (PF) load_s1(a.layer[0]);

    for (int st = 0; st < a.steps; ++st) {
        const int pos = a.pos0 + st;
        for (int l = 0; l < a.n_layers; ++l) {
            const LmXcdLayer& L = a.layer[l];
            float* kc = a.kc + (size_t)l * a.kv_lstride;
            float* vc = a.vc + (size_t)l * a.kv_lstride;
            // ================================================================ S1: RMSNorm + QKV + RoPE + cache append
            {
                if (tid < 4 * (XD / 4)) {  // one float4 of x per thread: thread = (sequence, column group)
                    const int sq = tid >> 7, c4 = tid & 127, bq = team + 8 * min(sq, nq - 1);
                    f32x4 v;
                    if (l == 0) {
                        v = *reinterpret_cast<const f32x4*>(a.emb + (size_t)s_tok[min(sq, nq - 1)] * XD + 4 * c4);  // codec_embedding gather
                    } else {
                        v = lmx_ld(a.xa + (size_t)bq * XD + 4 * c4);
                        lmx_wait(v);
                    }
                    *reinterpret_cast<f32x4*>(&s_x[sq][4 * c4]) = v;
                }
                __syncthreads();
                if (wave < 4) {  // RMS statistics of sequence `wave`
                    float sq = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float v = s_x[wave][lane + 64 * i];
                        sq = fmaf(v, v, sq);
                    }
#pragma unroll
                    for (int of = 32; of > 0; of >>= 1) sq += __shfl_xor(sq, of, 64);
                    if (lane == 0) s_rs[wave] = rsqrtf(sq / XD + eps);
                }
                if (!PF) load_s1(L);
#pragma unroll
                for (int part = 0; part < 3; ++part) {
                    const f32x4 acc = lmx_mm16<XD>(wr + 4 * part, s_x, wave, lane);
                    if (lane < 16) *reinterpret_cast<f32x4*>(&s_pb[part][wave][lane][0]) = acc;
                }
                __syncthreads();
                if (tid < 48) {
                    const int part = tid >> 4, e = tid & 15, rb = e >> 2, q = e & 3;
                    if (q < nq) {
                        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int w = 0; w < 8; ++w) v += *reinterpret_cast<const f32x4*>(&s_pb[part][w][e][0]);
                        const float rs = s_rs[q];
                        v *= rs;
                        const int b = team + 8 * q;
                        if (part == 2) {  // V rows 16 slot + 4 rb .. + 3: no rotation
                            lmx_st4(vc + (size_t)b * a.kv_bstride + (size_t)pos * XD + 16 * slot + 4 * rb, v);
                        } else {  // rows (2 p, 2 p + 1) = rotary pair p: dims (jj, jj + 32) of head h (rotate-half RoPE)
                            float* dst = part == 0 ? a.qbuf + (size_t)b * XD : kc + (size_t)b * a.kv_bstride + (size_t)pos * XD;
#pragma unroll
                            for (int pr = 0; pr < 2; ++pr) {
                                const int P = slot * 8 + 2 * rb + pr, h = P >> 5, jj = P & 31;
                                const float c = a.rope[((size_t)pos * 32 + jj) * 2], sn = a.rope[((size_t)pos * 32 + jj) * 2 + 1];
                                const float v1 = v[2 * pr], v2 = v[2 * pr + 1];
                                lmx_st1(dst + h * XHD + jj, v1 * c - v2 * sn);
                                lmx_st1(dst + h * XHD + jj + 32, v2 * c + v1 * sn);
                            }
                        }
                    }
                }
                lmx_arrive(ts, tid);
                if (PF) lmx_ld16<XD>(L.o + (size_t)slot * 16 * XD, wave, lane, wr);  // S3's weights ride through the attention stage
                if (!lmx_wait_team(ts, s_ok, tid)) return;
            }
            // ================================================================ S2: attention (lm_attn_kernel's body, one work item per slot)
            if (slot < nq * XH * S) {
                const int q = slot / (XH * S), h = (slot / S) % XH, sp = slot % S, b = team + 8 * q;
                constexpr int LPK = 16, KPI = 4, NI = 4;
                const int kq = lane / LPK, c4 = (lane % LPK) * 4;
                const int n_keys = pos + 1, n_tiles = (n_keys + 15) >> 4;
                const float* kb = kc + (size_t)b * a.kv_bstride + h * XHD + c4;
                const float* vb = vc + (size_t)b * a.kv_bstride + h * XHD + c4;
                f32x4 qv = lmx_ld(a.qbuf + (size_t)b * XD + h * XHD + c4);
                lmx_wait(qv);
                qv *= scale;
                float m_run = -INFINITY, l_run = 0.f;
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                for (int it = 0;; it += 2) {
                    const int ta = (it * 8 + wave) * S + sp, tb = ((it + 1) * 8 + wave) * S + sp;
                    if (ta >= n_tiles) break;
                    f32x4 kt[2][NI], vt[2][NI];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int t = u == 0 ? ta : min(tb, n_tiles - 1);
#pragma unroll
                        for (int j = 0; j < NI; ++j) {
                            const int row = min(t * 16 + j * KPI + kq, n_keys - 1);
                            // the newest row was written this step by other CUs of the team: never in this CU's L1 before (first touch)
                            kt[u][j] = lmx_ldw(kb + (size_t)row * XD);
                            vt[u][j] = lmx_ldw(vb + (size_t)row * XD);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int t = u == 0 ? ta : tb;
                        if (t >= n_tiles) break;
                        float sc[NI], tmax = -INFINITY;
#pragma unroll
                        for (int j = 0; j < NI; ++j) {
                            float d = qv.x * kt[u][j].x;
                            d = fmaf(qv.y, kt[u][j].y, d);
                            d = fmaf(qv.z, kt[u][j].z, d);
                            d = fmaf(qv.w, kt[u][j].w, d);
#pragma unroll
                            for (int of = 1; of < LPK; of <<= 1) d += __shfl_xor(d, of, 64);
                            sc[j] = (t * 16 + j * KPI + kq < n_keys) ? d : -INFINITY;
                            tmax = fmaxf(tmax, sc[j]);
                        }
#pragma unroll
                        for (int of = LPK; of < 64; of <<= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, of, 64));
                        const float m_new = fmaxf(m_run, tmax);
                        const float alpha = expf(m_run - m_new);
                        float psum = 0.f;
                        o *= alpha;
#pragma unroll
                        for (int j = 0; j < NI; ++j) {
                            const float p = expf(sc[j] - m_new);
                            psum += p;
                            o.x = fmaf(p, vt[u][j].x, o.x);
                            o.y = fmaf(p, vt[u][j].y, o.y);
                            o.z = fmaf(p, vt[u][j].z, o.z);
                            o.w = fmaf(p, vt[u][j].w, o.w);
                        }
#pragma unroll
                        for (int of = LPK; of < 64; of <<= 1) psum += __shfl_xor(psum, of, 64);
                        l_run = l_run * alpha + psum;
                        m_run = m_new;
                    }
                }
#pragma unroll
                for (int of = LPK; of < 64; of <<= 1) {
                    o.x += __shfl_xor(o.x, of, 64);
                    o.y += __shfl_xor(o.y, of, 64);
                    o.z += __shfl_xor(o.z, of, 64);
                    o.w += __shfl_xor(o.w, of, 64);
                }
                if (lane < LPK) {
                    *reinterpret_cast<f32x4*>(&s_o[wave][c4]) = o;
                    if (lane == 0) {
                        s_m[wave] = m_run;
                        s_l[wave] = l_run;
                    }
                }
                __syncthreads();
                if (tid < XHD + 2) {
                    float m = s_m[0];
                    for (int w = 1; w < 8; ++w) m = fmaxf(m, s_m[w]);
                    float lsum = 0.f, accv = 0.f;
                    for (int w = 0; w < 8; ++w) {
                        const float f = (s_m[w] == -INFINITY) ? 0.f : expf(s_m[w] - m);
                        lsum += s_l[w] * f;
                        if (tid < XHD) accv += s_o[w][tid] * f;
                    }
                    float* rec = a.att_part + ((size_t)(b * XH + h) * 4 + sp) * REC;
                    lmx_st1(rec + tid, tid < XHD ? accv : (tid == XHD ? m : lsum));
                }
            }
            lmx_arrive(ts, tid);
            if (!lmx_wait_team(ts, s_ok, tid)) return;
            // ================================================================ S3: merge of the attention partials + o_proj + residual
            {
                if (tid < 4 * (XD / 4)) {
                    const int sq = tid >> 7, c4 = tid & 127, bq = team + 8 * min(sq, nq - 1);
                    const int h = c4 >> 4, i = (c4 & 15) * 4;
                    const float* base = a.att_part + ((size_t)(bq * XH + h) * 4) * REC;
                    f32x4 ov[4], ml[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {  // every load first (records past S are never read: clamp to a valid one)
                        const int ss = min(s, S - 1);
                        ov[s] = lmx_ld(base + ss * REC + i);
                        ml[s] = lmx_ld(base + ss * REC + XHD);
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        lmx_wait(ov[s]);
                        lmx_wait(ml[s]);
                    }
                    float m = -INFINITY;
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        if (s < S) m = fmaxf(m, ml[s][0]);
                    float Lsum = 0.f;
                    f32x4 out = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float w = (s < S && ml[s][0] != -INFINITY) ? expf(ml[s][0] - m) : 0.f;
                        Lsum = fmaf(ml[s][1], w, Lsum);
                        out += ov[s] * w;
                    }
                    out *= 1.0f / Lsum;
                    *reinterpret_cast<f32x4*>(&s_x[sq][4 * c4]) = out;
                }
                __syncthreads();
                if (!PF) lmx_ld16<XD>(L.o + (size_t)slot * 16 * XD, wave, lane, wr);
                const f32x4 acc = lmx_mm16<XD>(wr, s_x, wave, lane);
                if (lane < 16) *reinterpret_cast<f32x4*>(&s_pb[0][wave][lane][0]) = acc;
                __syncthreads();
                if (tid < 16) {
                    const int rb = tid >> 2, q = tid & 3;
                    if (q < nq) {
                        const int b = team + 8 * q, col = 16 * slot + 4 * rb;
                        f32x4 res;
                        if (l == 0) {
                            res = *reinterpret_cast<const f32x4*>(a.emb + (size_t)s_tok[q] * XD + col);
                        } else {
                            res = lmx_ld(a.xa + (size_t)b * XD + col);
                            lmx_wait(res);
                        }
                        f32x4 v = res;
#pragma unroll
                        for (int w = 0; w < 8; ++w) v += *reinterpret_cast<const f32x4*>(&s_pb[0][w][tid][0]);
                        lmx_st4(a.xb + (size_t)b * XD + col, v);
                    }
                }
                lmx_arrive(ts, tid);
                if (PF) lmx_ld128(L.gu + (size_t)slot * 128 * XD, wave, lane, wr);
                if (!lmx_wait_team(ts, s_ok, tid)) return;
            }
            // ================================================================ S4: RMSNorm + gate / up + SwiGLU
            {
                if (tid < 4 * (XD / 4)) {
                    const int sq = tid >> 7, c4 = tid & 127, bq = team + 8 * min(sq, nq - 1);
                    f32x4 v = lmx_ld(a.xb + (size_t)bq * XD + 4 * c4);
                    lmx_wait(v);
                    *reinterpret_cast<f32x4*>(&s_x[sq][4 * c4]) = v;
                }
                __syncthreads();
                if (wave < 4) {
                    float sq = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float v = s_x[wave][lane + 64 * i];
                        sq = fmaf(v, v, sq);
                    }
#pragma unroll
                    for (int of = 32; of > 0; of >>= 1) sq += __shfl_xor(sq, of, 64);
                    if (lane == 0) s_rs[wave] = rsqrtf(sq / XD + eps);
                }
                if (!PF) lmx_ld128(L.gu + (size_t)slot * 128 * XD, wave, lane, wr);
                const f32x4 acc = lmx_mm128(wr, s_x, wave, lane);
                *reinterpret_cast<f32x4*>(&s_pa[wave >> 1][64 * (wave & 1) + lane][0]) = acc;
                __syncthreads();
                if (tid < 64) {  // lane (block, q): activation columns 64 slot + 4 block + r of sequence q
                    const int q = tid & 3, blk = tid >> 2;
                    if (q < nq) {
                        f32x4 g = {0.f, 0.f, 0.f, 0.f}, u = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            g += *reinterpret_cast<const f32x4*>(&s_pa[k][tid][0]);
                            u += *reinterpret_cast<const f32x4*>(&s_pa[k][64 + tid][0]);
                        }
                        const float rs = s_rs[q];
                        f32x4 act;
#pragma unroll
                        for (int r = 0; r < 4; ++r) act[r] = silu_f(g[r] * rs) * (u[r] * rs);
                        lmx_st4(a.act + (size_t)(team + 8 * q) * XI + 64 * slot + 4 * blk, act);
                    }
                }
                lmx_arrive(ts, tid);
                if (PF) lmx_ld16<XI>(L.down + (size_t)slot * 16 * XI, wave, lane, wr);
                if (!lmx_wait_team(ts, s_ok, tid)) return;
            }
            // ================================================================ S5: down_proj + residual
            {
                f32x4 av[4];  // 4 x 2048 floats = 2048 float4: four per thread, all in flight at once
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = tid + 512 * i, sq = f >> 9, c4 = f & 511, bq = team + 8 * min(sq, nq - 1);
                    av[i] = lmx_ld(a.act + (size_t)bq * XI + 4 * c4);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) lmx_wait(av[i]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = tid + 512 * i;
                    *reinterpret_cast<f32x4*>(&s_x[f >> 9][4 * (f & 511)]) = av[i];
                }
                __syncthreads();
                if (!PF) lmx_ld16<XI>(L.down + (size_t)slot * 16 * XI, wave, lane, wr);
                const f32x4 acc = lmx_mm16<XI>(wr, s_x, wave, lane);
                if (lane < 16) *reinterpret_cast<f32x4*>(&s_pb[0][wave][lane][0]) = acc;
                __syncthreads();
                if (tid < 16) {
                    const int rb = tid >> 2, q = tid & 3;
                    if (q < nq) {
                        const int b = team + 8 * q, col = 16 * slot + 4 * rb;
                        f32x4 v = lmx_ld(a.xb + (size_t)b * XD + col);
                        lmx_wait(v);
#pragma unroll
                        for (int w = 0; w < 8; ++w) v += *reinterpret_cast<const f32x4*>(&s_pb[0][w][tid][0]);
                        lmx_st4(a.xa + (size_t)b * XD + col, v);
                    }
                }
                lmx_arrive(ts, tid);
                // the next layer's QKV rows.  UNCONDITIONAL (the last layer re-loads its own, unused): a conditional re-definition of
                // the weight registers makes every stale value live around the layer loop and the kernel spills (137 VGPRs measured)
                if (PF) load_s1(a.layer[min(l + 1, a.n_layers - 1)]);
                if (!lmx_wait_team(ts, s_ok, tid)) return;
            }
        }
        // ==================================================================== head: final RMSNorm + vocabulary slice + arg-max
        {
            if (tid < 4 * (XD / 4)) {
                const int sq = tid >> 7, c4 = tid & 127, bq = team + 8 * min(sq, nq - 1);
                f32x4 v = lmx_ld(a.xa + (size_t)bq * XD + 4 * c4);
                lmx_wait(v);
                *reinterpret_cast<f32x4*>(&s_x[sq][4 * c4]) = v;
            }
            __syncthreads();
            const int rows_wg = a.width / 32, n_chunk = rows_wg / 128;  // this slot's rows: [slot * rows_wg, + rows_wg) of the slice
            float best = -INFINITY;
            int bi = 0x7fffffff;
            for (int ch = 0; ch < n_chunk; ++ch) {
                lmx_ld128(a.head + ((size_t)slot * rows_wg + (size_t)ch * 128) * XD, wave, lane, wr);  // not prefetched (see S5)
                const f32x4 acc = lmx_mm128(wr, s_x, wave, lane);
                *reinterpret_cast<f32x4*>(&s_pa[wave >> 1][64 * (wave & 1) + lane][0]) = acc;
                __syncthreads();
                if (tid < 128) {  // (group g, block, q): rows 64 g + 4 block + r of this chunk (the positive RMS factor does not move the arg-max)
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 4; ++k) v += *reinterpret_cast<const f32x4*>(&s_pa[k][tid][0]);
                    const int row0 = slot * rows_wg + ch * 128 + 64 * (tid >> 6) + 4 * ((tid & 63) >> 2);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (v[r] > best) {  // rows ascend with r and ch: the first maximum wins
                            best = v[r];
                            bi = row0 + r;
                        }
                }
                __syncthreads();
            }
            if (tid < 128) {  // thread -> (q = tid & 3, j = tid >> 2 of 32)
                s_bv[tid & 3][tid >> 2] = best;
                s_bi[tid & 3][tid >> 2] = bi;
            }
            __syncthreads();
            if (wave < nq) {  // wave q folds the 32 (group, block) candidates of sequence q: lanes j and j + 32 hold the same entry
                float v = s_bv[wave][lane & 31];
                int i = s_bi[wave][lane & 31];
#pragma unroll
                for (int of = 16; of > 0; of >>= 1) {
                    const float v2 = __shfl_xor(v, of, 64);
                    const int i2 = __shfl_xor(i, of, 64);
                    if (v2 > v || (v2 == v && i2 < i)) {
                        v = v2;
                        i = i2;
                    }
                }
                if (lane == 0) {
                    const int b = team + 8 * wave;
                    lmx_st1(a.pmax + (size_t)b * 32 + slot, v);
                    __hip_atomic_store(reinterpret_cast<unsigned*>(a.pidx + (size_t)b * 32 + slot), (unsigned)i, QX_RLX);
                }
            }
            lmx_arrive(ts, tid);
            if (PF) load_s1(a.layer[0]);  // the next step's first stage (unconditional: unused after the last step)
            if (!lmx_wait_team(ts, s_ok, tid)) return;
            if (wave < nq) {  // every workgroup folds its team's 32 slot maxima itself (one load per lane): no further barrier
                const int b = team + 8 * wave;
                float v = __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(a.pmax + (size_t)b * 32 + (lane & 31)), QX_RLX));
                int i = (int)__hip_atomic_load(reinterpret_cast<const unsigned*>(a.pidx + (size_t)b * 32 + (lane & 31)), QX_RLX);
#pragma unroll
                for (int of = 16; of > 0; of >>= 1) {
                    const float v2 = __shfl_xor(v, of, 64);
                    const int i2 = __shfl_xor(i, of, 64);
                    if (v2 > v || (v2 == v && i2 < i)) {
                        v = v2;
                        i = i2;
                    }
                }
                if (lane == 0) {
                    if (i == 0x7fffffff) i = 0;  // all-NaN row: stay inside the table
                    s_tok[wave] = a.lo + i;
                    if (slot == 0) {
                        a.tok[b] = a.lo + i;
                        if (st < a.keep) a.ids[(size_t)b * a.ids_ld + st] = i;
                    }
                }
            }
            __syncthreads();
        }
    }
}

bool lm_xcd_supported(int d, int heads, int inter, int global_size, int semantic_size) {
    return d == XD && heads == XH && inter == XI && global_size % 4096 == 0 && semantic_size % 4096 == 0 && global_size > 0 && semantic_size > 0;
}

size_t lm_xcd_sync_bytes() { return sizeof(unsigned) * 17 * SX_STRIDE; }

int launch_lm_xcd_decode(const LmXcdArgs& a, hipStream_t s) {
    QA_REQUIRE(a.B >= 1 && a.B <= 32 && a.n_layers >= 1 && a.n_layers <= LM_XCD_MAX_LAYERS && a.width % 4096 == 0 && a.steps >= 0,
               "lm_xcd: unsupported launch (B=%d layers=%d width=%d)", a.B, a.n_layers, a.width);
    if (a.steps == 0) return QA_OK;
    int dev = 0, cus = 0;
    QA_HIP(hipGetDevice(&dev));
    QA_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    QA_REQUIRE(cus == 256, "lm_xcd: laid out for 8 XCDs x 32 CUs, this device has %d CUs", cus);
    const int pad = 64 * 1024;  // with the ~55 KB of static LDS: one workgroup per CU, so the 256-workgroup grid lands 32 per XCD
    QA_HIP(hipMemsetAsync(a.sy, 0, lm_xcd_sync_bytes(), s));
    if (a.prefetch) {
        QA_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(lm_xcd_decode_kernel<true>), pad));
        hipLaunchKernelGGL(lm_xcd_decode_kernel<true>, dim3(256), dim3(512), pad, s, a);
    } else {
        QA_TRY(raise_dynamic_lds(reinterpret_cast<const void*>(lm_xcd_decode_kernel<false>), pad));
        hipLaunchKernelGGL(lm_xcd_decode_kernel<false>, dim3(256), dim3(512), pad, s, a);
    }
    QA_LAUNCH_CHECK();
    return QA_OK;
}

}  // namespace qa
