// lm_decode.hip - the UniSE AR-LM decode step as FIVE fat launches per layer plus two for the head
// (QuarkAudio-UniSE/model/llm/llm.py:150-227 body, llm_sft.py:137-193 loop, llm.py:253-288 sampling):
//
//   per layer  1. qkv     RMSNorm (folded) + QKV GEMV + rotate-half RoPE + K/V append to the cache at `pos`
//              2. attn    single-query attention over the cache, keys split over S workgroups -> (m, l, o) partials
//              3. oproj   combine of the partials (inside the A-operand loader) + o_proj GEMV + residual
//              4. gateup  RMSNorm (folded) + gate/up GEMV + SiLU(gate) * up
//              5. down    down_proj GEMV + residual
//   per step   6. head    final RMSNorm (folded) + output_head restricted to the active vocabulary slice + per-tile arg-max
//              7. pick    arg-max over the tiles (or top-k / top-p / temperature / multinomial) -> next token, ids, pos++
//
// Why this cut (MI355X_MICROARCH.md, price list): every seam above is an all-to-all dependency (each output needs the whole
// previous vector); a dependent kernel boundary costs ~1.2 us, any in-launch exchange across 256 CUs 3-5 us, so the step is
// cut exactly at the all-to-all seams and everything else (norms, RoPE, cache append, SwiGLU, residuals, embedding gather,
// softmax merge, arg-max) is fused into the GEMV that produces or consumes it.  Every GEMV streams its weights once from
// HBM (non-temporal), spread over >= 128 workgroups by narrowing the column tile (NT = 16 / 8 / 4 real columns per MFMA tile)
// instead of splitting K across workgroups (which would need another all-to-all combine).
// All loop-carried scalars (position, ids column, RNG step) live in a small device-side state block that kernel 7 advances,
// so one captured hipGraph of a step can be replayed for every step of a phase (lm.cpp).
#include <cstdlib>

#include "kernels.h"
#include "lm_decode.h"

namespace qa {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// streamed-once data (weights, KV rows): non-temporal 16-byte load (tools/variants.py -DQA_LM_NT=0 builds the cached-load variant)
#ifndef QA_LM_NT
#define QA_LM_NT 1
#endif
__device__ __forceinline__ float4 ldg_nt(const float* p) {
#if QA_LM_NT
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#else
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
#endif
    return make_float4(v.x, v.y, v.z, v.w);
}

// The prefetch plane of a launch (lm_decode.h, PfArgs): workgroup `id` of `count` touches one word per 64 bytes of the consumer tiles
// j = id, id + count, ...  Plain (cached) loads: the lines are meant to stay in this XCD's L2.  Four loads in flight per thread and pass;
// the empty asm consumes the sum so that the loads exist.
__device__ __forceinline__ void lm_prefetch(const PfArgs& pf, int id, int count) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    unsigned acc = 0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (!pf.p[r]) continue;
        const long long tb = pf.tile_bytes[r], step = (long long)nthr * 64;
        for (int j = id; j < pf.n_tiles[r]; j += count) {
            const char* base = pf.p[r] + (long long)j * tb;
            for (long long o = (long long)tid * 64; o < tb; o += 4 * step) {
                unsigned v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long long oo = o + u * step;
                    v[u] = oo < tb ? *reinterpret_cast<const unsigned*>(base + oo) : 0u;
                }
                acc += (v[0] + v[1]) + (v[2] + v[3]);
            }
        }
    }
    asm volatile("" ::"v"(acc));
}
// Kernel-argument preload (r06; build.py compiles with -mllvm -amdgpu-kernarg-preload-count=14): the scalars a working wave needs for its FIRST loads -
// weight pointer, activation pointer, K, M, row-group size, row stride, number of row groups - travel as leading SCALAR kernel arguments, which gfx950
// delivers in SGPRs at wave start; a struct argument is never preloaded, so without this every wave begins with an s_load round trip to the kernel-
// argument segment (and a second one for gridDim.y, a hidden argument) before it can form an address.  16-segment generate 103.4 -> 101.0 ms, 64 segments
// 194.8 -> 190.4, TSE 8 x 503 99.0 -> 97.2 (profiles/r06_lm_prefetch_ab.txt).
// r06, second step: the GEMV kernels form their weight addresses and issue the weight batch from those registers alone, BEFORE they touch the argument
// struct - the struct's scalar loads (row-group pointer shifts of q / k / v caches, RoPE table, residual, state) sat in program order in front of the
// first global_load (an `s_waitcnt lgkmcnt(0)` ahead of it in the ISA), so the preload had bought the scalars but not the head start: 102.7 -> 100.3 ms
// at 16 segments on one box (profiles/r06_lm_prefetch_ab.txt, session 10).  Issuing the ACTIVATION batch from the preloaded registers as well (kp_x: the
// plain rows of x, or the partial-record base of the o_proj launch with S | H << 8 | hd << 16 in kp_ldx) measured 0.3 - 0.4 ms slower than that in all three
// pairs and is not kept; kp_x / kp_ldx still carry those values so that the kernels read neither from the struct.
#define QA_KPRE_PARAMS const float *kp_w, const float *kp_x, int kp_K, int kp_M, int kp_rpg, int kp_ldx, int kp_nrg,
#define QA_KPRE_APPLY(a) \
    a.w = kp_w; a.K = kp_K; a.M = kp_M; a.rpg = kp_rpg;
#define QA_KPRE_APPLY_X(a) \
    a.x = kp_x; a.ldx = kp_ldx;
#define QA_KPRE_ARGS(a, nrg) (a).w, ((a).att_part ? (a).att_part : ((a).tok ? nullptr : (a).x)), (a).K, (a).M, (a).rpg, \
    ((a).att_part ? (int)((a).S | ((a).H << 8) | ((a).hd << 16)) : (int)(a).ldx), (int)(nrg),
#define QA_N_ROW_GROUPS kp_nrg  // gridDim.y is a hidden kernel argument: one more scalar load before the first address
#define QA_LM_PF_PLANE(pf)                                                                   \
    if (__builtin_expect(blockIdx.z != 0, 0)) {                                              \
        lm_prefetch(pf, blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);         \
        return;                                                                              \
    }

#ifdef QA_LM_TIMING  // tuning builds only (tools/variants.py): shader-cycle totals per GEMV kind and phase, wave 0 of every workgroup
__device__ unsigned long long g_lm_timing[6][6];
#define LMT_DECL long long lmt_last = __builtin_readcyclecounter(); const int lmt_kind = (MODE == GM_RESID && ATT) ? 4 : MODE;
#define LMT_TICK(i)                                                                                          \
    {                                                                                                        \
        if ((i) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                       \
        const long long now_ = __builtin_readcyclecounter();                                                 \
        if (threadIdx.x == 0) atomicAdd(&g_lm_timing[lmt_kind][i], (unsigned long long)(now_ - lmt_last));   \
        lmt_last = now_;                                                                                     \
    }
#define LMT_COUNT if (threadIdx.x == 0) atomicAdd(&g_lm_timing[lmt_kind][5], 1ULL);
extern "C" int qa_debug_lm_timing(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lm_timing), sizeof(g_lm_timing)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[36] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_lm_timing), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define LMT_DECL
#define LMT_TICK(i)
#define LMT_COUNT
#endif

// merged attention output for 4 * NF consecutive channels k.. of row `row` (all inside one head since 8 | hd):
// o = sum_s f_s o_s / sum_s f_s l_s with f_s = exp(m_s - max_s m_s); partial record = [o (hd) | m | l | pad2]
template <int NF>
__device__ __forceinline__ void att_merge(const GemvArgs& a, int row, int k, float4* out) {
    constexpr int SMAX = 4;  // launch_lm_attn caps the split count
    const int h = k / a.hd, i = k - h * a.hd, rec = a.hd + 4;
    const float* base = a.att_part + ((long long)(row * a.H + h) * a.S) * rec;
    float ms[SMAX], ls[SMAX];
    float4 o[SMAX][NF];
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {  // every load first (one round trip), then the arithmetic
        const bool on = s < a.S;
        ms[s] = on ? base[s * rec + a.hd] : -INFINITY;
        ls[s] = on ? base[s * rec + a.hd + 1] : 0.f;
#pragma unroll
        for (int f = 0; f < NF; ++f)
            o[s][f] = on ? *reinterpret_cast<const float4*>(base + s * rec + i + 4 * f) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float m = ms[0];
#pragma unroll
    for (int s = 1; s < SMAX; ++s) m = fmaxf(m, ms[s]);
    float L = 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f) out[f] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
        const float w = (ms[s] == -INFINITY) ? 0.f : expf(ms[s] - m);
        L = fmaf(ls[s], w, L);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            out[f].x = fmaf(w, o[s][f].x, out[f].x);
            out[f].y = fmaf(w, o[s][f].y, out[f].y);
            out[f].z = fmaf(w, o[s][f].z, out[f].z);
            out[f].w = fmaf(w, o[s][f].w, out[f].w);
        }
    }
    const float inv = 1.0f / L;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        out[f].x *= inv; out[f].y *= inv; out[f].z *= inv; out[f].w *= inv;
    }
}

// RMSNorm statistics (LlamaRMSNorm: mean of squares over K in fp32; the norm's weight is folded into W) come for free from the
// A-operand registers: every wave adds the squares of its K share per row into s_sq[wave][row] and the epilogue sums the 8 shares.
// Epilogue inputs that do not depend on the GEMV (residual rows, loop state, RoPE table entry) are loaded into registers at kernel
// entry: a dependent load of data another kernel just wrote costs ~2 us on this chip, so a kernel must not chain them.
struct EpiPre {
    float res[4];   // GM_RESID: residual values of this thread's outputs
    int pos;        // GM_QKV: position
    float c[4], s[4];  // GM_QKV: cos / sin of this thread's rotary pairs
};

// Row groups (r05: one chain for up to 64 sequences).  A launch with gridDim.y = RG > 1 serves batch rows [32 rg, 32 rg + 32) in the
// workgroups of its y-plane rg: every pointer that is indexed by the batch row moves to the group's first row and M becomes the group's
// row count, so the kernel bodies below never know about groups.  The weights of a column tile are then read by RG workgroups whose linear
// ids differ by a multiple of gridDim.x - a multiple of 8 for every GEMV of the step, i.e. the SAME XCD (workgroup b runs on XCD b % 8):
// the second reader hits the XCD's L2, HBM still streams every weight once per step, and the step issues 50 launches for 64 sequences
// instead of 2 x 50.  A row's arithmetic does not depend on its group (rows never mix inside a GEMV), so tokens do not change.
constexpr int LM_ROWS_PER_GROUP = 32;
// `rows` = rows per group = 16 MT of the kernel instance that calls it: 32 everywhere except the o_proj launch, which runs 16-row groups
// (MT = 1) as soon as there are more than 16 sequences - its workgroups pull the attention partials of ALL their rows (S records of
// 272 bytes per row and head, 6.4 KB per row at S = 3), and a launch costs ~3.3 us + 0.047 us per KB a workgroup pulls (DESIGN.md
// section 11): 204 KB -> 13 us at 32 rows per workgroup, 102 KB -> 8 us at 16
__device__ __forceinline__ void row_group(GemvArgs& a, int rg, int n_tiles, int rows) {
    const long long r0 = (long long)rg * rows;
    a.M = min(rows, a.M - (int)r0);
    if (a.x) a.x += r0 * a.ldx;
    if (a.tok) a.tok += r0;
    if (a.att_part) a.att_part += r0 * a.H * a.S * (a.hd + 4);
    if (a.q) a.q += r0 * a.d;
    if (a.kc) a.kc += r0 * a.kv_bstride;
    if (a.vc) a.vc += r0 * a.kv_bstride;
    if (a.res) a.res += r0 * a.ldr;
    if (a.res_tok) a.res_tok += r0;
    if (a.y) a.y += r0 * a.ldy;
    if (a.pmax) a.pmax += r0 * n_tiles;
    if (a.pidx) a.pidx += r0 * n_tiles;
    if (a.logits) a.logits += r0 * a.ldl;
}

template <int MT, int NT, int MODE>
__device__ __forceinline__ void epi_prefetch(const GemvArgs& a, int tile, int tid, EpiPre& e) {
    if (MODE == GM_RESID) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + u * 512;
            e.res[u] = 0.f;
            if (i < MT * 16 * NT) {
                const int row = i / NT, j = i - row * NT, n = tile * NT + j;
                if (row < a.M)
                    e.res[u] = a.res_tok ? a.res_table[a.res_tok[row] * a.ldr + n] : a.res[(long long)row * a.ldr + n];
            }
        }
    } else if (MODE == GM_QKV) {
        constexpr int HP = NT / 2;
        e.pos = a.pos >= 0 ? a.pos : a.state[ST_POS];
        const int hd = a.hd, half = hd >> 1, tps = a.d / NT;
        const int sec = tile / tps, c0 = (tile - sec * tps) * NT, h = c0 / hd;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + u * 512;
            e.c[u] = 1.f;
            e.s[u] = 0.f;
            if (i < MT * 16 * HP && sec != 2) {
                const int row = i / HP, j = i - row * HP;
                const int ri = ((c0 - h * hd) / NT) * HP + j;
                e.c[u] = a.rope[((long long)e.pos * half + ri) * 2];
                e.s[u] = a.rope[((long long)e.pos * half + ri) * 2 + 1];
                (void)row;
            }
        }
    }
}

// Fused epilogues over the K-split partial tiles part[wave][m][row][col] (summed in a fixed order: deterministic, no atomics).
template <int MT, int NT, int MODE>
__device__ __forceinline__ void gemv_epilogue(const GemvArgs& a, const float (*part)[MT][16][17], const float (*s_sq)[MT * 16], int tile,
                                              int tid, const EpiPre& e) {
    const int M = a.M;
    auto total = [&](int row, int col) {
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) v += part[wv][row >> 4][row & 15][col];
        return v;
    };
    auto rstd = [&](int row) {
        float sq = 0.f;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) sq += s_sq[wv][row];
        return rsqrtf(sq / a.K + a.rms_eps);
    };
    if (MODE == GM_QKV || MODE == GM_GATEUP) {
        constexpr int HP = NT / 2;  // pairs per tile: columns (j, j + HP)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + u * 512;
            if (i >= MT * 16 * HP) break;
            const int row = i / HP, j = i - row * HP;
            if (row >= M) continue;
            const float rs = rstd(row);
            const float v1 = total(row, j) * rs, v2 = total(row, j + HP) * rs;
            if (MODE == GM_GATEUP) {
                a.y[(long long)row * a.ldy + tile * HP + j] = silu_f(v1) * v2;  // LlamaMLP: down(silu(gate) * up)
            } else {
                const int d = a.d, hd = a.hd, half = hd >> 1, tps = d / NT;
                const int sec = tile / tps, c0 = (tile - sec * tps) * NT;
                const int h = c0 / hd, ri = ((c0 - h * hd) / NT) * HP + j;  // rotary index in [0, hd/2)
                const int pos = e.pos;
                if (sec == 2) {  // V: no rotation
                    float* dst = a.vc + (long long)row * a.kv_bstride + (long long)pos * d + h * hd;
                    dst[ri] = v1;
                    dst[ri + half] = v2;
                } else {  // rotate-half RoPE (LlamaRotaryEmbedding / apply_rotary_pos_emb)
                    const float c = e.c[u], sn = e.s[u];
                    float* dst = sec == 0 ? a.q + (long long)row * d + h * hd
                                          : a.kc + (long long)row * a.kv_bstride + (long long)pos * d + h * hd;
                    dst[ri] = v1 * c - v2 * sn;
                    dst[ri + half] = v2 * c + v1 * sn;
                }
            }
        }
    } else if (MODE == GM_RESID) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + u * 512;
            if (i >= MT * 16 * NT) break;
            const int row = i / NT, j = i - row * NT;
            if (row >= M) continue;
            a.y[(long long)row * a.ldy + tile * NT + j] = total(row, j) + e.res[u];
        }
    } else {  // GM_HEAD: logits of this tile's NT vocabulary entries and their per-row maximum (first maximum wins)
        for (int i = tid; i < MT * 16 * NT; i += 512) {  // MT * 16 * NT <= 512: one pass, whole NT-lane groups stay converged
            const int row = i / NT, j = i - row * NT;
            const int n = tile * NT + j;
            float v = -INFINITY;
            if (row < M) {
                v = total(row, j) * rstd(row);
                if (a.logits) a.logits[(long long)row * a.ldl + n] = v;
            }
            int bi = n;
#pragma unroll
            for (int o = NT >> 1; o > 0; o >>= 1) {
                const float v2 = __shfl_xor(v, o, 64);
                const int i2 = __shfl_xor(bi, o, 64);
                if (v2 > v || (v2 == v && i2 < bi)) {
                    v = v2;
                    bi = i2;
                }
            }
            if (j == 0 && row < M) {
                a.pmax[(long long)row * gridDim.x + tile] = v;
                a.pidx[(long long)row * gridDim.x + tile] = bi;
            }
        }
    }
}

// y[M <= 16*MT, tile of NT columns] = epi(A[M, K] W_tile[NT, K]^T): one workgroup per column tile, its 8 waves split K, every
// lane streams 32-byte pieces of its weight row straight from HBM into v_mfma_f32_16x16x4_f32 (batch rows = M side; with
// NT < 16 the surplus MFMA columns duplicate real ones and are ignored - the 4x4x1 kernel below is the one used for narrow
// tiles), partial tiles are summed through LDS, then the mode's fused epilogue runs.
// NB = 32-wide K chunks per wave that are loaded as one batch (all of a wave's K share when K / 256 <= 8): every weight and
// activation load of the batch is issued before the first MFMA, so a wave pays one memory round trip per batch instead of one per
// chunk (hipcc does not software-pipeline the chunk loop by itself).
template <int MT, int NT, int MODE, bool ATT, int NB>
__global__ __launch_bounds__(512) void lm_gemv_kernel(QA_KPRE_PARAMS const GemvArgs a_in, const PfArgs pf) {
    QA_LM_PF_PLANE(pf)
    __shared__ float part[8][MT][16][17];
    __shared__ float s_sq[8][MT * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int tile = blockIdx.x;
    LMT_DECL
    // ---- from the preloaded scalars alone (no scalar load yet): the row group and the weight batch
    const int K = kp_K;
    const int rows_pg = kp_rpg ? kp_rpg : 16 * MT;
    const int r0 = kp_nrg > 1 ? (int)blockIdx.y * rows_pg : 0;
    const int M = kp_nrg > 1 ? min(rows_pg, kp_M - r0) : kp_M;
    const int kw = K >> 3, k0 = wave * kw, nchunk = kw >> 5;
    const int kbase = k0 + 8 * kq;
    const float* wp = kp_w + ((long long)tile * NT + (li & (NT - 1))) * K + k0 + 8 * kq;
    // weights do not depend on anything computed here: get the first batch in flight before touching the activations
    float4 wr[NB][2];
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        wr[c][0] = ldg_nt(wp + c * 32);
        wr[c][1] = ldg_nt(wp + c * 32 + 4);
    }
    float4 xa[MT][NB][2];
    __builtin_amdgcn_sched_barrier(0);  // the argument struct (scalar loads) only behind the weight loads
    GemvArgs a = a_in;
    QA_KPRE_APPLY(a)
    if (ATT) { a.S = kp_ldx & 255; a.H = (kp_ldx >> 8) & 255; a.hd = kp_ldx >> 16; a.att_part = kp_x; } else { a.x = kp_x; a.ldx = kp_ldx; }
    if (QA_N_ROW_GROUPS > 1) row_group(a, blockIdx.y, gridDim.x, rows_pg);
    const float* xrow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = min(m * 16 + li, M - 1);
        xrow[m] = a.tok ? a.table + a.tok[row] * (long long)kp_ldx : a.x + (long long)row * a.ldx;
    }
    EpiPre epi;
    epi_prefetch<MT, NT, MODE>(a, tile, tid, epi);
    f32x4 acc[MT];
    float sq[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        sq[m] = 0.f;
    }
    for (int c0 = 0; c0 < nchunk; c0 += NB) {
        if (c0 > 0) {
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                wr[c][0] = ldg_nt(wp + (c0 + c) * 32);
                wr[c][1] = ldg_nt(wp + (c0 + c) * 32 + 4);
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                if (ATT) {
                    att_merge<2>(a, min(m * 16 + li, M - 1), kbase + (c0 + c) * 32, xa[m][c]);
                } else {
                    xa[m][c][0] = *reinterpret_cast<const float4*>(xrow[m] + kbase + (c0 + c) * 32);
                    xa[m][c][1] = *reinterpret_cast<const float4*>(xrow[m] + kbase + (c0 + c) * 32 + 4);
                }
            }
        __builtin_amdgcn_sched_barrier(0);  // keep every load of the batch ahead of the first MFMA (hipcc sinks them one by one)
        LMT_TICK(0)
#pragma unroll
        for (int c = 0; c < NB; ++c)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float4 a0 = xa[m][c][0], a1 = xa[m][c][1], w0 = wr[c][0], w1 = wr[c][1];
                if (MODE != GM_RESID)
                    sq[m] += (a0.x * a0.x + a0.y * a0.y + a0.z * a0.z + a0.w * a0.w) + (a1.x * a1.x + a1.y * a1.y + a1.z * a1.z + a1.w * a1.w);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, w0.x, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, w0.y, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, w0.z, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, w0.w, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, w1.x, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, w1.y, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, w1.z, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, w1.w, acc[m], 0, 0, 0);
            }
    }
    LMT_TICK(1)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][m][4 * kq + r][li] = acc[m][r];
    if (MODE != GM_RESID) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float v = sq[m];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (kq == 0) s_sq[wave][m * 16 + li] = v;
        }
    }
    __syncthreads();
    LMT_TICK(2)
    gemv_epilogue<MT, NT, MODE>(a, part, s_sq, tile, tid, epi);
    LMT_TICK(3)
    LMT_COUNT
}

// The same GEMV for NARROW column tiles (NT = 4 * C columns, C = 1 or 2) on v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4x1
// blocks per instruction.  Block (g, p) = lane >> 2 multiplies batch rows 4g..4g+3 (A: lane & 3 selects the row) with the tile's
// columns (B: lane & 3 selects the column) at K phase p: a lane's float4 holds k = 16 * step + 4p .. + 3, one component per MFMA.
// Every FMA is useful (the 16x16x4 form wastes 16 / NT of the matrix pipe on duplicated columns: 1.3 us per down_proj
// workgroup); the four K phases are summed with two shuffles at the end.  NS = 16-wide K steps per wave loaded as one batch.
// R8 (r05, the o_proj launch): EIGHT batch rows per workgroup - row blocks g = 0, 1 only, and lane bit 5 becomes a third K-phase bit
// (8 phases of 4 k, 32-wide steps), so a workgroup pulls the attention partials of 8 rows instead of 16 (row_group above).  A row's
// products are the same set in the same per-lane order; only the cross-lane fold gains one shuffle.
template <int MT, int C, int MODE, bool ATT, int NS, bool R8 = false>
__global__ __launch_bounds__(512) void lm_gemv4_kernel(QA_KPRE_PARAMS const GemvArgs a_in, const PfArgs pf) {
    static_assert(!R8 || MT == 1, "8-row groups exist for one row tile");
    QA_LM_PF_PLANE(pf)
    constexpr int NT = 4 * C;
    __shared__ float part[8][MT][16][17];
    __shared__ float s_sq[8][MT * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int KS = R8 ? 32 : 16;  // k per step
    const int g = R8 ? (lane >> 4) & 1 : lane >> 4, p = R8 ? ((lane >> 2) & 3) + 4 * (lane >> 5) : (lane >> 2) & 3, i4 = lane & 3;
    const int tile = blockIdx.x;
    LMT_DECL
    // ---- from the preloaded scalars alone (see lm_gemv_kernel): row group, weight batch
    const int K = kp_K;
    constexpr int rows_pg = R8 ? 8 : 16 * MT;
    const int r0 = kp_nrg > 1 ? (int)blockIdx.y * rows_pg : 0;
    const int M = kp_nrg > 1 ? min(rows_pg, kp_M - r0) : kp_M;
    const int kw = K >> 3, k0 = wave * kw, nstep = kw / KS;
    const int kbase = k0 + 4 * p;
    const float* wp[C];
#pragma unroll
    for (int c = 0; c < C; ++c) wp[c] = kp_w + ((long long)tile * NT + c * 4 + i4) * K + kbase;
    float4 wr[NS][C];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int c = 0; c < C; ++c) wr[s][c] = ldg_nt(wp[c] + s * KS);
    float4 xa[MT][NS];
    __builtin_amdgcn_sched_barrier(0);  // the argument struct (scalar loads) only behind the weight loads
    GemvArgs a = a_in;
    QA_KPRE_APPLY(a)
    if (ATT) { a.S = kp_ldx & 255; a.H = (kp_ldx >> 8) & 255; a.hd = kp_ldx >> 16; a.att_part = kp_x; } else { a.x = kp_x; a.ldx = kp_ldx; }
    if (QA_N_ROW_GROUPS > 1) row_group(a, blockIdx.y, gridDim.x, rows_pg);
    const float* xrow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = min(m * 16 + 4 * g + i4, M - 1);
        xrow[m] = a.tok ? a.table + a.tok[row] * (long long)kp_ldx : a.x + (long long)row * a.ldx;
    }
    EpiPre epi;
    epi_prefetch<MT, NT, MODE>(a, tile, tid, epi);
    f32x4 acc[MT][C];
    float sq[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        sq[m] = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) acc[m][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int s0 = 0; s0 < nstep; s0 += NS) {
        if (s0 > 0) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int c = 0; c < C; ++c) wr[s][c] = ldg_nt(wp[c] + (s0 + s) * KS);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (ATT) att_merge<1>(a, min(m * 16 + 4 * g + i4, M - 1), kbase + (s0 + s) * KS, &xa[m][s]);
                else xa[m][s] = *reinterpret_cast<const float4*>(xrow[m] + kbase + (s0 + s) * KS);
            }
        __builtin_amdgcn_sched_barrier(0);
        LMT_TICK(0)
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (MODE != GM_RESID) {
                    const float4 x4 = xa[m][s];
                    sq[m] += x4.x * x4.x + x4.y * x4.y + x4.z * x4.z + x4.w * x4.w;
                }
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float4 x4 = xa[m][s], w4 = wr[s][c];
                    acc[m][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(x4.x, w4.x, acc[m][c], 0, 0, 0);
                    acc[m][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(x4.y, w4.y, acc[m][c], 0, 0, 0);
                    acc[m][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(x4.z, w4.z, acc[m][c], 0, 0, 0);
                    acc[m][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(x4.w, w4.w, acc[m][c], 0, 0, 0);
                }
            }
    }
    LMT_TICK(1)
    // D of block (g, p): VGPR r = batch row 4g + r, lane & 3 = column; sum the 4 K phases (lane bits 2, 3)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[m][c][r];
                v += __shfl_xor(v, 4, 64);
                v += __shfl_xor(v, 8, 64);
                if (R8) v += __shfl_xor(v, 32, 64);
                if (p == 0) part[wave][m][4 * g + r][c * 4 + i4] = v;
            }
    if (MODE != GM_RESID) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float v = sq[m];
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 8, 64);
            if (R8) v += __shfl_xor(v, 32, 64);
            if (p == 0) s_sq[wave][m * 16 + 4 * g + i4] = v;
        }
    }
    __syncthreads();
    LMT_TICK(2)
    gemv_epilogue<MT, NT, MODE>(a, part, s_sq, tile, tid, epi);
    LMT_TICK(3)
    LMT_COUNT
}

// column-tile width: as wide as possible while the launch still spreads over >= 96 workgroups (measured on MI355X at B = 16:
// qkv 96 x 16 columns beats 192 x 8 by 3 %, o_proj / down_proj 128 x 4 beat 64 x 8 by 3-9 %, gate/up 256 x 16 beats 512 x 8 by 9 %)
int lm_pick_nt(int N) {
    int nt = 16;
    while (nt > 4 && N / nt < 96) nt >>= 1;
    return nt;
}

// 32-wide chunks per batch of the 16x16x4 kernel for a given K: the whole per-wave share when it is at most 8 chunks (K <= 2048)
static int gemv_nb(int K) {
    const int nchunk = K / 256;
    if (nchunk == 1 || nchunk == 2 || nchunk == 4 || nchunk == 8) return nchunk;
    return nchunk % 8 == 0 ? 8 : 0;
}

// the o_proj launch widens its column tile with the batch; widths 4 / 8 / 16 share one per-row summation order only on the
// 8-row kernel (lm_gemv4_kernel R8, taken when a wave's K share is an even number of 32-wide chunks) - where it does not exist (hidden 256)
// width 16 would fall to the 16x16x4 kernel with another K partition, so lm.cpp keeps the model's width there
bool gemv_r8_ok(int K) { return gemv_nb(K) > 0 && gemv_nb(K) % 2 == 0; }

// the prefetch plane: a second z-plane of workgroups when the launch carries prefetch work (lm_decode.h, PfArgs)
static dim3 with_pf_plane(dim3 g, const PfArgs& pf) {
    if (pf.p[0] || pf.p[1]) g.z = 2;
    return g;
}

template <int MODE, bool ATT, int NB>
static int launch_gemv_nb(const GemvArgs& a_in, int nt, hipStream_t s, const PfArgs* pf_in) {
    GemvArgs a = a_in;
    PfArgs pf = pf_in ? *pf_in : PfArgs{};
    // y: row groups.  16 rows each (MT = 1) for up to 16 sequences (8 for the o_proj launch, ATT: see row_group), 32 rows (MT = 2) otherwise
    // ... and for the qkv launch up to 32 sequences (139.4 -> 136.0 ms at 32 segments; at 64 segments 32-row groups win: 196 vs 198 ms)
    const bool mt1 = a.M <= 16 || ATT || (MODE == GM_QKV && a.M <= 32);
    if (MODE == GM_QKV && a.M > 8 && a.M <= 16 && nt == 16 && (knob(K_LM_ROWSPLIT) & 1)) {  // 9 .. 16 sequences: two 8-row groups (see launch_lm_mlp)
        a.rpg = 8;
        const dim3 grid = with_pf_plane(dim3((unsigned)(a.N / nt), (unsigned)ceil_div(a.M, 8)), pf);
        hipLaunchKernelGGL((lm_gemv_kernel<1, 16, MODE, ATT, NB>), grid, dim3(512), 0, s, QA_KPRE_ARGS(a, grid.y) a, pf);
        QA_LAUNCH_CHECK();
        return QA_OK;
    }
    if constexpr (ATT) {  // the o_proj launch on narrow tiles: 8-row groups (lm_gemv4_kernel R8)
        if (NB % 2 == 0) {
            const dim3 grid8 = with_pf_plane(dim3((unsigned)(a.N / nt), (unsigned)ceil_div(a.M, 8)), pf);
            if (nt == 16) hipLaunchKernelGGL((lm_gemv4_kernel<1, 4, MODE, ATT, NB, true>), grid8, dim3(512), 0, s, QA_KPRE_ARGS(a, grid8.y) a, pf);
            else if (nt == 8) hipLaunchKernelGGL((lm_gemv4_kernel<1, 2, MODE, ATT, NB, true>), grid8, dim3(512), 0, s, QA_KPRE_ARGS(a, grid8.y) a, pf);
            else hipLaunchKernelGGL((lm_gemv4_kernel<1, 1, MODE, ATT, NB, true>), grid8, dim3(512), 0, s, QA_KPRE_ARGS(a, grid8.y) a, pf);
            QA_LAUNCH_CHECK();
            return QA_OK;
        }
    }
    const dim3 grid = with_pf_plane(dim3((unsigned)(a.N / nt), (unsigned)ceil_div(a.M, mt1 ? 16 : LM_ROWS_PER_GROUP)), pf);
    // narrow tiles (NT = 8 / 4) run on the 4x4x1 MFMA, where every FMA is useful (the 16x16x4 form on duplicated columns measured equal:
    // the matrix pipe is not what bounds the step, DESIGN_HISTORY.md section 7a)
#define QA_GV(MT, NT) hipLaunchKernelGGL((lm_gemv_kernel<MT, NT, MODE, ATT, NB>), grid, dim3(512), 0, s, QA_KPRE_ARGS(a, grid.y) a, pf)
#define QA_G4(MT, C) hipLaunchKernelGGL((lm_gemv4_kernel<MT, C, MODE, ATT, 2 * NB>), grid, dim3(512), 0, s, QA_KPRE_ARGS(a, grid.y) a, pf)
    if (mt1) {
        if (nt == 16) QA_GV(1, 16);
        else if (nt == 8) QA_G4(1, 2);
        else QA_G4(1, 1);
    } else {
        if (nt == 16) QA_GV(2, 16);
        else if (nt == 8) QA_G4(2, 2);
        else QA_G4(2, 1);
    }
#undef QA_GV
#undef QA_G4
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// K = hidden for every mode but the down projection (K = intermediate = 4 * hidden): only the batch sizes those shapes need exist
template <int MODE, bool ATT>
static int launch_gemv_mode(const GemvArgs& a, int nt, hipStream_t s, const PfArgs* pf) {
    const int nb = gemv_nb(a.K);
    constexpr bool wide = MODE == GM_RESID && !ATT;  // down_proj
    if constexpr (!wide) {
        if (nb == 1) return launch_gemv_nb<MODE, ATT, 1>(a, nt, s, pf);
        if (nb == 2) return launch_gemv_nb<MODE, ATT, 2>(a, nt, s, pf);
        if (nb == 4) return launch_gemv_nb<MODE, ATT, 4>(a, nt, s, pf);
    } else {
        if (nb == 4) return launch_gemv_nb<MODE, ATT, 4>(a, nt, s, pf);
        if (nb == 8) return launch_gemv_nb<MODE, ATT, 8>(a, nt, s, pf);
    }
    set_error("lm_gemv: K=%d has no kernel instance (mode %d)", a.K, MODE);
    return QA_ERR_UNSUPPORTED;
}

bool lm_gemv_supported(int hidden, int intermediate) {
    const int nb_d = gemv_nb(hidden), nb_i = gemv_nb(intermediate);
    return hidden % 256 == 0 && intermediate % 256 == 0 && (nb_d == 1 || nb_d == 2 || nb_d == 4) && (nb_i == 4 || nb_i == 8);
}

int launch_lm_gemv(const GemvArgs& a, int mode, int nt, hipStream_t s, const PfArgs* pf) {
    QA_REQUIRE(a.M >= 1 && a.M <= LM_MAX_ROWS, "lm_gemv: M=%d must be in [1, %d]", a.M, LM_MAX_ROWS);
    QA_REQUIRE(a.K % 256 == 0 && (a.ldx % 4) == 0, "lm_gemv: K=%d must be a multiple of 256", a.K);
    QA_REQUIRE((nt == 16 || nt == 8 || nt == 4) && a.N % nt == 0, "lm_gemv: N=%d not a multiple of the tile width %d", a.N, nt);
    // the o_proj launch hands (S, H, hd) to its kernel packed into one preloaded scalar (QA_KPRE_ARGS)
    QA_REQUIRE(!a.att_part || (a.S >= 1 && a.S < 256 && a.H >= 1 && a.H < 256 && a.hd >= 1 && a.hd < 32768), "lm_gemv: S=%d H=%d hd=%d out of range", a.S, a.H, a.hd);
    switch (mode) {
        case GM_QKV: return launch_gemv_mode<GM_QKV, false>(a, nt, s, pf);
        case GM_GATEUP: return launch_gemv_mode<GM_GATEUP, false>(a, nt, s, pf);
        case GM_RESID: return a.att_part ? launch_gemv_mode<GM_RESID, true>(a, nt, s, pf) : launch_gemv_mode<GM_RESID, false>(a, nt, s, pf);
        case GM_HEAD: return launch_gemv_mode<GM_HEAD, false>(a, nt, s, pf);
        default: set_error("lm_gemv: bad mode %d", mode); return QA_ERR_INVALID;
    }
}

// ------------------------------------------------------------------------------------------------
// Fused MLP of the decode step (QA_LM_MLP_FUSED): gate/up GEMV + SwiGLU + the down projection of the SAME 16 activation columns in
// one launch, then one small reduce launch - instead of gate/up (256 workgroups) and down (128 workgroups that each pull the whole
// [M, 4d] activation, 128 KB, through one CU).  Workgroup j owns activation columns [16 j, 16 j + 16): it needs x (32 KB), its 32
// gate / up rows (64 KB) and the 16-column slice of W_down (32 KB, re-laid out slice-major at load time); `act` never exists in HBM.
// What the K split of the down projection costs is a cross-workgroup sum: every workgroup writes its [M, d] partial (4 MB in all at
// M = 16) and lm_mlp_reduce_kernel adds the I / 16 partials and the residual in a fixed order (deterministic, no atomics).
// Phase 1 is lm_gemv_kernel<MT, 16, GM_GATEUP> on two adjacent column tiles of the gate/up decode layout; phase 2 runs the K = 16
// product on v_mfma_f32_16x16x4_f32 straight from an LDS copy of the activation tile, W_down slice prefetched at kernel entry.
// AC = activation columns per workgroup: 16 (two gate/up tiles, I / 16 partials) or 8 (one tile, I / 8 partials, twice the workgroups)
template <int MT, int NB, int AC>
__global__ __launch_bounds__(512) void lm_mlp_kernel(QA_KPRE_PARAMS const float* __restrict__ wd, const GemvArgs a_in, float* __restrict__ partial,
                                                     const PfArgs pf) {
    QA_LM_PF_PLANE(pf)
    GemvArgs a = a_in;
    QA_KPRE_APPLY(a)
    QA_KPRE_APPLY_X(a)
    if (QA_N_ROW_GROUPS > 1) row_group(a, blockIdx.y, gridDim.x, a.rpg ? a.rpg : 16 * MT);
    constexpr int NTL = AC / 8;  // gate/up decode tiles (8 gate + 8 up rows each) per workgroup
    __shared__ float part[8][NTL][MT][16][17];
    __shared__ float s_sq[8][MT * 16];
    __shared__ __attribute__((aligned(16))) float s_act[MT * 16][20];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int j = blockIdx.x;
    const int K = a.K, M = a.M, d = a.K;  // launch_lm_mlp requires K == d; K is a preloaded argument
    const int kw = K >> 3, k0 = wave * kw;
    // every load of the workgroup in flight before the first MFMA: gate / up rows, x, and the W_down slice of phase 2
    const float* wp[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) wp[t] = a.w + ((long long)(NTL * j + t) * 16 + li) * K + k0 + 8 * kq;
    float4 wr[NTL][NB][2];
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            wr[t][c][0] = ldg_nt(wp[t] + c * 32);
            wr[t][c][1] = ldg_nt(wp[t] + c * 32 + 4);
        }
    const int ntw = d >> 7;  // 16-column output tiles per wave in phase 2 (d / 8 columns per wave)
    constexpr int KL = AC / 4;  // W_down values per lane and output tile: k = KL kq + i
    float wdr[4][KL];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int n = min(wave * (d >> 3) + t * 16 + li, d - 1);
        const float* src = wd + ((long long)j * d + n) * AC + KL * kq;
        if (KL == 4) {
            const float4 v = ldg_nt(src);
            wdr[t][0] = v.x; wdr[t][1] = v.y; wdr[t][KL - 2] = v.z; wdr[t][KL - 1] = v.w;
        } else {
            const float2 v = *reinterpret_cast<const float2*>(src);
            wdr[t][0] = v.x; wdr[t][1] = v.y;
        }
    }
    float4 xa[MT][NB][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const float* xrow = a.x + (long long)min(m * 16 + li, M - 1) * a.ldx + k0 + 8 * kq;
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            xa[m][c][0] = *reinterpret_cast<const float4*>(xrow + c * 32);
            xa[m][c][1] = *reinterpret_cast<const float4*>(xrow + c * 32 + 4);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[NTL][MT];
    float sq[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        sq[m] = 0.f;
#pragma unroll
        for (int t = 0; t < NTL; ++t) acc[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int c = 0; c < NB; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float4 a0 = xa[m][c][0], a1 = xa[m][c][1];
            sq[m] += (a0.x * a0.x + a0.y * a0.y + a0.z * a0.z + a0.w * a0.w) + (a1.x * a1.x + a1.y * a1.y + a1.z * a1.z + a1.w * a1.w);
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                const float4 w0 = wr[t][c][0], w1 = wr[t][c][1];
                acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, w0.x, acc[t][m], 0, 0, 0);
                acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, w0.y, acc[t][m], 0, 0, 0);
                acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, w0.z, acc[t][m], 0, 0, 0);
                acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, w0.w, acc[t][m], 0, 0, 0);
                acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, w1.x, acc[t][m], 0, 0, 0);
                acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, w1.y, acc[t][m], 0, 0, 0);
                acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, w1.z, acc[t][m], 0, 0, 0);
                acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, w1.w, acc[t][m], 0, 0, 0);
            }
        }
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[wave][t][m][4 * kq + r][li] = acc[t][m][r];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        float v = sq[m];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (kq == 0) s_sq[wave][m * 16 + li] = v;
    }
    __syncthreads();
    // SwiGLU of the 16 x (MT * 16) activation tile: column c of the tile = pair (c & 7, (c & 7) + 8) of weight tile c >> 3, summed over
    // the 8 K shares in the order lm_gemv_kernel's epilogue uses (bit-identical to the two-launch path's activation)
    if (tid < MT * 16 * AC) {
        const int row = tid / AC, c = tid % AC, t = c >> 3, p = c & 7;
        float sqs = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) {
            sqs += s_sq[wv][row];
            v1 += part[wv][t][row >> 4][row & 15][p];
            v2 += part[wv][t][row >> 4][row & 15][p + 8];
        }
        const float rs = rsqrtf(sqs / K + a.rms_eps);
        s_act[row][c] = silu_f(v1 * rs) * (v2 * rs);
    }
    __syncthreads();
    // phase 2: partial[j][row][n] = sum_{k < 16} act[row][k] * W_down[n][16 j + k].  Operands swapped (W is the MFMA's row operand):
    // D^T[n][row], so a lane ends up with 4 CONSECUTIVE output columns n = 4 kq .. 4 kq + 3 of batch row li - one 16-byte store per tile
    float* pj = partial + ((long long)blockIdx.y * gridDim.x + j) * (MT * 16) * d;  // [row group][j][MT * 16 rows][d]
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        float av[KL];
#pragma unroll
        for (int i = 0; i < KL; ++i) av[i] = s_act[m * 16 + li][KL * kq + i];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t >= ntw) break;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < KL; ++i) o = __builtin_amdgcn_mfma_f32_16x16x4f32(wdr[t][i], av[i], o, 0, 0, 0);
            const int n = wave * (d >> 3) + t * 16 + 4 * kq;
            if (m * 16 + li < M) *reinterpret_cast<f32x4*>(pj + (long long)(m * 16 + li) * d + n) = o;  // rows past the group's M are never read back
        }
    }
}

// x[row][col] = res[row][col] + sum_j partial[j][row][col]: ONE WAVE per (row, 32-column block), no LDS, no barrier: lane = (group g of 8,
// 4-column slot l8); a lane adds the partials j = g, g + 8, ... (RJ loads in flight at once), then the 8 groups fold with three xor
// shuffles - a fixed order: deterministic.
template <int RJ>
__global__ __launch_bounds__(64) void lm_mlp_reduce_kernel(const float* __restrict__ partial, int n_part, int m_pad, int rpg, int d,
                                                           const float* __restrict__ res, long long ldr, float* __restrict__ y, long long ldy) {
    const int lane = threadIdx.x, g = lane >> 3, l8 = lane & 7;
    const int row = blockIdx.y, col = blockIdx.x * 32 + 4 * l8;
    const int rg = row / rpg, rl = row - rg * rpg;  // row group (rpg rows each, in slabs of m_pad rows) and row inside it
    const float* p0 = partial + ((long long)rg * n_part * m_pad + rl) * d + col;
    const long long pstride = (long long)m_pad * d;
    f32x4 res4 = {0.f, 0.f, 0.f, 0.f};
    if (g == 0) res4 = *reinterpret_cast<const f32x4*>(res + (long long)row * ldr + col);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int j0 = g; j0 < n_part; j0 += 8 * RJ) {
        f32x4 t[RJ];
#pragma unroll
        for (int u = 0; u < RJ; ++u) {
            const int jj = j0 + 8 * u;
            t[u] = jj < n_part ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p0 + jj * pstride)) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < RJ; ++u) v += t[u];
    }
#pragma unroll
    for (int of = 8; of < 64; of <<= 1) {
        v.x += __shfl_xor(v.x, of, 64);
        v.y += __shfl_xor(v.y, of, 64);
        v.z += __shfl_xor(v.z, of, 64);
        v.w += __shfl_xor(v.w, of, 64);
    }
    if (g == 0) *reinterpret_cast<f32x4*>(y + (long long)row * ldy + col) = res4 + v;
}

bool lm_mlp_fused_supported(int d, int I, int nt_gu) {
    return nt_gu == 16 && I % 16 == 0 && d % 128 == 0 && d <= 512 && (d == 256 || d == 512);
}

// a: x / ldx / w (gate-up decode layout, NT = 16) / M / K = d / d / rms_eps; wd: W_down in slice-major layout [I / 16][d][16];
// partial: [I / 16][16 * MT][d] scratch; y = res + down(act)
int launch_lm_mlp(const GemvArgs& a, int I, int ac, const float* wd, float* partial, const float* res, long long ldr, float* y, long long ldy,
                  hipStream_t s, const PfArgs* pf_in) {
    QA_REQUIRE(a.M >= 1 && a.M <= LM_MAX_ROWS && a.K == a.d && lm_mlp_fused_supported(a.d, I, 16), "lm_mlp: unsupported shape M=%d d=%d I=%d", a.M, a.d, I);
    // 16-row groups up to 32 sequences (two groups x 128 workgroups: each pulls 32 KB of x instead of 64 beside its 96 KB of weights -
    // 144.0 -> 139.4 ms per generate at 32 segments), 32-row groups above (at 64 segments four groups = 512 workgroups lose: 207 vs 196 ms)
    const int mt = a.M <= 32 ? 1 : 2;
    const int n_part = I / ac;
    // 9 .. 16 sequences: two groups of 8 rows on the 16-row tile (rows 8 .. 15 of a group re-read its last row): x 16 KB per workgroup
    // instead of 32; with the same split of the qkv launch 111.0 -> 109.7 ms per generate at 16 segments
    const int rpg = (a.M > 8 && a.M <= 16 && (knob(K_LM_ROWSPLIT) & 2)) ? 8 : 16 * mt;
    GemvArgs ag = a;
    ag.rpg = rpg;
    PfArgs pf = pf_in ? *pf_in : PfArgs{};
    const dim3 grid = with_pf_plane(dim3((unsigned)n_part, (unsigned)ceil_div(a.M, rpg)), pf);  // partial: [row group][n_part][16 mt][d]
    QA_REQUIRE(ac == 16, "lm_mlp: %d activation columns per workgroup (only 16 is built)", ac);
#define QA_MLP(MT, NB) hipLaunchKernelGGL((lm_mlp_kernel<MT, NB, 16>), grid, dim3(512), 0, s, QA_KPRE_ARGS(ag, grid.y) wd, ag, partial, pf)
    if (a.d == 512) {
        if (mt == 1) { QA_MLP(1, 2); } else { QA_MLP(2, 2); }
    } else {
        if (mt == 1) { QA_MLP(1, 1); } else { QA_MLP(2, 1); }
    }
#undef QA_MLP
    QA_LAUNCH_CHECK();
    hipLaunchKernelGGL((lm_mlp_reduce_kernel<16>), dim3((unsigned)(a.d / 32), (unsigned)a.M), dim3(64), 0, s, partial, n_part, 16 * mt, rpg, a.d, res,
                       ldr, y, ldy);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// ------------------------------------------------------------------------------------------------
// Single-query attention over the KV cache.  The 16-key tiles of a (sequence, head) are dealt round-robin to the S workgroups of
// its split and their NW waves; a wave keeps two tiles (its K and V rows) in flight.  Lane map: LPK = HD / 4 lanes cover one key
// row with one float4 each, so a load instruction reads 64 / LPK whole rows of 4 * HD contiguous bytes (full cache lines; the
// first version gave each lane 64 contiguous bytes and touched 32 lines per instruction for 1 KB of payload).  Scores are
// reduced over the LPK lanes of a key with shuffles, softmax is online per wave, the NW wave states are merged through LDS into
// one partial record [o (HD, un-normalised, relative to m) | m | l | pad2] that the o_proj GEMV merges across the S splits
// (att_merge).  n_keys = pos + 1 (the new key included): `pos` is a launch argument when the host drives the loop, or read from the
// device-side loop state (pos < 0) when a captured step is replayed.
// DEVPOS: the position comes from the device-side loop state (a captured step that is replayed); the eager path passes it as an argument and must not
// touch `state` at all - hipcc otherwise loads state[ST_POS] speculatively: two dependent scalar loads in front of the first K / V address.
// Argument order: everything the first K / V loads need sits in the first 14 dwords (kernel-argument preload, see QA_KPRE_PARAMS above).
template <int HD, int NW, bool DEVPOS>
__global__ __launch_bounds__(NW * 64) void lm_attn_kernel(const float* __restrict__ q, const float* __restrict__ kc, const float* __restrict__ vc,
                                                          long long kv_bstride, int pos, int ldq, int ldkv, int S, int H, float scale,
                                                          float* __restrict__ part, const int* __restrict__ state) {  // S, H = gridDim.z, .x (hidden arguments otherwise)
    constexpr int LPK = HD / 4;    // lanes per key row
    constexpr int KPI = 64 / LPK;  // key rows per load instruction
    constexpr int NI = 16 / KPI;   // load instructions per 16-key tile (per operand)
    __shared__ float s_m[NW], s_l[NW];
    __shared__ __attribute__((aligned(16))) float s_o[NW][HD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, h = blockIdx.x, sp = blockIdx.z;
    const int kq = lane / LPK, c4 = (lane % LPK) * 4;
    const int n_keys = (DEVPOS ? state[ST_POS] : pos) + 1;
    const int n_tiles = (n_keys + 15) >> 4;
    const float* kb = kc + (long long)b * kv_bstride + h * HD + c4;
    const float* vb = vc + (long long)b * kv_bstride + h * HD + c4;
    float4 qv = *reinterpret_cast<const float4*>(q + (long long)b * ldq + h * HD + c4);
    qv.x *= scale; qv.y *= scale; qv.z *= scale; qv.w *= scale;
    float m_run = -INFINITY, l_run = 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0;; it += 2) {
        const int ta = (it * NW + wave) * S + sp, tb = ((it + 1) * NW + wave) * S + sp;
        if (ta >= n_tiles) break;
        float4 kt[2][NI], vt[2][NI];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = u == 0 ? ta : min(tb, n_tiles - 1);  // a missing second tile re-reads a valid one and is skipped below
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int row = min(t * 16 + j * KPI + kq, n_keys - 1);  // clamp: rows past n_keys are uninitialised cache memory
                kt[u][j] = ldg_nt(kb + (long long)row * ldkv);
                vt[u][j] = ldg_nt(vb + (long long)row * ldkv);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = u == 0 ? ta : tb;
            if (t >= n_tiles) break;
            float sc[NI];
            float tmax = -INFINITY;
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
            const float m_new = fmaxf(m_run, tmax);  // every processed tile has at least one valid key
            const float alpha = expf(m_run - m_new);
            float psum = 0.f;
            o.x *= alpha; o.y *= alpha; o.z *= alpha; o.w *= alpha;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const float p = expf(sc[j] - m_new);  // exp(-inf) = 0 for masked keys (their V row is a clamped, valid row)
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
    for (int of = LPK; of < 64; of <<= 1) {  // sum the key groups that share a column slice
        o.x += __shfl_xor(o.x, of, 64);
        o.y += __shfl_xor(o.y, of, 64);
        o.z += __shfl_xor(o.z, of, 64);
        o.w += __shfl_xor(o.w, of, 64);
    }
    if (lane < LPK) {
        *reinterpret_cast<float4*>(&s_o[wave][c4]) = o;
        if (lane == 0) {
            s_m[wave] = m_run;
            s_l[wave] = l_run;
        }
    }
    __syncthreads();
    if (tid < HD + 2) {
        float m = s_m[0];
        for (int w = 1; w < NW; ++w) m = fmaxf(m, s_m[w]);
        float l = 0.f, accv = 0.f;
        for (int w = 0; w < NW; ++w) {
            const float f = (s_m[w] == -INFINITY) ? 0.f : expf(s_m[w] - m);
            l += s_l[w] * f;
            if (tid < HD) accv += s_o[w][tid] * f;
        }
        float* rec = part + ((long long)(b * H + h) * S + sp) * (HD + 4);
        if (tid < HD) rec[tid] = accv;
        else if (tid == HD) rec[HD] = m;
        else rec[HD + 1] = l;
    }
}

int launch_lm_attn(const float* q, long long ldq, const float* kc, const float* vc, long long kv_bstride, long long ldkv,
                   float* part, int B, int H, int hd, int S, const int* state, float scale, int pos, hipStream_t s) {
    QA_REQUIRE(S >= 1 && S <= 4, "lm_attn: bad split count %d", S);
    const dim3 grid(H, B, S);
    switch (hd) {
        case 64:
            if (pos >= 0) hipLaunchKernelGGL((lm_attn_kernel<64, 8, false>), grid, dim3(512), 0, s, q, kc, vc, kv_bstride, pos, (int)ldq, (int)ldkv, S, H, scale, part, state);
            else hipLaunchKernelGGL((lm_attn_kernel<64, 8, true>), grid, dim3(512), 0, s, q, kc, vc, kv_bstride, pos, (int)ldq, (int)ldkv, S, H, scale, part, state);
            break;
        case 128:
            if (pos >= 0) hipLaunchKernelGGL((lm_attn_kernel<128, 8, false>), grid, dim3(512), 0, s, q, kc, vc, kv_bstride, pos, (int)ldq, (int)ldkv, S, H, scale, part, state);
            else hipLaunchKernelGGL((lm_attn_kernel<128, 8, true>), grid, dim3(512), 0, s, q, kc, vc, kv_bstride, pos, (int)ldq, (int)ldkv, S, H, scale, part, state);
            break;
        case 32:
            if (pos >= 0) hipLaunchKernelGGL((lm_attn_kernel<32, 8, false>), grid, dim3(512), 0, s, q, kc, vc, kv_bstride, pos, (int)ldq, (int)ldkv, S, H, scale, part, state);
            else hipLaunchKernelGGL((lm_attn_kernel<32, 8, true>), grid, dim3(512), 0, s, q, kc, vc, kv_bstride, pos, (int)ldq, (int)ldkv, S, H, scale, part, state);
            break;
        default: set_error("lm_attn: head_dim=%d unsupported", hd); return QA_ERR_UNSUPPORTED;
    }
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// ------------------------------------------------------------------------------------------------
// Greedy pick (llm.py:286 with the range mask of llm_sft.py:150-153 / :180-182 already applied by restricting output_head to the
// slice): arg-max over the head kernel's per-tile maxima, first maximum wins.  tok[b] = lo + argmax; ids[b, col] = argmax for col < keep;
// pos++, col++, step++.  r05: ONE WAVE PER SEQUENCE, four sequences per workgroup (it was one workgroup walking the batch in rounds of 16
// sequences: 5.3 us at 16 sequences, 15.6 us at 64 - four dependent round trips for 250 KB).  The loop state is advanced exactly once, by
// the workgroup that draws the last arrival ticket: every workgroup reads `col` BEFORE it takes its ticket, so nobody can observe the
// advanced state, and the next kernel of the stream sees it through the kernel boundary.
constexpr int PICK_SEQS = 4;
__global__ __launch_bounds__(64 * PICK_SEQS) void lm_pick_kernel(const float* __restrict__ pmax, const int* __restrict__ pidx, int n_tiles,
                                                                 int B, int lo, long long* __restrict__ tok, long long* __restrict__ ids,
                                                                 long long ids_ld, int keep, int* __restrict__ state, int col_arg) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = col_arg >= 0 ? col_arg : state[ST_COL];
    const int b = blockIdx.x * PICK_SEQS + wave;
    if (b < B) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int t = lane; t < n_tiles; t += 64) {
            const float v = pmax[(long long)b * n_tiles + t];
            const int i = pidx[(long long)b * n_tiles + t];
            if (v > best || (v == best && i < bi)) {
                best = v;
                bi = i;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(best, o, 64);
            const int i2 = __shfl_xor(bi, o, 64);
            if (v2 > best || (v2 == best && i2 < bi)) {
                best = v2;
                bi = i2;
            }
        }
        if (lane == 0) {
            if (bi == 0x7fffffff) bi = 0;  // all-NaN row: stay inside the table
            tok[b] = lo + bi;
            if (col < keep) ids[(long long)b * ids_ld + col] = bi;
        }
    }
    __syncthreads();
    if (tid == 0) {
        // acquire-release at agent scope: the ticket orders every workgroup's read of `col` before the last arrival's writes of
        // the state words without leaning on the kernel boundary.  PRECONDITION: state[ST_TICKET] == 0 on entry - lm_phase_init_kernel sets
        // it and the last arrival below restores it; a step aborted in mid-kernel would leave it non-zero (the state would never advance).
        const int ticket = __hip_atomic_fetch_add(&state[ST_TICKET], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == (int)gridDim.x - 1) {  // the last workgroup to arrive: everybody has read col
            state[ST_TICKET] = 0;
            state[ST_POS] += 1;
            state[ST_COL] = col + 1;
            state[ST_STEP] += 1;
        }
    }
}

int launch_lm_pick(const float* pmax, const int* pidx, int n_tiles, int B, int lo, long long* tok, long long* ids, long long ids_ld,
                   int keep, int* state, int col, hipStream_t s) {
    hipLaunchKernelGGL(lm_pick_kernel, dim3((unsigned)ceil_div(B, PICK_SEQS)), dim3(64 * PICK_SEQS), 0, s, pmax, pidx, n_tiles, B, lo, tok, ids,
                       ids_ld, keep, state, col);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// phase start: tok[:] = first_id, pos, col = 0 (the RNG step counter keeps running across the two phases of a call)
__global__ void lm_phase_init_kernel(long long* tok, long long first_id, int B, int* state, int pos, int reset_step, unsigned seed_lo,
                                     unsigned seed_hi, int seq0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) tok[i] = first_id;
    if (i == 0) {
        state[ST_POS] = pos;
        state[ST_COL] = 0;
        state[ST_SEQ0] = seq0;
        state[ST_TICKET] = 0;
        if (reset_step) {
            state[ST_STEP] = 0;
            state[ST_SEED_LO] = (int)seed_lo;
            state[ST_SEED_HI] = (int)seed_hi;
        }
    }
}
int launch_lm_phase_init(long long* tok, long long first_id, int B, int* state, int pos, int reset_step, unsigned long long seed,
                         int seq0, hipStream_t s) {
    hipLaunchKernelGGL(lm_phase_init_kernel, dim3((unsigned)ceil_div(B, 64)), dim3(64), 0, s, tok, first_id, B, state, pos, reset_step,
                       (unsigned)(seed & 0xffffffffull), (unsigned)(seed >> 32), seq0);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

__global__ void lm_advance_kernel(int* state) {
    if (threadIdx.x == 0) {
        state[ST_POS] += 1;
        state[ST_COL] += 1;
        state[ST_STEP] += 1;
    }
}
int launch_lm_advance(int* state, hipStream_t s) {
    hipLaunchKernelGGL(lm_advance_kernel, dim3(1), dim3(64), 0, s, state);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

// ------------------------------------------------------------------------------------------------
// CustomLlamaModel.sample_logits (llm.py:253-288) on the active vocabulary slice, one workgroup per sequence:
//   1. sort the slice descending (value, then lower index first) - bitonic sort of 64-bit keys in LDS;
//   2. top-k: keep everything >= the k-th largest value (the reference removes `logits < kth`, so ties with it survive);
//   3. top-p on the UN-tempered logits: softmax over the survivors, inclusive cumsum, shift right by one: sorted position
//      j >= 1 is removed iff cumsum[j-1] > top_p (the token that crosses top_p is kept);
//   4. / temperature, softmax, one categorical draw (inverse CDF with a Philox4x32-10 uniform keyed by (seed, sequence, step));
//      do_sample = 0 returns the first sorted element instead (arg-max; the filters cannot remove it).
// The reference draws from torch's global RNG; the streams necessarily differ, the distribution is the same (tested).
__device__ __forceinline__ unsigned f2ord(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ float philox_uniform(unsigned seed_lo, unsigned seed_hi, unsigned seq, unsigned step) {
    unsigned c[4] = {step, 0u, seq, 0u};
    unsigned k0 = seed_lo, k1 = seed_hi;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return (float)(c[0] >> 8) * (1.0f / 16777216.0f);  // [0, 1) with 24 random bits
}

// block-wide inclusive scan of one float per thread (1024 threads), returns this thread's inclusive prefix; *total = sum
__device__ __forceinline__ float block_scan_1024(float v, float* s_wave, float* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    __syncthreads();  // s_wave may still be read from a previous call
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    float off = 0.f, tot = 0.f;
    for (int w = 0; w < 16; ++w) {
        const float t = s_wave[w];
        if (w < wave) off += t;
        tot += t;
    }
    *total = tot;
    return x + off;
}

__global__ __launch_bounds__(1024) void lm_sample_kernel(const float* __restrict__ logits, long long ldl, int width, int n_pow2, int lo,
                                                         int top_k, float top_p, float temperature, int do_sample,
                                                         long long* __restrict__ tok, long long* __restrict__ ids, long long ids_ld,
                                                         int keep, const int* __restrict__ state) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];  // n_pow2 entries
    __shared__ float s_wave[16];
    __shared__ int s_cnt, s_pick;
    const int tid = threadIdx.x, b = blockIdx.x;
    const float* row = logits + (long long)b * ldl;
    for (int i = tid; i < n_pow2; i += 1024)
        keys[i] = i < width ? (((unsigned long long)f2ord(row[i])) << 32) | (unsigned)(~(unsigned)i) : 0ull;
    __syncthreads();
    for (int k = 2; k <= n_pow2; k <<= 1) {  // bitonic sort, descending
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n_pow2; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long x = keys[i], y = keys[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? (x < y) : (x > y)) {
                        keys[i] = y;
                        keys[ixj] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
    const float top = ord2f((unsigned)(keys[0] >> 32));
    // ---- top-k: survivors are a prefix [0, n1)
    int n1 = width;
    if (top_k > 0 && top_k < width) {
        const unsigned kth = (unsigned)(keys[top_k - 1] >> 32);
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        int c = 0;
        for (int i = tid; i < width; i += 1024) c += ((unsigned)(keys[i] >> 32) >= kth) ? 1 : 0;
        atomicAdd(&s_cnt, c);
        __syncthreads();
        n1 = s_cnt;
        __syncthreads();
    }
    // -inf entries (nothing below the range mask survives a finite maximum anyway) never carry probability
    // ---- top-p: thread t owns the contiguous sorted positions [t*per, (t+1)*per)
    const int per = n_pow2 / 1024 > 0 ? n_pow2 / 1024 : 1;
    int n2 = n1;
    if (top_p < 1.0f) {
        float loc = 0.f;
        for (int u = 0; u < per; ++u) {
            const int i = tid * per + u;
            if (i < n1) loc += expf(ord2f((unsigned)(keys[i] >> 32)) - top);
        }
        float Z;
        const float incl = block_scan_1024(loc, s_wave, &Z);
        // position j >= 1 is removed iff cumsum[j-1] / Z > top_p  <=>  kept count = 1 + #{j >= 1 : cumsum[j-1] <= top_p * Z}
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        float run = incl - loc;  // exclusive prefix at this thread's first position = cumsum[first - 1]
        int c = 0;
        for (int u = 0; u < per; ++u) {
            const int i = tid * per + u;
            if (i < n1) {
                if (i == 0 || run / Z <= top_p) ++c;
                run += expf(ord2f((unsigned)(keys[i] >> 32)) - top);
            }
        }
        atomicAdd(&s_cnt, c);
        __syncthreads();
        n2 = s_cnt;  // monotone cumsum => the kept positions are exactly the prefix [0, n2)
        __syncthreads();
    }
    // ---- temperature + categorical draw over [0, n2)
    int pick = 0;
    if (do_sample) {
        const float invT = 1.0f / temperature;
        float loc = 0.f;
        for (int u = 0; u < per; ++u) {
            const int i = tid * per + u;
            if (i < n2) loc += expf((ord2f((unsigned)(keys[i] >> 32)) - top) * invT);
        }
        float Z;
        const float incl = block_scan_1024(loc, s_wave, &Z);
        const float uu = philox_uniform((unsigned)state[ST_SEED_LO], (unsigned)state[ST_SEED_HI], (unsigned)(b + state[ST_SEQ0]), (unsigned)state[ST_STEP]) * Z;
        if (tid == 0) s_pick = n2 - 1;  // rounding guard: u * Z may land on the total
        __syncthreads();
        float run = incl - loc;
        for (int u = 0; u < per; ++u) {
            const int i = tid * per + u;
            if (i < n2) {
                const float nxt = run + expf((ord2f((unsigned)(keys[i] >> 32)) - top) * invT);
                if (uu < nxt) atomicMin(&s_pick, i);  // first position whose cumulative mass exceeds u (robust to scan rounding)
                run = nxt;
            }
        }
        __syncthreads();
        pick = s_pick;
    }
    if (tid == 0) {
        const int idx = (int)(~(unsigned)(keys[pick] & 0xffffffffull));
        tok[b] = lo + idx;
        const int col = state[ST_COL];
        if (col < keep) ids[(long long)b * ids_ld + col] = idx;
    }
}

// the sampler sorts up to 16384 64-bit keys in LDS (128 KiB): raise the kernel's dynamic-LDS limit once per device, outside
// any stream capture
int lm_sample_prepare() { return raise_dynamic_lds(reinterpret_cast<const void*>(lm_sample_kernel), 16384 * 8); }

int launch_lm_sample(const float* logits, long long ldl, int width, int B, int lo, int top_k, float top_p, float temperature,
                     int do_sample, long long* tok, long long* ids, long long ids_ld, int keep, const int* state, hipStream_t s) {
    int n_pow2 = 1024;
    while (n_pow2 < width) n_pow2 <<= 1;
    QA_REQUIRE(n_pow2 <= 16384, "lm_sample: vocabulary slice of %d entries exceeds the 16384 the sampler sorts in LDS", width);
    const size_t lds = (size_t)n_pow2 * sizeof(unsigned long long);
    hipLaunchKernelGGL(lm_sample_kernel, dim3(B), dim3(1024), lds, s, logits, ldl, width, n_pow2, lo, top_k, top_p, temperature,
                       do_sample, tok, ids, ids_ld, keep, state);
    QA_LAUNCH_CHECK();
    return QA_OK;
}

}  // namespace qa
