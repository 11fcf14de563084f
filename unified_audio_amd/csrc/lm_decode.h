// lm_decode.h - host-side interface of the fused UniSE decode step kernels (lm_decode.hip), used by lm.cpp and api.cpp.
#pragma once
#include "common.h"

namespace qa {

enum { GM_QKV = 0, GM_GATEUP = 1, GM_RESID = 2, GM_HEAD = 3 };
constexpr int LM_MAX_ROWS = 64;  // sequences one fused decode step serves: two row groups of 32 in every launch (lm_decode.hip, row_group)
// device-side loop state (int words): position of the token being processed (= keys already cached), ids column, RNG step, seed
// ST_SEQ0: index of the chain's first sequence inside the call (the sampler keys its Philox stream by the GLOBAL sequence index)
// ST_TICKET: arrival counter of lm_pick_kernel's workgroups (the last arrival advances the state and clears it)
enum { ST_POS = 0, ST_COL = 1, ST_STEP = 2, ST_SEQ0 = 3, ST_SEED_LO = 4, ST_SEED_HI = 5, ST_TICKET = 6, ST_WORDS = 8 };

// Cross-launch weight prefetch (r06, QA_LM_PF; measured in profiles/r06_lm_prefetch_ab.txt).  A GEMV / fused-MLP launch can carry a SECOND
// Z-PLANE of workgroups (blockIdx.z = 1) that do no arithmetic: they touch one word per 64 bytes of the weight tiles the NEXT launch of the
// step will stream - weights never depend on activations - so that the lines sit in the L2 of the XCD that will read them when that launch
// starts.  Placement: workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md, dispatch) and every grid of the step has a multiple of 8 workgroups
// per row of x, so prefetch workgroup `id` of `count` takes the consumer's tiles j = id, id + count, ... - the same residue mod 8 as the
// consumer workgroup j.  A hint only: it cannot change a result.  The plane is recognised by blockIdx.z (a scalar register at wave start):
// testing a kernel argument instead put one more dependent scalar load in front of every working wave's first weight load (4 ms per
// 16-segment generate with every plane switched off).
struct PfArgs {
    const char* p[2];          // up to two regions of contiguous consumer tiles (gate/up rows and the W_down slices of the fused MLP)
    long long tile_bytes[2];
    int n_tiles[2];
};

struct GemvArgs {
    // A operand: rows of x (or, with tok != nullptr, rows table[tok[m]] - the codec_embedding gather of the step's token)
    const float* x;
    long long ldx;
    const long long* tok;
    const float* table;
    // ... or the merge of the attention partials (GM_RESID with att_part != nullptr)
    const float* att_part;
    int S, H, hd;
    // B operand: [n_tiles * NT][K] rows in decode layout (tile-major; paired modes: NT/2 "first" rows then NT/2 "second" rows)
    const float* w;
    int M, N, K;
    int rpg;  // rows per row group of this launch (0: 16 per 16-row tile of the kernel instance); set by the launchers
    float rms_eps;
    const int* state;
    int pos;  // >= 0: position of the step (host-driven loop); < 0: read it from state (captured step)
    // GM_QKV
    const float* rope;  // [pos][hd/2][2] cos, sin
    float* q;           // [M, d]
    float* kc;
    float* vc;  // caches [M][cap][d]
    long long kv_bstride;
    int d;
    // GM_RESID / GM_GATEUP output, GM_RESID residual (rows of `res`, or table[res_tok[m]])
    const float* res;
    long long ldr;
    const long long* res_tok;
    const float* res_table;
    float* y;
    long long ldy;
    // GM_HEAD
    float* pmax;
    int* pidx;
    float* logits;  // nullable: [M, N] full slice logits for the sampling path
    long long ldl;
};

int lm_pick_nt(int N);
bool lm_gemv_supported(int hidden, int intermediate);  // kernel instances exist for these K
bool gemv_r8_ok(int K);  // the 8-row o_proj kernel (lm_gemv4_kernel R8) exists for this K: every o_proj tile width then shares one K order
int launch_lm_gemv(const GemvArgs& a, int mode, int nt, hipStream_t s, const PfArgs* pf = nullptr);
bool lm_mlp_fused_supported(int d, int I, int nt_gu);
int launch_lm_mlp(const GemvArgs& a, int I, int ac, const float* wd, float* partial, const float* res, long long ldr, float* y, long long ldy,
                  hipStream_t s, const PfArgs* pf = nullptr);
int launch_lm_attn(const float* q, long long ldq, const float* kc, const float* vc, long long kv_bstride, long long ldkv,
                   float* part, int B, int H, int hd, int S, const int* state, float scale, int pos, hipStream_t s);
int launch_lm_pick(const float* pmax, const int* pidx, int n_tiles, int B, int lo, long long* tok, long long* ids, long long ids_ld,
                   int keep, int* state, int col, hipStream_t s);
int launch_lm_phase_init(long long* tok, long long first_id, int B, int* state, int pos, int reset_step, unsigned long long seed,
                         int seq0, hipStream_t s);
int launch_lm_advance(int* state, hipStream_t s);
int lm_sample_prepare();
int launch_lm_sample(const float* logits, long long ldl, int width, int B, int lo, int top_k, float top_p, float temperature,
                     int do_sample, long long* tok, long long* ids, long long ids_ld, int keep, const int* state, hipStream_t s);

}  // namespace qa
