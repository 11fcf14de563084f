// lm_xcd.h - host-side interface of the per-XCD persistent decode kernel (lm_xcd.hip), used by lm.cpp.
#pragma once
#include "common.h"

namespace qa {

constexpr int LM_XCD_MAX_LAYERS = 16;

// weights of one layer in the tile-major layouts lm.cpp builds for this kernel (device pointers; RMSNorm gains folded)
struct LmXcdLayer {
    const float* qkv;   // [slot 32][part q,k,v][wave 8][j 4][lane 64][4]
    const float* o;     // [slot 32][wave 8][j 4][lane 64][4]
    const float* gu;    // [slot 32][group gate,up][K slice 4][j 32][lane 64][4]
    const float* down;  // [slot 32][wave 8][j 16][lane 64][4]
};

struct LmXcdArgs {
    LmXcdLayer layer[LM_XCD_MAX_LAYERS];
    int n_layers;
    const float* head;  // active vocabulary slice of the phase: [slot 32][chunk][group 2][K slice 4][j 32][lane 64][4]
    const float* emb;   // codec_embedding [V][512]
    const float* rope;  // [pos][32][2] cos, sin
    float *kc, *vc;     // KV caches [layer][B][cap][512]
    long long kv_bstride, kv_lstride;
    float *xa, *xb, *qbuf;  // [B][512] team hand-off buffers: layer input, post-attention stream, rotated queries
    float* act;             // [B][2048]
    float* att_part;        // [B][8][4][68]
    float* pmax;            // [B][32] per-slot maxima of the head
    int* pidx;
    long long* tok;         // [B]: the phase's last token on return
    long long* ids;         // [B][ids_ld]: arg-max indices of the first `keep` steps
    long long ids_ld;
    int keep;
    int B, pos0, steps, lo, width;
    long long tok_init;  // token every sequence starts the phase with
    float rms_eps;
    unsigned* sy;        // lm_xcd_sync_bytes() of device memory, zeroed by the launcher
    unsigned* err_host;  // mapped pinned word: set by a kernel whose bounded barrier spin ran out
    unsigned spin_limit;
    int fault;           // tests: the barrier waits for one member more than exists
    int prefetch;        // 1: the next stage's weights are loaded between arrival at a team barrier and the wait for the team
};

bool lm_xcd_supported(int d, int heads, int inter, int global_size, int semantic_size);
size_t lm_xcd_sync_bytes();
int launch_lm_xcd_decode(const LmXcdArgs& a, hipStream_t s);

}  // namespace qa
