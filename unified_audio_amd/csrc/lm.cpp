// lm.cpp - UniSE decoder-only AR-LM: prompt assembly, prefill and the greedy global + semantic decode loop.
//
// Mirrors LLM_SFT.generate (QuarkAudio-UniSE/model/llm/llm_sft.py:93-195) over the Llama body the reference lifts out of
// HF transformers (model/llm/llm.py:63-79,150-227): RMSNorm -> QKV (no bias) -> rotate-half RoPE -> causal attention over
// the KV cache -> O -> +res -> RMSNorm -> SwiGLU -> +res, final RMSNorm, output_head, vocabulary-range mask, greedy arg-max
// (llm.py:253-288 with do_sample=False, the reference's test path: model/model.py:173).
//
// The whole loop is device-resident: token ids never visit the host between steps (embedding gather and arg-max read / write a
// device int64 vector), the KV cache is owned by the handle's workspace, prefill runs on the implicit-GEMM / flash kernels
// and each decode step on the weight-streaming skinny GEMM + single-query attention kernels.
#include <memory>

#include <cstdlib>

#include "host_util.h"
#include "lm_decode.h"

namespace qa {
int launch_assemble_prompt(float* x, const float* task_vec, const float* enroll_sos, const float* enroll_emb,
                           const float* mix_sos, const float* mix_emb, int B, int Ne, int Nm, int d, hipStream_t s);
int launch_skinny_gemm(const float* x, long long ldx, const float* w, const float* bias, const float* gate,
                       long long ldg, const float* res, long long ldr, float* y, long long ldy, int M, int N, int K,
                       int act, hipStream_t s, float rms_eps, int dual);
int launch_rope_kv(float* qkv, const float* cs, float* kc, float* vc, int B, int n, int H, int hd, int pos0, int max_len,
                   hipStream_t s);

}  // namespace qa

using namespace qa;

namespace {
struct LMLayer {
    // RMSNorm weights are folded into the consuming projections (W' = W diag(w)): qkv <- input_layernorm, gate/up <- post_attention_layernorm
    ConvW qkv, o, gate, up, down;
    const float* gate_up = nullptr;  // decode layout: per 16-column group 16 gate rows then 16 up rows (both norm-folded)
    // fused decode step (lm_decode.hip): tile-major rows; qkv_dec pairs rotary partners (i, i + hd/2) inside a tile,
    // gu_dec holds per tile NT/2 gate rows then NT/2 up rows (both norm-folded)
    const float* qkv_dec = nullptr;
    const float* gu_dec = nullptr;
    const float* down_dec = nullptr;  // fused MLP: W_down slice-major [I / 16][d][16] (slice j = columns 16 j .. 16 j + 15 of every row)
};
constexpr int LM_MAX_POS = 4096;  // max_position_embeddings (conf/config.yaml:146)
constexpr int LM_MAX_CHAINS = 16;
}  // namespace

struct StepGraph {  // one captured decode step of a phase, replayable because every loop-carried scalar lives in device state
    hipGraphExec_t exec = nullptr;
    hipGraph_t graph = nullptr;
    uint64_t key = 0;
    void reset() {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        exec = nullptr;
        graph = nullptr;
        key = 0;
    }
};

struct qa_lm {
    qa_lm_spec spec{};
    int device = 0;
    bool fused_ok = false;   // the shapes fit the fused decode step (required since r05: build_lm refuses a spec that does not tile)
    bool mlp_fused = false;  // QA_LM_MLP_FUSED at create time: gate/up + SwiGLU + down as one launch + a reduce launch
    int mlp_ac = 16;         // activation columns per workgroup of that launch (8 measured equal at B = 16, -4 % at B = 64: profiles/r03_lm_ab.txt)
    int att_split = 256;     // keys per workgroup of the decode attention, at most 4 splits (flat between 256 and 384, worse below: same log; re-measured in r05 with the cheaper o_proj: 128 / 160 / 192 keys 112.2 - 112.5 ms against 110.4 at 16 segments)
    int nt_qkv = 0, nt_o = 0, nt_gu = 0, nt_down = 0;
    hipStream_t cap_stream = nullptr;
    std::vector<StepGraph> graphs;  // [2 * chain + phase]: phase 0 global, 1 semantic
    std::vector<hipStream_t> chain_streams;  // internal streams of the chains of a B > 32 call
    std::vector<hipEvent_t> chain_join;
    hipEvent_t ev_fork = nullptr;
    unsigned long long calls = 0;
    WeightStore store;
    const float *task_emb = nullptr, *enroll_sos = nullptr, *mix_sos = nullptr, *codec_emb = nullptr, *ones = nullptr,
                *rope = nullptr;  // `ones`: unit RMSNorm weight (the learned ones are folded into the projections)
    ConvW adapter, head;
    std::vector<LMLayer> layers;
    char* ws = nullptr;
    size_t ws_cap = 0;
    Ctx ctx;
};

namespace {

int vocab_of(const qa_lm_spec& s) { return 3 + s.global_size + s.semantic_size; }

// y[rows, N] = epi(x[rows, K] W^T): skinny kernel for decode-sized M, implicit GEMM otherwise
bool skinny_ok(int64_t rows, const ConvW& w) { return rows <= 32 && w.C_in % 256 == 0; }

int lm_linear(Ctx& c, const float* x, int64_t rows, const ConvW& w, float* y, const float* res = nullptr,
              const float* gate = nullptr, int n_rows_w = -1, const float* w_ptr = nullptr, float rms_eps = 0.f) {
    if (c.dry) return QA_OK;
    const int N = n_rows_w >= 0 ? n_rows_w : w.N;
    const float* wp = w_ptr ? w_ptr : w.w;
    if (skinny_ok(rows, w))
        return launch_skinny_gemm(x, w.C_in, wp, w.b, gate, N, res, N, y, N, (int)rows, N, w.C_in, ACT_NONE, c.stream, rms_eps, 0);
    QA_REQUIRE(rms_eps == 0.f, "lm_linear: fused RMSNorm is only available on the skinny path");
    qa_conv_args a{};
    a.x = x; a.w = wp; a.bias = w.b; a.residual = res; a.gate = gate; a.y = y;
    a.B = 1; a.T_in = rows; a.C_in = w.C_in; a.T_out = rows; a.N = N;
    a.ldx = w.C_in; a.ldy = N; a.ldr = N; a.ldg = N;
    a.ksize = 1; a.stride = 1;
    ConvParams p;
    QA_TRY(conv_params_from_args(a, &p));
    return launch_conv_gemm(p, c.stream);
}

// tile width of the head GEMV over a vocabulary slice of `width` entries (0: the slice does not tile)
int head_nt(int width) {
    int nt = lm_pick_nt(width);
    while (nt >= 4 && width % nt) nt >>= 1;
    return nt >= 4 ? nt : 0;
}

int build_lm(qa_lm* lm, const HostTable& tab) {
    const qa_lm_spec& sp = lm->spec;
    const int d = sp.hidden, V = vocab_of(sp), I = sp.intermediate;
    QA_REQUIRE(sp.n_heads > 0 && d % sp.n_heads == 0, "lm spec: hidden %d not divisible by %d heads", d, sp.n_heads);
    const int hd = d / sp.n_heads;
    QA_REQUIRE(hd == 32 || hd == 64 || hd == 128, "lm spec: head_dim %d unsupported", hd);
    QA_REQUIRE(d % 32 == 0 && I % 32 == 0 && sp.feats_dim % 32 == 0, "lm spec: widths must be multiples of 32");
    // fused decode step: K of every GEMV a multiple of 256, every N a multiple of its tile width, rotary pairs inside a tile
    lm->nt_qkv = lm_pick_nt(3 * d);
    lm->nt_o = lm_pick_nt(d);
    lm->nt_gu = lm_pick_nt(2 * I);
    lm->nt_down = lm_pick_nt(d);
    lm->mlp_fused = knob(K_LM_MLP_FUSED) != 0 && lm_mlp_fused_supported(d, I, lm->nt_gu);
    lm->fused_ok = lm_gemv_supported(d, I) && d % lm->nt_qkv == 0 && hd % lm->nt_qkv == 0 && d % lm->nt_o == 0 &&
                   (2 * I) % lm->nt_gu == 0 && hd % 8 == 0 && head_nt(sp.global_size) && head_nt(sp.semantic_size);
    // r05: the per-op decode step of round 1 (skinny GEMM + attention_decode kernels behind QA_LM_UNFUSED, the path a spec took when the
    // fused step did not tile it) is gone - a spec the fused step cannot tile is refused here, with the reason
    QA_REQUIRE(lm->fused_ok, "lm spec: the decode step needs hidden %% 256 == 0 (got %d), intermediate %% 256 == 0 (got %d), head_dim %% 8 == 0 and "
               "global / semantic vocabulary sizes that are multiples of 4 (got %d / %d)", d, I, sp.global_size, sp.semantic_size);
    WeightStore& st = lm->store;
    bool ok = true;
    std::vector<std::pair<const float**, size_t>> pend;
    auto vec = [&](const float** dst, const std::string& name, int64_t n) {
        const float* p = tab.get(name, n);
        if (!p) {
            ok = false;
            return;
        }
        pend.push_back({dst, st.add(p, n)});
    };
    vec(&lm->task_emb, "task_embedding.weight", (int64_t)sp.num_tasks * d);
    vec(&lm->enroll_sos, "enroll_sos_embedding.weight", d);
    vec(&lm->mix_sos, "mix_sos_embedding.weight", d);
    vec(&lm->codec_emb, "codec_embedding.weight", (int64_t)V * d);
    {
        std::vector<float> ones(d, 1.0f);
        pend.push_back({&lm->ones, st.add(ones)});
    }
    // fold W' = W diag(w_norm): rows of W scaled column-wise by the RMSNorm weight that precedes the projection
    auto folded = [&](const std::string& wname, int64_t rows, const float* nw, std::vector<float>* out) -> bool {
        const float* w = tab.get(wname, rows * d);
        if (!w || !nw) return false;
        out->resize((size_t)rows * d);
        for (int64_t r = 0; r < rows; ++r)
            for (int k = 0; k < d; ++k) (*out)[(size_t)r * d + k] = w[r * d + k] * nw[k];
        return true;
    };
    lm->adapter.N = d; lm->adapter.C_in = sp.feats_dim;
    vec(&lm->adapter.w, "adapter.weight", (int64_t)d * sp.feats_dim);
    vec(&lm->adapter.b, "adapter.bias", d);
    lm->head.N = V; lm->head.C_in = d;
    {
        std::vector<float> hw;
        if (folded("output_head.weight", V, tab.get("norm.weight", d), &hw)) {
            pend.push_back({&lm->head.w, st.add(hw)});
        } else {
            ok = false;
        }
    }
    lm->layers.resize(sp.n_layers);
    for (int i = 0; i < sp.n_layers; ++i) {
        LMLayer& L = lm->layers[i];
        const std::string p = "layers." + std::to_string(i);
        const float* ln1 = tab.get(p + ".input_layernorm.weight", d);
        const float* ln2 = tab.get(p + ".post_attention_layernorm.weight", d);
        if (!ln1 || !ln2) ok = false;
        std::vector<float> wq((size_t)3 * d * d, 0.f);
        const char* nm[3] = {".self_attn.q_proj.weight", ".self_attn.k_proj.weight", ".self_attn.v_proj.weight"};
        for (int j = 0; j < 3; ++j) {
            std::vector<float> f;
            if (!folded(p + nm[j], d, ln1, &f)) ok = false;
            else std::memcpy(&wq[(size_t)j * d * d], f.data(), sizeof(float) * d * d);
        }
        L.qkv.N = 3 * d; L.qkv.C_in = d;
        pend.push_back({&L.qkv.w, st.add(wq)});
        if (lm->fused_ok) {  // tile t of a section: rows [t*NT/2, (t+1)*NT/2) of each head-half, first halves then partner halves
            const int nt = lm->nt_qkv, hp = nt / 2;
            std::vector<float> wd((size_t)3 * d * d);
            size_t r = 0;
            for (int sec = 0; sec < 3; ++sec)
                for (int h = 0; h < sp.n_heads; ++h)
                    for (int t = 0; t < hd / nt; ++t)
                        for (int half = 0; half < 2; ++half)
                            for (int j = 0; j < hp; ++j, ++r) {
                                const size_t src = (size_t)sec * d + (size_t)h * hd + (size_t)half * (hd / 2) + (size_t)t * hp + j;
                                std::memcpy(&wd[r * d], &wq[src * d], sizeof(float) * d);
                            }
            pend.push_back({&L.qkv_dec, st.add(wd)});
        }
        L.o.N = d; L.o.C_in = d;
        vec(&L.o.w, p + ".self_attn.o_proj.weight", (int64_t)d * d);
        L.gate.N = I; L.gate.C_in = d;
        L.up.N = I; L.up.C_in = d;
        {
            std::vector<float> g, u;
            if (folded(p + ".mlp.gate_proj.weight", I, ln2, &g) && folded(p + ".mlp.up_proj.weight", I, ln2, &u)) {
                pend.push_back({&L.gate.w, st.add(g)});
                pend.push_back({&L.up.w, st.add(u)});
                if (I % 16 == 0) {  // decode layout for the dual-accumulator skinny kernel
                    std::vector<float> gu((size_t)2 * I * d);
                    for (int blk = 0; blk < I / 16; ++blk) {
                        std::memcpy(&gu[(size_t)(blk * 32) * d], &g[(size_t)(blk * 16) * d], sizeof(float) * 16 * d);
                        std::memcpy(&gu[(size_t)(blk * 32 + 16) * d], &u[(size_t)(blk * 16) * d], sizeof(float) * 16 * d);
                    }
                    pend.push_back({&L.gate_up, st.add(gu)});
                }
                if (lm->fused_ok) {
                    const int hp = lm->nt_gu / 2;
                    std::vector<float> gd((size_t)2 * I * d);
                    for (int t = 0; t < I / hp; ++t) {
                        std::memcpy(&gd[(size_t)(t * 2 * hp) * d], &g[(size_t)(t * hp) * d], sizeof(float) * hp * d);
                        std::memcpy(&gd[(size_t)(t * 2 * hp + hp) * d], &u[(size_t)(t * hp) * d], sizeof(float) * hp * d);
                    }
                    pend.push_back({&L.gu_dec, st.add(gd)});
                }
            } else {
                ok = false;
            }
        }
        L.down.N = d; L.down.C_in = I;
        vec(&L.down.w, p + ".mlp.down_proj.weight", (int64_t)I * d);
        if (lm->fused_ok && lm->mlp_fused) {
            const float* wdn = tab.get(p + ".mlp.down_proj.weight", (int64_t)I * d);
            if (wdn) {
                std::vector<float> ws((size_t)I * d);
                const int ac = lm->mlp_ac;
                for (int j = 0; j < I / ac; ++j)
                    for (int n = 0; n < d; ++n) std::memcpy(&ws[((size_t)j * d + n) * ac], &wdn[(size_t)n * I + (size_t)j * ac], sizeof(float) * ac);
                pend.push_back({&L.down_dec, st.add(ws)});
            } else {
                ok = false;
            }
        }
    }
    if (!ok) return QA_ERR_MISSING;
    // LlamaRotaryEmbedding (default rope): inv_freq = theta^(-2i/hd), cos / sin of pos * inv_freq in fp32
    const int half = hd / 2;
    std::vector<float> cs((size_t)LM_MAX_POS * half * 2);
    for (int i = 0; i < half; ++i) {
        const float inv = 1.0f / std::pow(sp.rope_theta, (float)(2 * i) / (float)hd);
        for (int t = 0; t < LM_MAX_POS; ++t) {
            const float fr = (float)t * inv;
            cs[((size_t)t * half + i) * 2] = (float)std::cos((double)fr);
            cs[((size_t)t * half + i) * 2 + 1] = (float)std::sin((double)fr);
        }
    }
    pend.push_back({&lm->rope, st.add(cs)});
    QA_TRY(st.upload());
    for (auto& pv : pend) *pv.first = st.ptr(pv.second);
    return QA_OK;
}

struct LMBuffers {
    float *x, *hn, *qkv, *att, *g, *u, *logits;
    float *kc, *vc;
    long long* tok;
    // fused decode step
    float *q, *att_part, *pmax, *mlp_part;
    int *pidx, *state;
    long long *ids_g, *ids_s;
    int cap, S_att;
};

struct SampleCfg {
    int do_sample;
    int top_k;
    float top_p, temperature;
    unsigned long long seed;
};

// one pass of the Llama body over `n` new positions per sequence, positions pos0..pos0+n-1 (the prefill)
int lm_body(qa_lm* lm, Ctx& c, LMBuffers& b, int B, int n, int pos0, int max_len, bool skip_last_mlp) {
    const qa_lm_spec& sp = lm->spec;
    const int d = sp.hidden, H = sp.n_heads, hd = d / H;
    const int64_t rows = (int64_t)B * n;
    const float scale = 1.0f / std::sqrt((float)hd);
    const size_t cache_stride = (size_t)B * max_len * d;
    for (int i = 0; i < sp.n_layers; ++i) {
        const LMLayer& L = lm->layers[i];
        float* kc = b.kc + i * cache_stride;
        float* vc = b.vc + i * cache_stride;
        if (!c.dry) {
            const bool last = skip_last_mlp && i == sp.n_layers - 1;  // prefill: only the KV cache of the last layer is consumed
            const bool dec = skinny_ok(rows, L.qkv);                   // decode step: norms fused into the weight-streaming GEMMs
            if (dec) {
                QA_TRY(lm_linear(c, b.x, rows, L.qkv, b.qkv, nullptr, nullptr, -1, nullptr, sp.rms_eps));
            } else {
                QA_TRY(launch_rmsnorm(b.x, lm->ones, b.hn, rows, d, sp.rms_eps, c.stream));
                QA_TRY(lm_linear(c, b.hn, rows, L.qkv, b.qkv));
            }
            QA_TRY(launch_rope_kv(b.qkv, lm->rope, kc, vc, B, n, H, hd, pos0, max_len, c.stream));
            if (last) break;
            QA_TRY(launch_attention(b.qkv, 3 * d, kc, vc, d, b.att, d, B, n, pos0 + n, (long long)max_len * d, H, hd, scale, 1, c.stream));
            QA_TRY(lm_linear(c, b.att, rows, L.o, b.x, b.x));
            if (dec && L.gate_up) {
                QA_TRY(launch_skinny_gemm(b.x, d, L.gate_up, nullptr, nullptr, 0, nullptr, 0, b.u, L.gate.N, (int)rows, L.gate.N, d,
                                          ACT_NONE, c.stream, sp.rms_eps, 1));
            } else {
                QA_TRY(launch_rmsnorm(b.x, lm->ones, b.hn, rows, d, sp.rms_eps, c.stream));
                QA_TRY(lm_linear(c, b.hn, rows, L.gate, b.g));
                QA_TRY(lm_linear(c, b.hn, rows, L.up, b.u, nullptr, b.g));
            }
            QA_TRY(lm_linear(c, b.u, rows, L.down, b.x, b.x));
        }
    }
    return QA_OK;
}

// ONE decode step as 5 launches per layer + 2 (lm_decode.hip).  Everything step-dependent (position, ids column, RNG step) is read
// from b.state on the device, so the launch arguments are identical for every step of a phase: the sequence can be captured once
// into a hipGraph and replayed.
int fused_step(qa_lm* lm, const LMBuffers& b, int B, int lo, int width, long long* ids, int ids_ld, int keep, const SampleCfg& sc,
               hipStream_t s, int pos, int col) {  // pos / col >= 0: host-driven loop; -1: read from the device state (captured step)
    const qa_lm_spec& sp = lm->spec;
    const int d = sp.hidden, H = sp.n_heads, hd = d / H, I = sp.intermediate;
    const float scale = 1.0f / std::sqrt((float)hd);
    const long long kv_bstride = (long long)b.cap * d;
    const size_t cache_stride = (size_t)B * b.cap * d;
    // key split of the attention launch: a workgroup's 8 waves hold 2 tiles of 16 keys each per round
    const int S_att = pos >= 0 ? std::max(1, std::min(4, (int)ceil_div(pos + 1, lm->att_split))) : b.S_att;
    // cross-launch prefetch (QA_LM_PF, lm_decode.h PfArgs): which launch carries the prefetch plane for which consumer
    const long long pfk = knob(K_LM_PF);
    const bool fused_mlp = lm->mlp_fused;
    auto pf_region = [](PfArgs& pf, int r, const float* w, long long tile_bytes, int n_tiles) {
        pf.p[r] = reinterpret_cast<const char*>(w);
        pf.tile_bytes[r] = tile_bytes;
        pf.n_tiles[r] = n_tiles;
    };
    auto pf_mlp = [&](PfArgs& pf, const LMLayer& L) {  // the fused MLP launch: workgroup j streams 2 gate/up tiles (32 rows) and W_down slice j
        if (!fused_mlp || !L.down_dec) {  // separate launches (QA_LM_MLP_FUSED=0): the gate/up launch streams tiles of nt_gu rows
            pf_region(pf, 0, L.gu_dec, (long long)lm->nt_gu * d * 4, 2 * I / lm->nt_gu);
            return;
        }
        pf_region(pf, 0, L.gu_dec, (long long)2 * lm->mlp_ac * d * 4, I / lm->mlp_ac);
        pf_region(pf, 1, L.down_dec, (long long)lm->mlp_ac * d * 4, I / lm->mlp_ac);
    };
    for (int i = 0; i < sp.n_layers; ++i) {
        const LMLayer& L = lm->layers[i];
        float* kc = b.kc + i * cache_stride;
        float* vc = b.vc + i * cache_stride;
        GemvArgs a{};
        a.M = B; a.rms_eps = sp.rms_eps; a.state = b.state; a.pos = pos; a.H = H; a.hd = hd; a.d = d;
        // o_proj's tile width (computed here: the qkv launch may prefetch its weights)
        int nt_o = lm->nt_o;
        if (gemv_r8_ok(d)) {
            if (B > 16 && nt_o < 8 && d % 8 == 0) nt_o = 8;
            if (B > 32 && nt_o < 16 && d % 16 == 0) nt_o = 16;
        }
        // 1. RMSNorm + QKV + RoPE + cache append; layer 0 gathers its input rows from codec_embedding (llm_sft.py:140,169)
        GemvArgs q = a;
        q.x = b.x; q.ldx = d;
        if (i == 0) { q.tok = b.tok; q.table = lm->codec_emb; }
        q.w = L.qkv_dec; q.N = 3 * d; q.K = d;
        q.rope = lm->rope; q.q = b.q; q.kc = kc; q.vc = vc; q.kv_bstride = kv_bstride;
        PfArgs qpf{};
        if (pfk & 1) pf_region(qpf, 0, L.o.w, (long long)nt_o * d * 4, d / nt_o);
        QA_TRY(launch_lm_gemv(q, GM_QKV, lm->nt_qkv, s, &qpf));
        // 2. attention over the cache (pos + 1 keys), split over S_att workgroups per (sequence, head)
        QA_TRY(launch_lm_attn(b.q, d, kc, vc, kv_bstride, d, b.att_part, B, H, hd, S_att, b.state, scale, pos, s));
        // 3. merge of the partials + o_proj + residual
        GemvArgs o = a;
        o.att_part = b.att_part; o.S = S_att;
        o.w = L.o.w; o.N = d; o.K = d; o.ldx = d;
        if (i == 0) { o.res_tok = b.tok; o.res_table = lm->codec_emb; } else { o.res = b.x; }
        o.ldr = d; o.y = b.x; o.ldy = d;
        // tile width by the batch: the 8-row groups (lm_gemv4_kernel R8) hold a workgroup's pull of attention partials at 52 KB whatever the
        // batch, but every column tile re-reads and re-merges the partials of its rows - so the launch keeps ~256 workgroups: 4 columns x 2
        // row groups at 16 sequences, 8 x 4 at 32, 16 x 8 at 64 (r05 A/B, generate ms at 16 / 32 / 64 sequences: width 4 110.4 / 147.4 / 206.5,
        // width 8 113.1 / 143.7 / 198.6, width 16 118.5 / 149.2 / 196.4; profiles/r05_lm_oproj_ab.txt)
        PfArgs opf{};
        if (pfk & 2) pf_mlp(opf, L);
        QA_TRY(launch_lm_gemv(o, GM_RESID, nt_o, s, &opf));
        // 4. RMSNorm + gate / up + SwiGLU
        GemvArgs g = a;
        g.x = b.x; g.ldx = d; g.w = L.gu_dec; g.N = 2 * I; g.K = d; g.y = b.u; g.ldy = I;
        if (lm->mlp_fused && L.down_dec) {  // 4 + 5 as the fused MLP launch + the reduce launch (lm_decode.hip)
            PfArgs gpf{};
            if (pfk & 4) {
                if (i + 1 < sp.n_layers) pf_region(gpf, 0, lm->layers[i + 1].qkv_dec, (long long)lm->nt_qkv * d * 4, 3 * d / lm->nt_qkv);
                else pf_region(gpf, 0, lm->head.w + (size_t)lo * d, (long long)head_nt(width) * d * 4, width / head_nt(width));
            }
            QA_TRY(launch_lm_mlp(g, I, lm->mlp_ac, L.down_dec, b.mlp_part, b.x, d, b.x, d, s, &gpf));
            continue;
        }
        PfArgs gupf{};
        if (pfk & 2) pf_region(gupf, 0, L.down.w, (long long)lm->nt_down * I * 4, d / lm->nt_down);
        QA_TRY(launch_lm_gemv(g, GM_GATEUP, lm->nt_gu, s, &gupf));
        // 5. down_proj + residual
        GemvArgs dn = a;
        dn.x = b.u; dn.ldx = I; dn.w = L.down.w; dn.N = d; dn.K = I; dn.res = b.x; dn.ldr = d; dn.y = b.x; dn.ldy = d;
        PfArgs dpf{};
        if (pfk & 4) {
            if (i + 1 < sp.n_layers) pf_region(dpf, 0, lm->layers[i + 1].qkv_dec, (long long)lm->nt_qkv * d * 4, 3 * d / lm->nt_qkv);
            else pf_region(dpf, 0, lm->head.w + (size_t)lo * d, (long long)head_nt(width) * d * 4, width / head_nt(width));
        }
        QA_TRY(launch_lm_gemv(dn, GM_RESID, lm->nt_down, s, &dpf));
    }
    // 6. final RMSNorm (weight folded into output_head) + the rows of output_head inside the active vocabulary slice (the range
    //    mask of llm_sft.py:150-153 / :180-182 sets everything else to -inf) + per-tile arg-max
    const int nt = head_nt(width);
    GemvArgs hg{};
    hg.M = B; hg.rms_eps = sp.rms_eps; hg.state = b.state; hg.pos = pos; hg.H = H; hg.hd = hd; hg.d = d;
    hg.x = b.x; hg.ldx = d; hg.w = lm->head.w + (size_t)lo * d; hg.N = width; hg.K = d;
    hg.pmax = b.pmax; hg.pidx = b.pidx; hg.logits = sc.do_sample ? b.logits : nullptr; hg.ldl = width;
    PfArgs hpf{};
    if (pfk & 8) pf_region(hpf, 0, lm->layers[0].qkv_dec, (long long)lm->nt_qkv * d * 4, 3 * d / lm->nt_qkv);
    QA_TRY(launch_lm_gemv(hg, GM_HEAD, nt, s, &hpf));
    // 7. next token
    if (!sc.do_sample) {
        QA_TRY(launch_lm_pick(b.pmax, b.pidx, width / nt, B, lo, b.tok, ids, ids_ld, keep, b.state, col, s));
    } else {
        QA_TRY(launch_lm_sample(b.logits, width, width, B, lo, sc.top_k, sc.top_p, sc.temperature, 1, b.tok, ids, ids_ld, keep, b.state, s));
        QA_TRY(launch_lm_advance(b.state, s));
    }
    return QA_OK;
}

uint64_t mix_key(uint64_t h, uint64_t v) {
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    return h;
}

// QA_LM_GRAPH=1: replay one captured step per token instead of launching its kernels from the host.  Measured equal within 1 %
// on MI355X (the step is bound by its ~62 dependent kernels, not by the host), and a replayed step must read the position from
// device memory - one more dependent load per kernel - and cannot size the attention grid to the current key count: off by default.
bool use_graphs() { return knob(K_LM_GRAPH) != 0; }

// One chain = one batch of <= 64 sequences (LM_MAX_ROWS) with its own buffers, KV cache and (for B > 64, or QA_LM_CHAINS) its own internal stream.
struct Chain {
    int b0 = 0, B = 0;  // sequences [b0, b0 + B) of the call
    LMBuffers b{};
    float *emix = nullptr, *eenr = nullptr;
    hipStream_t s = nullptr;
};

int chain_alloc(qa_lm* lm, Ctx& c, Chain& ch, int L, int cap, int G, int S, int Nm, int Ne, bool enroll) {
    const qa_lm_spec& sp = lm->spec;
    const int d = sp.hidden, I = sp.intermediate, H = sp.n_heads, B = ch.B;
    const int64_t prow = (int64_t)B * L;
    LMBuffers& b = ch.b;
    b.cap = cap;
    b.S_att = std::max(1, std::min(4, (int)ceil_div(cap, lm->att_split)));  // captured steps: sized for the cache capacity
    b.x = c.arena.alloc<float>(prow * d);
    b.hn = c.arena.alloc<float>(prow * d);
    b.qkv = c.arena.alloc<float>(prow * 3 * d);
    b.att = c.arena.alloc<float>(prow * d);
    b.g = c.arena.alloc<float>(prow * I);
    b.u = c.arena.alloc<float>(prow * I);
    const int wmax = std::max(sp.global_size, sp.semantic_size);
    b.logits = c.arena.alloc<float>((size_t)B * wmax);
    b.kc = c.arena.alloc<float>((size_t)sp.n_layers * B * cap * d);
    b.vc = c.arena.alloc<float>((size_t)sp.n_layers * B * cap * d);
    b.tok = c.arena.alloc<long long>(B);
    b.q = c.arena.alloc<float>((size_t)B * d);
    b.att_part = c.arena.alloc<float>((size_t)B * H * b.S_att * (d / H + 4));
    b.mlp_part = c.arena.alloc<float>(lm->mlp_fused ? (size_t)(I / lm->mlp_ac) * 32 * ceil_div(B, 32) * d : 0);  // [row group][I / ac][32][d]
    b.pmax = c.arena.alloc<float>((size_t)B * (wmax / 4 + 1));
    b.pidx = c.arena.alloc<int>((size_t)B * (wmax / 4 + 1));
    b.state = c.arena.alloc<int>(ST_WORDS);
    b.ids_g = c.arena.alloc<long long>((size_t)B * std::max(G, 1));
    b.ids_s = c.arena.alloc<long long>((size_t)B * std::max(S, 1));
    ch.emix = c.arena.alloc<float>((size_t)B * Nm * d);
    ch.eenr = enroll ? c.arena.alloc<float>((size_t)B * Ne * d) : nullptr;
    return QA_OK;
}

// r05: up to 64 sequences are ONE chain - every GEMV / fused-MLP launch of the step carries two row groups of 32 (gridDim.y, lm_decode.hip
// row_group), so the step streams the 217 MB of weights once and issues 50 launches where two chains of 32 issued 100.
// Batches of more than 64 sequences run as ceil(B / 64) independent CHAINS
// on internal streams: a decode step is bound by the latency of its ~62 dependent launches and leaves the device almost idle
// (DESIGN.md section 11), so chains overlap nearly for free - tokens/s scales with the number of chains until the CUs fill.  Every
// chain replays ONE captured step per token (hipGraph), round-robin over the chains, so the host issues two graph launches per
// step instead of 124 kernel launches (eager launches go host-bound below ~3 us per kernel).  QA_LM_CHAINS forces a chain count.
int generate_graph(qa_lm* lm, Ctx& c, int task, const float* enroll, int Ne, const float* mix, int Nm, int B, int G,
                   int S, long long* gids, long long* sids, const SampleCfg& sc) {
    const qa_lm_spec& sp = lm->spec;
    const int d = sp.hidden;
    const int L = 1 + (enroll ? 1 + Ne : 0) + 1 + Nm;
    const int max_len = L + G + 1 + S;
    QA_REQUIRE(max_len <= LM_MAX_POS, "generate: %d positions exceed max_position_embeddings %d", max_len, LM_MAX_POS);
    const int cap = (int)round_up(max_len, 64);  // cache row stride: shapes that round alike share their captured step graphs
    // A caller that is CAPTURING its stream into a hipGraph gets the plain single-chain launches on that stream only: no
    // internal streams, no capture of our own inside theirs (the multi-chain replay path would fail there where the eager path works).
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(c.stream, &cap_status) != hipSuccess) {
        (void)hipGetLastError();
        cap_status = hipStreamCaptureStatusNone;
    }
    const bool capturing = cap_status == hipStreamCaptureStatusActive;
    int nc = (int)knob(K_LM_CHAINS);
    if (nc <= 0) nc = (int)ceil_div(B, LM_MAX_ROWS);  // r05: one chain serves up to 64 sequences (two row groups per launch)
    if (capturing) nc = 1;
    nc = std::max(1, std::min(std::min(nc, B), LM_MAX_CHAINS));
    if (ceil_div(B, nc) > LM_MAX_ROWS) nc = (int)ceil_div(B, LM_MAX_ROWS);  // a forced count (or a capturing caller) that would put > 64 sequences in a chain
    QA_REQUIRE(!capturing || nc == 1, "generate: a caller that captures its stream can pass at most %d sequences per call (got %d)", LM_MAX_ROWS, B);
    const int cb = (int)ceil_div(B, nc);
    nc = (int)ceil_div(B, cb);
    std::vector<Chain> chains(nc);
    for (int i = 0; i < nc; ++i) {
        chains[i].b0 = i * cb;
        chains[i].B = std::min(cb, B - i * cb);
        QA_TRY(chain_alloc(lm, c, chains[i], L, cap, G, S, Nm, Ne, enroll != nullptr));
    }
    if (c.dry) return QA_OK;
    const bool multi = nc > 1;
    if (multi) {  // fork: the chains' streams start behind everything already queued on the caller's stream
        while ((int)lm->chain_streams.size() < nc) {
            hipStream_t st = nullptr;
            hipEvent_t ev = nullptr;
            QA_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            QA_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            lm->chain_streams.push_back(st);
            lm->chain_join.push_back(ev);
        }
        if (!lm->ev_fork) QA_HIP(hipEventCreateWithFlags(&lm->ev_fork, hipEventDisableTiming));
        QA_HIP(hipEventRecord(lm->ev_fork, c.stream));
        for (int i = 0; i < nc; ++i) {
            chains[i].s = lm->chain_streams[i];
            QA_HIP(hipStreamWaitEvent(chains[i].s, lm->ev_fork, 0));
        }
    } else {
        chains[0].s = c.stream;
    }
    hipStream_t caller = c.stream;

    // ---- prompt (llm_sft.py:110-128) and prefill (llm_sft.py:130-135), chain by chain (these launches fill the device by themselves)
    for (Chain& ch : chains) {
        c.stream = ch.s;
        const float* mix_c = mix + (size_t)ch.b0 * Nm * sp.feats_dim;
        const float* enr_c = enroll ? enroll + (size_t)ch.b0 * Ne * sp.feats_dim : nullptr;
        int st = lm_linear(c, mix_c, (int64_t)ch.B * Nm, lm->adapter, ch.emix);
        if (st == QA_OK && enroll) st = lm_linear(c, enr_c, (int64_t)ch.B * Ne, lm->adapter, ch.eenr);
        if (st == QA_OK)
            st = launch_assemble_prompt(ch.b.x, lm->task_emb + (size_t)task * d, enroll ? lm->enroll_sos : nullptr, ch.eenr, lm->mix_sos,
                                        ch.emix, ch.B, Ne, Nm, d, ch.s);
        if (st == QA_OK) st = lm_body(lm, c, ch.b, ch.B, L, 0, cap, true);
        c.stream = caller;
        QA_TRY(st);
    }

    // ---- decode: G+1 global tokens (the last is fed to the cache but discarded), then S semantic tokens
    int pos = L;
    const bool graphs = !capturing && (multi || use_graphs());
    auto phase = [&](int which, long long first_id, int steps, int lo, int width, int keep) -> int {
        const int ids_ld = keep;
        for (Chain& ch : chains)
            QA_TRY(launch_lm_phase_init(ch.b.tok, first_id, ch.B, ch.b.state, pos, which == 0, sc.seed, ch.b0, ch.s));
        if (graphs && steps > 0) {
            if ((int)lm->graphs.size() < 2 * nc) lm->graphs.resize(2 * nc);
            for (int i = 0; i < nc; ++i) {
                Chain& ch = chains[i];
                long long* ids = which == 0 ? ch.b.ids_g : ch.b.ids_s;
                StepGraph& g = lm->graphs[2 * i + which];
                uint64_t key = 0x51ull;
                for (uint64_t v : {(uint64_t)(uintptr_t)lm->ws, (uint64_t)B, (uint64_t)nc, (uint64_t)ch.b0, (uint64_t)ch.B, (uint64_t)cap, (uint64_t)L,
                                   (uint64_t)G, (uint64_t)S, (uint64_t)Ne, (uint64_t)Nm, (uint64_t)(enroll != nullptr), (uint64_t)lo, (uint64_t)width, (uint64_t)keep, (uint64_t)sc.do_sample,
                                   (uint64_t)sc.top_k, (uint64_t)(sc.top_p * 1e6f), (uint64_t)(sc.temperature * 1e6f), (uint64_t)knob(K_LM_PF)})
                    key = mix_key(key, v);
                if (!g.exec || g.key != key) {
                    g.reset();
                    if (!lm->cap_stream) QA_HIP(hipStreamCreateWithFlags(&lm->cap_stream, hipStreamNonBlocking));
                    QA_HIP(hipStreamBeginCapture(lm->cap_stream, hipStreamCaptureModeThreadLocal));
                    const int st = fused_step(lm, ch.b, ch.B, lo, width, ids, ids_ld, keep, sc, lm->cap_stream, -1, -1);
                    hipGraph_t graph = nullptr;
                    const hipError_t e = hipStreamEndCapture(lm->cap_stream, &graph);
                    if (st != QA_OK) {
                        if (graph) (void)hipGraphDestroy(graph);
                        return st;
                    }
                    QA_HIP(e);
                    g.graph = graph;
                    QA_HIP(hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0));
                    g.key = key;
                }
            }
            for (int st = 0; st < steps; ++st)  // round-robin: every chain advances one token per turn
                for (int i = 0; i < nc; ++i) QA_HIP(hipGraphLaunch(lm->graphs[2 * i + which].exec, chains[i].s));
            pos += steps;
            return QA_OK;
        }
        for (int st = 0; st < steps; ++st)
            for (Chain& ch : chains)
                QA_TRY(fused_step(lm, ch.b, ch.B, lo, width, which == 0 ? ch.b.ids_g : ch.b.ids_s, ids_ld, keep, sc, ch.s, pos + st, st));
        pos += steps;
        return QA_OK;
    };
    QA_TRY(phase(0, 0, G + 1, 3, sp.global_size, G));                            // llm_sft.py:137-164
    QA_TRY(phase(1, 1, S, 3 + sp.global_size, sp.semantic_size, S));            // llm_sft.py:166-193
    for (int i = 0; i < nc; ++i) {
        Chain& ch = chains[i];
        if (G > 0)
            QA_HIP(hipMemcpyAsync(gids + (size_t)ch.b0 * G, ch.b.ids_g, sizeof(long long) * (size_t)ch.B * G, hipMemcpyDeviceToDevice, ch.s));
        if (S > 0)
            QA_HIP(hipMemcpyAsync(sids + (size_t)ch.b0 * S, ch.b.ids_s, sizeof(long long) * (size_t)ch.B * S, hipMemcpyDeviceToDevice, ch.s));
        if (multi) {  // join: the caller's stream continues behind every chain
            QA_HIP(hipEventRecord(lm->chain_join[i], ch.s));
            QA_HIP(hipStreamWaitEvent(caller, lm->chain_join[i], 0));
        }
    }
    return QA_OK;
}

int ensure_ws(qa_lm* lm, size_t bytes) {
    if (bytes <= lm->ws_cap) return QA_OK;
    QA_HIP(hipDeviceSynchronize());  // earlier calls may still be running out of the old workspace
    for (StepGraph& g : lm->graphs) g.reset();  // captured steps point into the old workspace
    if (lm->ws) QA_HIP(hipFree(lm->ws));
    lm->ws = nullptr;
    lm->ws_cap = 0;
    const size_t cap = bytes + bytes / 8;
    QA_HIP(hipMalloc(reinterpret_cast<void**>(&lm->ws), cap));
    lm->ws_cap = cap;
    return QA_OK;
}

}  // namespace

extern "C" {

int qa_lm_create(qa_lm** out, const qa_lm_spec* spec, const qa_tensor* tensors, int64_t n_tensors, int device) {
    if (!out || !spec || !tensors) {
        set_error("qa_lm_create: null argument");
        return QA_ERR_INVALID;
    }
    *out = nullptr;
    QA_HIP(hipSetDevice(device));
    std::unique_ptr<qa_lm> lm(new qa_lm());
    lm->spec = *spec;
    lm->device = device;
    HostTable tab(tensors, n_tensors);
    const int st = build_lm(lm.get(), tab);
    if (st != QA_OK) {
        lm->store.release();
        return st;
    }
    *out = lm.release();
    return QA_OK;
}

void qa_lm_destroy(qa_lm* lm) {
    if (!lm) return;
    (void)hipSetDevice(lm->device);
    (void)hipDeviceSynchronize();
    for (StepGraph& g : lm->graphs) g.reset();
    if (lm->cap_stream) (void)hipStreamDestroy(lm->cap_stream);
    for (hipStream_t st : lm->chain_streams) (void)hipStreamDestroy(st);
    for (hipEvent_t ev : lm->chain_join) (void)hipEventDestroy(ev);
    if (lm->ev_fork) (void)hipEventDestroy(lm->ev_fork);
    lm->store.release();
    if (lm->ws) (void)hipFree(lm->ws);
    delete lm;
}

static int lm_generate_impl(qa_lm* lm, int32_t task, const float* enroll_feats, int64_t n_enroll, const float* mix_feats,
                            int64_t n_mix, int64_t B, int32_t global_length, int32_t semantic_length, const SampleCfg& sc,
                            int64_t* global_ids, int64_t* semantic_ids, void* stream) {
    if (!lm || !mix_feats || !global_ids || !semantic_ids) {
        set_error("qa_lm_generate: null argument");
        return QA_ERR_INVALID;
    }
    QA_REQUIRE(task >= 0 && task < lm->spec.num_tasks, "qa_lm_generate: task %d out of range (KeyError in the reference)", task);
    QA_REQUIRE(B > 0 && n_mix > 0 && global_length >= 0 && semantic_length >= 0, "qa_lm_generate: bad shape");
    QA_REQUIRE(!enroll_feats || n_enroll > 0, "qa_lm_generate: enrollment given with no frames");
    QA_REQUIRE(sc.temperature > 0.f && sc.temperature <= 1.0f, "qa_lm_generate: temperature must be in (0, 1] (llm.py:278)");
    QA_REQUIRE(sc.top_k >= 0 && sc.top_p > 0.f, "qa_lm_generate: bad top_k / top_p");
    QA_HIP(hipSetDevice(lm->device));
    if (sc.do_sample) QA_TRY(lm_sample_prepare());
    Ctx& c = lm->ctx;
    c.stream = static_cast<hipStream_t>(stream);
    c.dry = true;
    c.arena.begin(nullptr, 0);
    QA_TRY(generate_graph(lm, c, task, enroll_feats, (int)n_enroll, mix_feats, (int)n_mix, (int)B, global_length, semantic_length,
                          (long long*)global_ids, (long long*)semantic_ids, sc));
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(c.stream, &cs) != hipSuccess) (void)hipGetLastError();
        // growing the workspace synchronises the device and allocates: neither is possible inside a caller's stream capture
        QA_REQUIRE(cs != hipStreamCaptureStatusActive || c.arena.peak() <= lm->ws_cap,
                   "qa_lm_generate under a stream capture needs %zu bytes of workspace, the handle holds %zu: make one call of the same shape outside "
                   "the capture first (with QA_LM_CHAINS=1 for batches above 32: a capturing caller gets the single-chain launches)",
                   c.arena.peak(), lm->ws_cap);
    }
    QA_TRY(ensure_ws(lm, c.arena.peak()));
    c.dry = false;
    c.arena.begin(lm->ws, lm->ws_cap);
    int st = generate_graph(lm, c, task, enroll_feats, (int)n_enroll, mix_feats, (int)n_mix, (int)B, global_length, semantic_length,
                            (long long*)global_ids, (long long*)semantic_ids, sc);
    if (st != QA_OK) {  // an error between the fork and the join of a multi-chain call: the chains' streams may still be running out of the
        c.stream = static_cast<hipStream_t>(stream);  // workspace the next call re-uses - quiesce them (error path only)
        for (hipStream_t cs : lm->chain_streams) (void)hipStreamSynchronize(cs);
    }
    return st;
}

int qa_lm_generate(qa_lm* lm, int32_t task, const float* enroll_feats, int64_t n_enroll, const float* mix_feats,
                   int64_t n_mix, int64_t B, int32_t global_length, int32_t semantic_length, float temperature,
                   int32_t top_k, float top_p, int64_t* global_ids, int64_t* semantic_ids, void* stream) {
    const SampleCfg sc{0, top_k, top_p, temperature, 0ull};
    return lm_generate_impl(lm, task, enroll_feats, n_enroll, mix_feats, n_mix, B, global_length, semantic_length, sc, global_ids,
                            semantic_ids, stream);
}

int qa_lm_generate_sampled(qa_lm* lm, int32_t task, const float* enroll_feats, int64_t n_enroll, const float* mix_feats,
                           int64_t n_mix, int64_t B, int32_t global_length, int32_t semantic_length, float temperature,
                           int32_t top_k, float top_p, uint64_t seed, int64_t* global_ids, int64_t* semantic_ids, void* stream) {
    const SampleCfg sc{1, top_k, top_p, temperature, (unsigned long long)seed};
    return lm_generate_impl(lm, task, enroll_feats, n_enroll, mix_feats, n_mix, B, global_length, semantic_length, sc, global_ids,
                            semantic_ids, stream);
}

int qa_sample_logits(const float* logits, int64_t B, int64_t width, int64_t ld, int32_t top_k, float top_p, float temperature,
                     int32_t do_sample, uint64_t seed, int64_t* out_index, void* stream) {
    if (!logits || !out_index) {
        set_error("qa_sample_logits: null argument");
        return QA_ERR_INVALID;
    }
    QA_REQUIRE(B > 0 && width > 0 && ld >= width, "qa_sample_logits: bad shape");
    QA_REQUIRE(temperature > 0.f && temperature <= 1.0f, "qa_sample_logits: temperature must be in (0, 1] (llm.py:278)");
    QA_REQUIRE(top_k >= 0 && top_p > 0.f, "qa_sample_logits: bad top_k / top_p");
    hipStream_t s = static_cast<hipStream_t>(stream);
    QA_TRY(lm_sample_prepare());
    char* scratch = nullptr;
    QA_HIP(hipMalloc(reinterpret_cast<void**>(&scratch), 256 + sizeof(long long) * (size_t)B));
    int* state = reinterpret_cast<int*>(scratch);
    long long* tok = reinterpret_cast<long long*>(scratch + 256);
    int st = launch_lm_phase_init(tok, 0, (int)B, state, 0, 1, (unsigned long long)seed, 0, s);
    if (st == QA_OK)
        st = launch_lm_sample(logits, ld, (int)width, (int)B, 0, top_k, top_p, temperature, do_sample, tok, (long long*)out_index, 1, 1,
                              state, s);
    const hipError_t e = hipStreamSynchronize(s);
    (void)hipFree(scratch);
    if (st != QA_OK) return st;
    QA_HIP(e);
    return QA_OK;
}

}  // extern "C"
