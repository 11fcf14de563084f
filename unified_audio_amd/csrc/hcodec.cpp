// hcodec.cpp - H-Codec 1.0 encode / decode graphs on top of the HIP kernels (host orchestration, C-ABI handle).
//
// Mirrors, stage for stage, Codec.encode / Codec.decode of the reference
// (QuarkAudio-HCodec/HCodec-1.0/vq/codec.py:166-187) with every activation kept time-major / channel-last
// ([B, frames, C] rows) so that all convolutions and linears are one implicit-GEMM kernel (conv_gemm.hip).
// Weights arrive in the reference's state_dict layout; weight-norm is folded and filters re-laid out here, once.
#include <memory>

#include "host_util.h"

namespace qa {

namespace {

struct TransformerLayerW {
    const float *ln1, *ln2;
    ConvW ih;            // [4d][d] rows in (unit, gate) order, bias = b_ih + b_hh
    const float* w_hh;   // [4d][d] rows in (unit, gate) order
    ConvW qkv, o, w1, w3, w2;
};

struct TransformerW {
    std::vector<TransformerLayerW> layers;
    int d = 0, heads = 0, inter = 0;
    const float* rope = nullptr;  // [MAX_POS][hd/2][2]
};

struct ResBlockW {  // SEANetResnetBlock
    ConvW k3, pw, sc;
};
struct DecResW {  // ResnetBlock (GroupNorm + swish)
    const float *n1w, *n1b, *n2w, *n2b;
    ConvW c1, c2;
};
struct ConvNeXtW {
    const float *dw, *dwb, *lnw, *lnb, *gamma;
    ConvW pw1, pw2;
};

struct MimiLayerW {  // StreamingTransformerLayer (mimi/transformer.py:436-594)
    const float *n1w, *n1b, *n2w, *n2b, *ls1, *ls2;
    ConvW in_proj, out_proj, lin1, lin2;
};
struct MimiW {
    std::vector<MimiLayerW> layers;
    int d = 0, heads = 0, ff = 0;
    int causal = 0, context = 0;  // StreamingMultiheadAttention causal / context (mimi/transformer.py:403-413); context is ignored unless causal
    const float* rope = nullptr;  // interleaved-pair table [MAX_POS][hd/2][2]
};
// _MHAState of every layer (mimi/transformer.py:284-293): RingKVCache [B, cap, d] x 2 and the offset
struct MimiStream {
    std::vector<float*> kc, vc;
    int B = 0, cap = 0, offset = 0;
    // streaming_forever(): past the static RoPE table the step reads a rolling window of it (positions rope_base .. rope_base + len)
    // rebuilt on the host whenever the stream leaves it, so the offset is unbounded like the reference's (module/rope.py computes
    // the angles from the running offset)
    float* rope_win = nullptr;
    int rope_base = 0, rope_len = 0;
};

constexpr int MAX_POS = 8192;

}  // namespace

}  // namespace qa

using namespace qa;

struct qa_hcodec {
    qa_hcodec_spec spec{};
    int device = 0;
    WeightStore store;
    // encoder
    const float *conv0_w = nullptr, *conv0_b = nullptr;
    std::vector<ResBlockW> res;
    std::vector<ConvW> down;
    TransformerW enc_tr;
    ConvW enc_out;
    // semantic encoder
    ConvW sem_in;
    struct SemBlock {
        ConvW u1[2], u2[2], conv;
        int stride;
    };
    std::vector<SemBlock> sem_blocks;
    ConvW sem_out;
    // rvq
    const float *cb_a = nullptr, *cb_s = nullptr, *e2_a = nullptr, *e2_s = nullptr;
    float* e2_dev = nullptr;
    // decoder
    ConvW up;
    const float *up_dw = nullptr, *up_dwb = nullptr;
    DecResW dres[4];
    TransformerW dec_tr;
    const float *gn_w = nullptr, *gn_b = nullptr, *norm_w = nullptr, *norm_b = nullptr, *fnorm_w = nullptr,
                *fnorm_b = nullptr;
    std::vector<ConvNeXtW> cnx;
    ConvW head, basis;
    const float* window = nullptr;
    int spec_ld = 0;
    // H-Codec 2.0 encoder
    ConvW stft_basis, enc_embed, enc_out20, dec_embed20;
    const float *enc_norm_w = nullptr, *enc_norm_b = nullptr, *enc_fnorm_w = nullptr, *enc_fnorm_b = nullptr;
    std::vector<ConvNeXtW> enc_cnx;
    int stft_ld = 0;
    // H-Codec 1.5
    MimiW agg_sem, agg_ac, bottleneck;
    const float *qemb_sem = nullptr, *qemb_ac = nullptr;
    int* host_sync = nullptr;  // pinned host scalar for the data-dependent group / frame counts
    hipStream_t side = nullptr;  // second stream: the two aggregator stacks are independent and run concurrently
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // workspace
    char* ws = nullptr;
    size_t ws_cap = 0;
    Ctx ctx;
};

// One StreamingTransformer with its optional streaming state (mimi/transformer.py:605-698)
struct qa_mimi {
    qa_mimi_spec spec{};
    int device = 0;
    WeightStore store;
    MimiW w;
    MimiStream st;       // st.B > 0: inside `with model.streaming(B)`
    float* ring = nullptr;  // backing store of the ring caches
    char* ws = nullptr;
    size_t ws_cap = 0;
    Ctx ctx;
};

namespace qa {
namespace {

// ---------------------------------------------------------------- weight folding (host)

// apply_rope (mimi/module/rope.py:38-56): freqs = exp(i * (-ln(P) * 2 / D)), angle = freqs * t, all in fp32; (cos, sin) pairs of
// positions pos0 .. pos0 + n - 1 as [n][hd/2][2]
void mimi_rope_table(int hd, int pos0, int n, std::vector<float>* cs) {
    const int half = hd / 2;
    cs->resize((size_t)n * half * 2);
    const float coef = (float)(-std::log(10000.0) * 2.0 / hd);
    for (int i = 0; i < half; ++i) {
        const float fr = std::exp((float)i * coef);
        for (int t = 0; t < n; ++t) {
            const float ang = fr * (float)(pos0 + t);
            (*cs)[((size_t)t * half + i) * 2] = (float)std::cos((double)ang);
            (*cs)[((size_t)t * half + i) * 2 + 1] = (float)std::sin((double)ang);
        }
    }
}

struct Folder {
    const HostTable& tab;
    WeightStore& store;
    bool ok = true;
    int status = QA_OK;

    const float* need(const std::string& name, int64_t numel) {
        const float* p = tab.get(name, numel);
        if (!p) {
            ok = false;
            status = QA_ERR_MISSING;
        }
        return p;
    }
    size_t vec(const std::string& name, int64_t n) {
        const float* p = need(name, n);
        if (!p) return 0;
        return store.add(p, n);
    }
    // Conv weight [N][C_in][k] (optionally weight-normed: prefix.weight_g / weight_v) -> [Np][k][Cp], bias -> [Np]
    // returns offsets through the out params (pointers are resolved after upload)
    void conv(const std::string& p, int N, int C_in, int k, bool wn, bool bias, int Np, int Cp, size_t* w_off,
              size_t* b_off) {
        std::vector<float> w((size_t)Np * k * Cp, 0.f);
        const int64_t numel = (int64_t)N * C_in * k;
        const float* v = need(p + (wn ? ".weight_v" : ".weight"), numel);
        const float* g = wn ? need(p + ".weight_g", N) : nullptr;
        if (v && (!wn || g)) {
            for (int n = 0; n < N; ++n) {
                float scale = 1.f;
                if (wn) {
                    // torch._weight_norm: w = v * (g / ||v||_2), norm over (C_in, k), computed in fp32
                    double ss = 0.0;
                    for (int64_t i = 0; i < (int64_t)C_in * k; ++i) {
                        const float a = v[(int64_t)n * C_in * k + i];
                        ss += (double)a * a;
                    }
                    scale = g[n] / (float)std::sqrt(ss);
                }
                for (int c = 0; c < C_in; ++c)
                    for (int j = 0; j < k; ++j)
                        w[((size_t)n * k + j) * Cp + c] = v[((int64_t)n * C_in + c) * k + j] * scale;
            }
        }
        *w_off = store.add(w);
        if (bias) {
            std::vector<float> b(Np, 0.f);
            const float* bp = need(p + ".bias", N);
            if (bp) std::memcpy(b.data(), bp, sizeof(float) * N);
            *b_off = store.add(b);
        }
    }
};

struct PendingConv {
    ConvW* dst;
    size_t w_off, b_off;
    bool has_bias;
};

struct Builder {
    Folder f;
    std::vector<PendingConv> pend;
    std::vector<std::pair<const float**, size_t>> pend_vec;

    void conv(ConvW* dst, const std::string& p, int N, int C_in, int k, bool wn, bool bias, int Np = -1, int Cp = -1) {
        if (Np < 0) Np = N;
        if (Cp < 0) Cp = C_in;
        PendingConv pc{dst, 0, 0, bias};
        f.conv(p, N, C_in, k, wn, bias, Np, Cp, &pc.w_off, &pc.b_off);
        dst->N = Np;
        dst->C_in = Cp;
        dst->ksize = k;
        dst->algo_n = N;
        dst->algo_cin = C_in;
        pend.push_back(pc);
    }
    void vec(const float** dst, const std::string& name, int64_t n) { pend_vec.push_back({dst, f.vec(name, n)}); }
    void raw(const float** dst, const std::vector<float>& v) { pend_vec.push_back({dst, f.store.add(v)}); }
    void resolve() {
        for (auto& pc : pend) {
            pc.dst->w = f.store.ptr(pc.w_off);
            pc.dst->b = pc.has_bias ? f.store.ptr(pc.b_off) : nullptr;
        }
        for (auto& pv : pend_vec) *pv.first = f.store.ptr(pv.second);
    }
};

// rows of an LSTM matrix / bias go from PyTorch's gate-major order (i,f,g,o blocks of d) to (unit, gate)
void build_transformer(Builder& b, TransformerW* tw, const std::string& p, int d, int n_layers, int heads, int inter = 0) {
    if (inter <= 0) inter = 4 * d;
    tw->d = d;
    tw->heads = heads;
    tw->inter = inter;
    tw->layers.resize(n_layers);
    for (int l = 0; l < n_layers; ++l) {
        TransformerLayerW& L = tw->layers[l];
        const std::string lp = p + ".layers." + std::to_string(l);
        const std::string ap = lp + ".self_attn";
        b.vec(&L.ln1, lp + ".input_layernorm.weight", d);
        b.vec(&L.ln2, lp + ".post_attention_layernorm.weight", d);
        const float* wih = b.f.need(ap + ".rnn.weight_ih_l0", (int64_t)4 * d * d);
        const float* whh = b.f.need(ap + ".rnn.weight_hh_l0", (int64_t)4 * d * d);
        const float* bih = b.f.need(ap + ".rnn.bias_ih_l0", 4 * d);
        const float* bhh = b.f.need(ap + ".rnn.bias_hh_l0", 4 * d);
        std::vector<float> wi((size_t)4 * d * d, 0.f), wh((size_t)4 * d * d, 0.f), bb((size_t)4 * d, 0.f);
        if (wih && whh && bih && bhh) {
            for (int u = 0; u < d; ++u)
                for (int g = 0; g < 4; ++g) {
                    const size_t dst = (size_t)u * 4 + g, src = (size_t)g * d + u;
                    std::memcpy(&wi[dst * d], &wih[src * d], sizeof(float) * d);
                    std::memcpy(&wh[dst * d], &whh[src * d], sizeof(float) * d);
                    bb[dst] = bih[src] + bhh[src];
                }
        }
        L.ih.N = 4 * d; L.ih.C_in = d; L.ih.ksize = 1;
        b.raw(&L.ih.w, wi);
        b.raw(&L.ih.b, bb);
        b.raw(&L.w_hh, wh);
        // fused QKV
        std::vector<float> wq((size_t)3 * d * d, 0.f), bq((size_t)3 * d, 0.f);
        const char* names[3] = {".q_proj", ".k_proj", ".v_proj"};
        for (int i = 0; i < 3; ++i) {
            const float* w = b.f.need(ap + names[i] + ".weight", (int64_t)d * d);
            const float* bi = b.f.need(ap + names[i] + ".bias", d);
            if (w) std::memcpy(&wq[(size_t)i * d * d], w, sizeof(float) * d * d);
            if (bi) std::memcpy(&bq[(size_t)i * d], bi, sizeof(float) * d);
        }
        L.qkv.N = 3 * d; L.qkv.C_in = d; L.qkv.ksize = 1;
        b.raw(&L.qkv.w, wq);
        b.raw(&L.qkv.b, bq);
        L.o.N = d; L.o.C_in = d; L.o.ksize = 1;
        b.vec(&L.o.w, ap + ".o_proj.weight", (int64_t)d * d);
        L.w1.N = inter; L.w1.C_in = d; L.w1.ksize = 1;
        b.vec(&L.w1.w, lp + ".mlp.w1.weight", (int64_t)inter * d);
        L.w3.N = inter; L.w3.C_in = d; L.w3.ksize = 1;
        b.vec(&L.w3.w, lp + ".mlp.w3.weight", (int64_t)inter * d);
        L.w2.N = d; L.w2.C_in = inter; L.w2.ksize = 1;
        b.vec(&L.w2.w, lp + ".mlp.w2.weight", (int64_t)inter * d);
    }
    // RoPE table, RotaryEmbedding of transformer.py:8-74 evaluated in fp32 like the reference
    const int hd = d / heads, half = hd / 2;
    std::vector<float> cs((size_t)MAX_POS * half * 2);
    for (int i = 0; i < half; ++i) {
        const float expo = (float)(2 * i) / (float)hd;
        const float inv = 1.0f / std::pow(10000.0f, expo);
        for (int t = 0; t < MAX_POS; ++t) {
            const float fr = (float)t * inv;
            cs[((size_t)t * half + i) * 2] = (float)std::cos((double)fr);
            cs[((size_t)t * half + i) * 2 + 1] = (float)std::sin((double)fr);
        }
    }
    b.raw(&tw->rope, cs);
}

void build_mimi(Builder& b, MimiW* mw, const std::string& p, int d, int n_layers, int heads, int ff, int causal = 0,
                int context = 0) {
    mw->d = d;
    mw->heads = heads;
    mw->ff = ff;
    mw->causal = causal;
    mw->context = context;
    mw->layers.resize(n_layers);
    for (int l = 0; l < n_layers; ++l) {
        MimiLayerW& L = mw->layers[l];
        const std::string lp = p + ".layers." + std::to_string(l);
        b.vec(&L.n1w, lp + ".norm1.weight", d);
        b.vec(&L.n1b, lp + ".norm1.bias", d);
        b.vec(&L.n2w, lp + ".norm2.weight", d);
        b.vec(&L.n2b, lp + ".norm2.bias", d);
        b.vec(&L.ls1, lp + ".layer_scale_1.scale", d);
        b.vec(&L.ls2, lp + ".layer_scale_2.scale", d);
        L.in_proj.N = 3 * d; L.in_proj.C_in = d;
        b.vec(&L.in_proj.w, lp + ".self_attn.in_proj_weight", (int64_t)3 * d * d);
        L.out_proj.N = d; L.out_proj.C_in = d;
        b.vec(&L.out_proj.w, lp + ".self_attn.out_proj.weight", (int64_t)d * d);
        L.lin1.N = ff; L.lin1.C_in = d;
        b.vec(&L.lin1.w, lp + ".linear1.weight", (int64_t)ff * d);
        L.lin2.N = d; L.lin2.C_in = ff;
        b.vec(&L.lin2.w, lp + ".linear2.weight", (int64_t)ff * d);
    }
    std::vector<float> cs;
    mimi_rope_table(d / heads, 0, MAX_POS, &cs);
    b.raw(&mw->rope, cs);
}

// ---------------------------------------------------------------- graph helpers

int conv_op(Ctx& c, const float* x, int64_t ldx, int B, int T_in, const ConvW& w, float* y, int64_t ldy, int T_out,
            int stride, int pad_left, int pad_right, int pad_mode, int prologue, int act, const float* gamma,
            const float* res, int64_t ldr, const float* gate, int post_act, int in_rep = 1, const float* rope = nullptr,
            int rope_n = 0, int rope_hd = 0, int rope_T = 0, int rope_pos0 = 0) {
    if (c.dry) return QA_OK;
    qa_conv_args a{};
    a.x = x; a.w = w.w; a.bias = w.b; a.gamma = gamma; a.residual = res; a.gate = gate; a.y = y;
    a.B = B; a.T_in = T_in; a.C_in = w.C_in; a.T_out = T_out; a.N = w.N;
    a.ldx = ldx; a.ldy = ldy; a.ldr = ldr; a.ldg = w.N;
    a.ksize = w.ksize; a.stride = stride; a.pad_left = pad_left; a.pad_right = pad_right; a.pad_mode = pad_mode;
    a.prologue = prologue; a.act = act; a.post_act = post_act;
    a.in_rep = in_rep;
    ConvParams p;
    QA_TRY(conv_params_from_args(a, &p));
    p.algo_n = w.algo_n;
    p.algo_k = w.algo_cin ? w.algo_cin * w.ksize : 0;
    p.rope = rope; p.rope_n = rope_n; p.rope_hd = rope_hd; p.rope_T = rope_T; p.rope_pos0 = rope_pos0;
    return launch_conv_gemm(p, c.stream);
}

// plain linear over `rows` rows
int linear_op(Ctx& c, const float* x, int64_t rows, const ConvW& w, float* y, int act = ACT_NONE,
              const float* res = nullptr, const float* gate = nullptr, const float* gamma = nullptr) {
    return conv_op(c, x, w.C_in, 1, (int)rows, w, y, w.N, (int)rows, 1, 0, 0, PAD_ZERO, ACT_NONE, act, gamma, res, w.N,
                   gate, ACT_NONE);
}

// "same" zero-padded stride-1 conv (vq/conv.py:33-56, semantic_module.py:28-31)
int conv_same(Ctx& c, const float* x, int B, int T, const ConvW& w, float* y, int prologue = ACT_NONE,
              int act = ACT_NONE, const float* res = nullptr, bool causal = false) {
    const int pad = (w.ksize - 1) / 2;
    return conv_op(c, x, w.C_in, B, T, w, y, w.N, T, 1, causal ? w.ksize - 1 : pad, causal ? 0 : pad, PAD_ZERO, prologue, act,
                   nullptr, res, w.N, nullptr, ACT_NONE);
}

// SConv1d geometry (encoder_modules/conv.py:195-211, non-causal): returns T_out and the paddings
struct SGeom {
    int T_out, left, right;
};
SGeom sconv_geom(int L, int k, int stride, bool causal = false) {
    int64_t t = 0;
    int32_t left = 0, right = 0;
    (void)qa_sconv_geometry(L, k, stride, &t, &left, &right);
    if (causal) {  // conv.py:203-206: pad (padding_total, extra_padding)
        right -= (k - stride) / 2;
        left = k - stride;
    }
    return {(int)t, left, right};
}
// zero padding of vq/conv.py's Conv1d (stride 1, :44-47): (k - 1, 0) if causal else (k / 2, k / 2)
inline int zpad_left(int k, bool causal) { return causal ? k - 1 : k / 2; }
inline int zpad_right(int k, bool causal) { return causal ? 0 : k / 2; }

int transformer_op(Ctx& c, const TransformerW& tw, float* x, int B, int N, const std::string& tap_prefix, bool causal = false) {
    const int d = tw.d, H = tw.heads, hd = d / H;
    const int64_t rows = (int64_t)B * N;
    QA_REQUIRE(N <= MAX_POS, "transformer: sequence of %d frames exceeds %d", N, MAX_POS);
    const size_t mark = c.arena.mark();
    float* hn = c.arena.alloc<float>(rows * d);
    const int wide = std::max(4 * d, tw.inter);
    float* big = c.arena.alloc<float>(rows * wide);   // xw (4d) / gate buffer
    float* big2 = c.arena.alloc<float>(rows * wide);  // swiglu product
    float* hl = c.arena.alloc<float>(rows * d);
    float* qkv = c.arena.alloc<float>(rows * 3 * d);
    float* att = c.arena.alloc<float>(rows * d);
    float* cst = c.arena.alloc<float>((size_t)B * d);
    // taps allocate from the arena: they are issued in the planning pass too (RUN skips only the launches there)
#define RUN(expr) do { if (!c.dry) QA_TRY(expr); } while (0)
    for (size_t l = 0; l < tw.layers.size(); ++l) {
        const TransformerLayerW& L = tw.layers[l];
        const std::string lp = tap_prefix + ".layers." + std::to_string(l);
        RUN(launch_rmsnorm(x, L.ln1, hn, rows, d, 1e-6f, c.stream));
        RUN(linear_op(c, hn, rows, L.ih, big));
        RUN(launch_lstm(big, L.w_hh, hl, cst, B, N, d, c.stream));
        c.tap(lp + ".self_attn.rnn", hl, rows * d);
        RUN(linear_op(c, hl, rows, L.qkv, qkv));
        RUN(launch_rope(qkv, tw.rope, B, N, H, hd, 3 * d, 0, c.stream));
        RUN(launch_attention(qkv, 3 * d, qkv + d, qkv + 2 * d, 3 * d, att, d, B, N, N, (long long)N * 3 * d, H, hd,
                             1.0f / std::sqrt((float)hd), causal ? 1 : 0, c.stream));
        c.tap(lp + ".att", att, rows * d);
        RUN(linear_op(c, att, rows, L.o, x, ACT_NONE, x));
        c.tap(lp + ".x_attn", x, rows * d);
        RUN(launch_rmsnorm(x, L.ln2, hn, rows, d, 1e-6f, c.stream));
        RUN(linear_op(c, hn, rows, L.w1, big));
        RUN(linear_op(c, hn, rows, L.w3, big2, ACT_NONE, nullptr, big));
        RUN(linear_op(c, big2, rows, L.w2, x, ACT_NONE, x));
        c.tap(lp + ".x_mlp", x, rows * d);
    }
#undef RUN
    c.arena.release(mark);
    return QA_OK;
}

// StreamingTransformer.forward, non-causal / non-streaming (mimi/transformer.py:377-425,553-594,674-698), in place on x
struct MimiTemps {
    float *hn, *qkv, *att, *u;
};
MimiTemps mimi_temps(Ctx& c, const MimiW& mw, int64_t rows) {
    MimiTemps t;
    t.hn = c.arena.alloc<float>(rows * mw.d);
    t.qkv = c.arena.alloc<float>(rows * 3 * mw.d);
    t.att = c.arena.alloc<float>(rows * mw.d);
    t.u = c.arena.alloc<float>(rows * mw.ff);
    return t;
}
// st != nullptr: streaming step of N frames at st->offset (layer index li selects the ring caches); the caller advances the offset
int mimi_layer(Ctx& c, const MimiW& mw, const MimiLayerW& L, float* x, const MimiTemps& t, int B, int N, MimiStream* st = nullptr,
               size_t li = 0) {
    const int d = mw.d, H = mw.heads, hd = d / H;
    const int64_t rows = (int64_t)B * N;
    const int pos0 = st ? st->offset : 0;
    const bool win = st && st->rope_len > 0;  // RoPE angles from the rolling window (positions beyond the static table)
    const float* rope = win ? st->rope_win : mw.rope;
    const int rope_pos0 = win ? pos0 - st->rope_base : pos0;
    const float scale = 1.0f / std::sqrt((float)hd);
    QA_TRY(launch_layernorm(x, L.n1w, L.n1b, t.hn, rows, d, 1e-5f, c.stream));
    // fused QKV projection with the interleaved-pair RoPE of q and k applied in the GEMM epilogue (one launch less per layer)
    QA_TRY(conv_op(c, t.hn, d, 1, (int)rows, L.in_proj, t.qkv, 3 * d, (int)rows, 1, 0, 0, PAD_ZERO, ACT_NONE, ACT_NONE, nullptr, nullptr,
                   3 * d, nullptr, ACT_NONE, 1, rope, 2 * d, hd, N, rope_pos0));
    if (st) {
        // RingKVCache.complete(): the chunk's keys / values are written first, then every query attends over the ring
        QA_TRY(launch_ring_append(t.qkv + d, t.qkv + 2 * d, 3 * d, st->kc[li], st->vc[li], B, N, d, st->cap, pos0, c.stream));
        QA_TRY(launch_attention(t.qkv, 3 * d, st->kc[li], st->vc[li], d, t.att, d, B, N, st->cap, (long long)st->cap * d, H, hd, scale, 1,
                                c.stream, nullptr, nullptr, 0, mw.context, pos0, pos0 + N));
    } else {
        QA_TRY(launch_attention(t.qkv, 3 * d, t.qkv + d, t.qkv + 2 * d, 3 * d, t.att, d, B, N, N, (long long)N * 3 * d, H, hd, scale,
                                mw.causal, c.stream, nullptr, nullptr, 0, mw.causal ? mw.context : 0));
    }
    QA_TRY(linear_op(c, t.att, rows, L.out_proj, x, ACT_NONE, x, nullptr, L.ls1));
    QA_TRY(launch_layernorm(x, L.n2w, L.n2b, t.hn, rows, d, 1e-5f, c.stream));
    QA_TRY(linear_op(c, t.hn, rows, L.lin1, t.u, ACT_GELU));
    return linear_op(c, t.u, rows, L.lin2, x, ACT_NONE, x, nullptr, L.ls2);
}
int mimi_op(Ctx& c, const MimiW& mw, float* x, int B, int N) {
    QA_REQUIRE(N <= MAX_POS, "mimi transformer: sequence of %d tokens exceeds %d", N, MAX_POS);
    const size_t mark = c.arena.mark();
    const MimiTemps t = mimi_temps(c, mw, (int64_t)B * N);
    if (!c.dry)
        for (const MimiLayerW& L : mw.layers) QA_TRY(mimi_layer(c, mw, L, x, t, B, N));
    c.arena.release(mark);
    return QA_OK;
}
// two independent stacks of equal depth, layer-interleaved on two streams (xa on the caller's stream, xb on `side`)
int mimi_pair_op(Ctx& c, hipStream_t side, const MimiW& wa, float* xa, const MimiW& wb, float* xb, int B, int N) {
    QA_REQUIRE(N <= MAX_POS && wa.layers.size() == wb.layers.size(), "mimi pair: mismatched stacks");
    const size_t mark = c.arena.mark();
    const MimiTemps ta = mimi_temps(c, wa, (int64_t)B * N);
    const MimiTemps tb = mimi_temps(c, wb, (int64_t)B * N);
    if (!c.dry) {
        hipStream_t main = c.stream;
        for (size_t l = 0; l < wa.layers.size(); ++l) {
            c.stream = main;
            int st = mimi_layer(c, wa, wa.layers[l], xa, ta, B, N);
            c.stream = side;
            if (st == QA_OK) st = mimi_layer(c, wb, wb.layers[l], xb, tb, B, N);
            c.stream = main;
            QA_TRY(st);
        }
    }
    c.arena.release(mark);
    return QA_OK;
}

int groupnorm_op(Ctx& c, const float* x, const float* w, const float* b, float* y, int B, int T, int C, int G,
                 int swish) {
    const size_t mark = c.arena.mark();
    double* scratch = c.arena.alloc<double>(groupnorm_scratch_bytes(B, T, G) / sizeof(double));
    int st = QA_OK;
    if (!c.dry) st = launch_groupnorm(x, w, b, y, scratch, B, T, C, G, 1e-6f, swish, c.stream);
    c.arena.release(mark);
    return st;
}

int dec_resblock_op(Ctx& c, const DecResW& w, float* x, int B, int T, int C, int G, bool causal = false) {
    const size_t mark = c.arena.mark();
    float* t1 = c.arena.alloc<float>((size_t)B * T * C);
    float* t2 = c.arena.alloc<float>((size_t)B * T * C);
    QA_TRY(groupnorm_op(c, x, w.n1w, w.n1b, t1, B, T, C, G, 1));
    QA_TRY(conv_same(c, t1, B, T, w.c1, t2, ACT_NONE, ACT_NONE, nullptr, causal));
    QA_TRY(groupnorm_op(c, t2, w.n2w, w.n2b, t1, B, T, C, G, 1));
    QA_TRY(conv_same(c, t1, B, T, w.c2, x, ACT_NONE, ACT_NONE, x, causal));
    c.arena.release(mark);
    return QA_OK;
}

// ---------------------------------------------------------------- encode / decode graphs

int convnext_op(Ctx& c, const ConvNeXtW& w, float* x, float* t1, float* u, int B, int T, int d, bool causal = false) {
    const int64_t rows = (int64_t)B * T;
    if (c.dry) return QA_OK;
    QA_TRY(launch_dwconv(x, w.dw, w.dwb, w.lnw, w.lnb, t1, B, T, d, 7, 1e-6f, c.stream, zpad_left(7, causal)));
    QA_TRY(linear_op(c, t1, rows, w.pw1, u, ACT_GELU));
    return linear_op(c, u, rows, w.pw2, x, ACT_NONE, x, nullptr, w.gamma);
}

// H-Codec 2.0 CodecEncoder.forward (HCodec-2.0/vq/codec_encoder.py:62-79): wav [B, T] -> emb [B, T / (hop*stride), code_dim]
int encoder20(qa_hcodec* h, Ctx& c, const float* wav, int B, int T, float** emb_out, int* n50_out, int* nf_out) {
    const qa_hcodec_spec& sp = h->spec;
    const int blk = (sp.n_fft - sp.hop) / 2;  // 480: pad = (n_fft - hop) / 2 and hop = 2 * blk, n_fft = 4 * blk
    const int N50 = T / sp.hop, d = sp.enc_dim, nb = sp.n_fft / 2 + 1;
    const int64_t rows = (int64_t)B * N50;
    // STFT as an implicit GEMM: the signal is a [B, T / blk, blk] "channel-last" tensor, a frame is 4 consecutive blocks
    // starting one block before 2 t (zero padded), the filter bank is the windowed DFT basis (re | im)
    float* ri = c.arena.alloc<float>(rows * 2 * nb);
    QA_TRY(conv_op(c, wav, blk, B, T / blk, h->stft_basis, ri, 2 * nb, N50, 2, 1, 1, PAD_ZERO, ACT_NONE, ACT_NONE, nullptr, nullptr,
                   0, nullptr, ACT_NONE));
    float* feat = c.arena.alloc<float>(rows * h->stft_ld);
    if (!c.dry) QA_TRY(launch_stft_post(ri, feat, rows, nb, 2 * nb, h->stft_ld, c.stream));
    c.tap("enc.stft", feat, rows * h->stft_ld);
    float* x = c.arena.alloc<float>(rows * d);
    float* t1 = c.arena.alloc<float>(rows * d);
    float* u = c.arena.alloc<float>(rows * sp.enc_inter);
    // vq/conv.py Conv1d (:39-47): zero padding (k - stride, 0) in the causal variant, (k / 2, k / 2) otherwise
    const bool cz = sp.causal != 0;
    QA_TRY(conv_op(c, feat, h->stft_ld, B, N50, h->enc_embed, t1, d, N50, 1, cz ? 2 : 1, cz ? 0 : 1, PAD_ZERO, ACT_NONE, ACT_NONE, nullptr,
                   nullptr, 0, nullptr, ACT_NONE));
    if (!c.dry) QA_TRY(launch_layernorm(t1, h->enc_norm_w, h->enc_norm_b, x, rows, d, 1e-6f, c.stream));
    for (const ConvNeXtW& w : h->enc_cnx) QA_TRY(convnext_op(c, w, x, t1, u, B, N50, d, cz));
    c.tap("enc.prior", x, rows * d);
    QA_TRY(transformer_op(c, h->enc_tr, x, B, N50, "encoder.post_net.1", cz));
    if (!c.dry) QA_TRY(launch_layernorm(x, h->enc_fnorm_w, h->enc_fnorm_b, t1, rows, d, 1e-6f, c.stream));
    const int k = h->enc_out20.ksize, st = sp.frame_stride;
    const int pl = cz ? k - st : k / 2, pr = cz ? 0 : k / 2;
    const int Nf = (N50 + pl + pr - k) / st + 1;
    float* emb = c.arena.alloc<float>((size_t)B * Nf * sp.code_dim);
    QA_TRY(conv_op(c, t1, d, B, N50, h->enc_out20, emb, sp.code_dim, Nf, st, pl, pr, PAD_ZERO, ACT_NONE, ACT_NONE, nullptr, nullptr,
                   0, nullptr, ACT_NONE));
    *emb_out = emb;
    *n50_out = N50;
    *nf_out = Nf;
    return QA_OK;
}

// SEANet encoder + semantic encoder: wav, feat -> emb, sem  [B, N25, code_dim] each (codec.py:169-170)
int encode_front(qa_hcodec* h, Ctx& c, const float* wav, int B, int T, const float* feat, int64_t fsb, int64_t fsc,
                 int64_t fst, int n_feat, float** emb_out, float** sem_out, int* n25_out) {
    const qa_hcodec_spec& sp = h->spec;
    float* emb = nullptr;
    int N50 = 0, N25 = 0;
    // ---- semantic encoder (independent of the acoustic branch until the RVQ / alignment)
    float* sem = nullptr;
    int Ls = 0;
    auto semantic_branch = [&]() -> int {
        QA_REQUIRE(n_feat > 0, "encode: feat has no frames");
        const float* f = feat;
        const int SC = sp.sem_ch;
        if (!(fsc == 1 && fst == sp.sem_in && fsb == (int64_t)n_feat * sp.sem_in)) {
            float* fcl = c.arena.alloc<float>((size_t)B * n_feat * sp.sem_in);
            if (!c.dry) QA_TRY(launch_to_channel_last(feat, fsb, fsc, fst, fcl, B, sp.sem_in, n_feat, c.stream));
            f = fcl;
        }
        Ls = n_feat;
        float* s = c.arena.alloc<float>((size_t)B * Ls * SC);
        float* tmp = c.arena.alloc<float>((size_t)B * Ls * SC);
        QA_TRY(conv_same(c, f, B, Ls, h->sem_in, s));
        for (size_t bi = 0; bi < h->sem_blocks.size(); ++bi) {
            const auto& blk = h->sem_blocks[bi];
            for (int u = 0; u < 2; ++u) {
                QA_TRY(conv_same(c, s, B, Ls, blk.u1[u], tmp, ACT_ELU, ACT_ELU));       // ELU(conv1(ELU(s)))
                QA_TRY(conv_same(c, tmp, B, Ls, blk.u2[u], s, ACT_NONE, ACT_NONE, s));  // s + conv2(.)
            }
            const int k = blk.conv.ksize, pad = (k - 1) / 2;
            const int To = (Ls + 2 * pad - k) / blk.stride + 1;
            float* y = c.arena.alloc<float>((size_t)B * To * SC);
            QA_TRY(conv_op(c, s, SC, B, Ls, blk.conv, y, SC, To, blk.stride, pad, pad, PAD_ZERO, ACT_NONE, ACT_NONE, nullptr,
                           nullptr, 0, nullptr, ACT_NONE));
            s = y;
            Ls = To;
        }
        sem = c.arena.alloc<float>((size_t)B * Ls * sp.code_dim);
        return conv_same(c, s, B, Ls, h->sem_out, sem);
    };
    if (sp.version == 20) {
        QA_TRY(encoder20(h, c, wav, B, T, &emb, &N50, &N25));
    } else {
    // ---- SEANet encoder
    int C = sp.n_filters, L = T;
    const bool cz = sp.causal != 0;
    // stage 0 at C = 32: conv0 + residual block + ELU in ONE launch (seanet_front.hip): the [B L, 32] conv0 output never exists in HBM
    const bool front = seanet_front_supported(C, C / 2, L);
    float* x = (front && !c.capture) ? nullptr : c.arena.alloc<float>((size_t)B * L * C);
    if (!c.dry && x) QA_TRY(launch_conv_in(wav, h->conv0_w, h->conv0_b, x, B, L, C, 7, c.stream, cz ? 6 : -1));
    if (x) c.tap("enc.conv0", x, (int64_t)B * L * C);
    for (int i = 0; i < sp.n_ratios; ++i) {
        const int r = sp.ratios[i];
        const ResBlockW& rb = h->res[i];
        const size_t mark = c.arena.mark();
        float* sc = c.arena.alloc<float>((size_t)B * L * C);
        if (i == 0 && front) {
            if (!c.dry)
                QA_TRY(launch_seanet_front(wav, h->conv0_w, h->conv0_b, rb.k3.w, rb.k3.b, rb.sc.w, rb.sc.b, rb.pw.w, rb.pw.b, sc,
                                           B, L, C, C / 2, cz ? 1 : 0, c.stream));
        } else {
        float* hh = c.arena.alloc<float>((size_t)B * L * rb.k3.N);
        // shortcut_1x1(x)
        QA_TRY(conv_op(c, x, C, B, L, rb.sc, sc, C, L, 1, 0, 0, PAD_ZERO, ACT_NONE, ACT_NONE, nullptr, nullptr, 0, nullptr,
                       ACT_NONE));
        // ELU(k3(ELU(x)))  (reflect pad 1,1)
        QA_TRY(conv_op(c, x, C, B, L, rb.k3, hh, rb.k3.N, L, 1, cz ? 2 : 1, cz ? 0 : 1, PAD_REFLECT, ACT_ELU, ACT_ELU, nullptr,
                       nullptr, 0, nullptr, ACT_NONE));
        // ELU(shortcut + 1x1(.))  -> the activation in front of the down-sampling conv is fused here
        QA_TRY(conv_op(c, hh, rb.pw.C_in, B, L, rb.pw, sc, C, L, 1, 0, 0, PAD_ZERO, ACT_NONE, ACT_NONE, nullptr, sc, C,
                       nullptr, ACT_ELU));
        }
        const SGeom g = sconv_geom(L, 2 * r, r, cz);
        // the strided conv writes below the mark: allocate its output after releasing the block temporaries is not
        // possible (sc is its input), so the output is carved above them and compacted by pointer swap.
        float* y = c.arena.alloc<float>((size_t)B * g.T_out * 2 * C);
        QA_TRY(conv_op(c, sc, C, B, L, h->down[i], y, 2 * C, g.T_out, r, g.left, g.right, PAD_REFLECT, ACT_NONE, ACT_NONE,
                       nullptr, nullptr, 0, nullptr, ACT_NONE));
        (void)mark;
        x = y;
        L = g.T_out;
        C *= 2;
        c.tap("enc.stage" + std::to_string(i), x, (int64_t)B * L * C);
    }
    QA_REQUIRE(C == sp.dimension, "encoder: channel ladder ends at %d, spec.dimension is %d", C, sp.dimension);
    N50 = L;
    QA_TRY(transformer_op(c, h->enc_tr, x, B, N50, "encoder.model." + std::to_string(3 * sp.n_ratios + 2), cz));
    c.tap("enc.transformer", x, (int64_t)B * N50 * C);
    const SGeom g = sconv_geom(N50, 4, 2, cz);
    N25 = g.T_out;
    emb = c.arena.alloc<float>((size_t)B * N25 * sp.code_dim);
    QA_TRY(conv_op(c, x, C, B, N50, h->enc_out, emb, sp.code_dim, N25, 2, g.left, g.right, PAD_REFLECT, ACT_ELU, ACT_NONE,
                   nullptr, nullptr, 0, nullptr, ACT_NONE));
    }
    c.tap("enc.emb", emb, (int64_t)B * N25 * sp.code_dim);
    QA_TRY(semantic_branch());
    QA_REQUIRE(Ls == N25, "encode: semantic stream has %d frames, acoustic stream %d (feat must have T/%d frames)", Ls,
               N25, T / std::max(1, N50));
    c.tap("enc.sem", sem, (int64_t)B * Ls * sp.code_dim);
    *emb_out = emb;
    *sem_out = sem;
    *n25_out = N25;
    return QA_OK;
}

int encode_graph(qa_hcodec* h, Ctx& c, const float* wav, int B, int T, const float* feat, int64_t fsb, int64_t fsc,
                 int64_t fst, int n_feat, long long* ac_out, long long* sc_out) {
    const qa_hcodec_spec& sp = h->spec;
    float *emb = nullptr, *sem = nullptr;
    int N25 = 0;
    QA_TRY(encode_front(h, c, wav, B, T, feat, fsb, fsc, fst, n_feat, &emb, &sem, &N25));
    // ---- RVQ (both streams)
    const int Q = sp.num_quantizers;
    long long* ia = c.arena.alloc<long long>((size_t)B * N25 * Q);
    long long* is = c.arena.alloc<long long>((size_t)B * N25 * Q);
    float* rvq_ws = c.arena.alloc<float>(rvq_scratch_floats((long long)B * N25, sp.codebook_size, sp.code_dim));
    if (!c.dry) {
        QA_TRY(launch_rvq_search(emb, (long long)B * N25, h->cb_a, h->e2_a, Q, sp.codebook_size, sp.code_dim, ia, nullptr, 0,
                                 rvq_ws, c.stream));
        QA_TRY(launch_rvq_search(sem, (long long)B * N25, h->cb_s, h->e2_s, Q, sp.codebook_size, sp.code_dim, is, nullptr, 0,
                                 rvq_ws, c.stream));
        QA_TRY(launch_codes_to_bqn(ia, ac_out, B, N25, Q, c.stream));
        QA_TRY(launch_codes_to_bqn(is, sc_out, B, N25, Q, c.stream));
    }
    return QA_OK;
}

int decode_tail(qa_hcodec* h, Ctx& c, const float* cat, int B, int N, float* wav_out);

int decode_graph(qa_hcodec* h, Ctx& c, const long long* ac, const long long* scodes, int B, int N, float* wav_out) {
    const qa_hcodec_spec& sp = h->spec;
    const int Q = sp.num_quantizers, D = sp.code_dim;
    const int64_t rows25 = (int64_t)B * N;
    long long* ia = c.arena.alloc<long long>(rows25 * Q);
    long long* is = c.arena.alloc<long long>(rows25 * Q);
    float* cat = c.arena.alloc<float>(rows25 * 2 * D);
    if (!c.dry) {
        QA_TRY(launch_codes_from_bqn(ac, ia, B, N, Q, c.stream));
        QA_TRY(launch_codes_from_bqn(scodes, is, B, N, Q, c.stream));
        QA_TRY(launch_rvq_lookup(ia, rows25, h->cb_a, Q, sp.codebook_size, D, cat, 2 * D, c.stream));
        QA_TRY(launch_rvq_lookup(is, rows25, h->cb_s, Q, sp.codebook_size, D, cat + D, 2 * D, c.stream));
    }
    return decode_tail(h, c, cat, B, N, wav_out);
}

// CodecDecoder.forward (codec_decoder.py:58-67) from the concatenated [acoustic | semantic] embeddings [B, N, 2*code_dim]
int decode_tail(qa_hcodec* h, Ctx& c, const float* cat, int B, int N, float* wav_out) {
    const qa_hcodec_spec& sp = h->spec;
    const int d = sp.dec_dim;
    const int64_t rows25 = (int64_t)B * N;
    // sub-pixel upsampler: 1x1 conv to 2*d channels; in channel-last layout the pixel shuffle (vq/conv.py:86-88) is a
    // pure reinterpretation [B, N, 2, d] -> [B, 2N, d]
    const bool v20 = sp.version == 20;
    const bool cz = sp.causal != 0;
    const int N50 = (v20 ? sp.frame_stride : 2) * N;
    const int64_t rows = (int64_t)B * N50;
    float* x = c.arena.alloc<float>(rows * d);
    if (v20) {
        // H-Codec 2.0 (codec_decoder.py:30-31,64-65): x.repeat_interleave(s) -> Conv1d k = s + 1, "same" zero padding.  The
        // repetition is folded into the implicit-GEMM gather (frame r reads row r / s), nothing is materialised.
        const int k = h->dec_embed20.ksize;
        QA_TRY(conv_op(c, cat, 2 * sp.code_dim, B, N, h->dec_embed20, x, d, N50, 1, cz ? k - 1 : k / 2, cz ? 0 : k / 2, PAD_ZERO, ACT_NONE,
                       ACT_NONE, nullptr, nullptr, 0, nullptr, ACT_NONE, sp.frame_stride));
    } else {
        float* up = c.arena.alloc<float>(rows25 * 2 * d);
        QA_TRY(linear_op(c, cat, rows25, h->up, up));
        if (!c.dry) QA_TRY(launch_dwconv(up, h->up_dw, h->up_dwb, nullptr, nullptr, x, B, N50, d, 5, 0.f, c.stream, zpad_left(5, cz)));
    }
    c.tap("dec.embed", x, rows * d);
    QA_TRY(dec_resblock_op(c, h->dres[0], x, B, N50, d, sp.gn_groups, cz));
    QA_TRY(dec_resblock_op(c, h->dres[1], x, B, N50, d, sp.gn_groups, cz));
    c.tap("dec.prior_res1", x, rows * d);
    QA_TRY(transformer_op(c, h->dec_tr, x, B, N50, "decoder.prior_net.3", cz));
    c.tap("dec.transformer", x, rows * d);
    QA_TRY(dec_resblock_op(c, h->dres[2], x, B, N50, d, sp.gn_groups, cz));
    QA_TRY(dec_resblock_op(c, h->dres[3], x, B, N50, d, sp.gn_groups, cz));
    float* t1 = c.arena.alloc<float>(rows * d);
    float* u = c.arena.alloc<float>(rows * sp.dec_inter);
    QA_TRY(groupnorm_op(c, x, h->gn_w, h->gn_b, t1, B, N50, d, sp.gn_groups, 0));
    if (!c.dry) QA_TRY(launch_layernorm(t1, h->norm_w, h->norm_b, x, rows, d, 1e-6f, c.stream));
    c.tap("dec.prior", x, rows * d);
    for (const ConvNeXtW& w : h->cnx) QA_TRY(convnext_op(c, w, x, t1, u, B, N50, d, cz));
    if (!c.dry) QA_TRY(launch_layernorm(x, h->fnorm_w, h->fnorm_b, t1, rows, d, 1e-6f, c.stream));
    c.tap("dec.backbone", t1, rows * d);
    // ISTFT head
    const int nb = sp.n_fft / 2 + 1;
    float* y = c.arena.alloc<float>(rows * 2 * nb);
    float* S = c.arena.alloc<float>(rows * h->spec_ld);
    float* frames = c.arena.alloc<float>(rows * sp.n_fft);
    QA_TRY(linear_op(c, t1, rows, h->head, y));
    if (!c.dry) QA_TRY(launch_istft_spec(y, S, rows, nb, 2 * nb, h->spec_ld, c.stream));
    c.tap("dec.spec", S, rows * h->spec_ld);
    QA_TRY(linear_op(c, S, rows, h->basis, frames));
    if (!c.dry) QA_TRY(launch_istft_ola(frames, h->window, wav_out, B, N50, sp.n_fft, sp.hop, c.stream));
    return QA_OK;
}

// ---- H-Codec 1.5 (codec_adaptive.py:150-199)

int read_scalar(Ctx& c, qa_hcodec* h, const int* dev, int* out) {
    QA_HIP(hipMemcpyAsync(h->host_sync, dev, sizeof(int), hipMemcpyDeviceToHost, c.stream));
    QA_HIP(hipStreamSynchronize(c.stream));  // data-dependent shape: the reference syncs here too (modeling_flexicodec_new.py:910)
    lstm_call_note_sync();                   // ... which also puts every recurrence launched so far behind a host synchronisation
    *out = *h->host_sync;
    return QA_OK;
}

int encode_adaptive_graph(qa_hcodec* h, Ctx& c, const float* wav, int B, int T, const float* feat, int64_t fsb, int64_t fsc,
                          int64_t fst, int n_feat, long long* ac_out, long long* sc_out, int* G_out, float threshold) {
    const qa_hcodec_spec& sp = h->spec;
    const int D = sp.code_dim, Q = sp.num_quantizers;
    float *emb = nullptr, *sem = nullptr;
    int N = 0;
    QA_TRY(encode_front(h, c, wav, B, T, feat, fsb, fsc, fst, n_feat, &emb, &sem, &N));
    int* seg = c.arena.alloc<int>((size_t)B * N);
    int* start = c.arena.alloc<int>((size_t)B * N);
    int* len = c.arena.alloc<int>((size_t)B * N);
    int* nseg = c.arena.alloc<int>(B);
    int* gmax = c.arena.alloc<int>(1);
    int G = N;  // planning pass: worst case, every frame its own group
    if (!c.dry) {
        QA_TRY(launch_align(sem, B, N, D, threshold, sp.max_tokens_per_group, seg, start, len, nseg, gmax, c.stream));
        QA_TRY(read_scalar(c, h, gmax, &G));
        QA_REQUIRE(G >= 1 && G <= N, "encode: alignment produced %d groups for %d frames", G, N);
    }
    *G_out = G;
    const int S = N + G;
    float* inter_s = c.arena.alloc<float>((size_t)B * S * D);
    float* inter_a = c.arena.alloc<float>((size_t)B * S * D);
    float* agg_a = c.arena.alloc<float>((size_t)B * G * D);
    float* agg_s = c.arena.alloc<float>((size_t)B * G * D);
    // semantic_aggregator(sem), acoustic_aggregator(emb): both use the alignment of the semantic stream and are otherwise
    // independent, so the two 32-layer stacks run concurrently on two streams (fork / join with events; no host sync)
    hipStream_t side = serial_mode() ? c.stream : h->side;  // qa_set_serial(1): one stream
    if (!c.dry) {
        QA_TRY(launch_agg_build(sem, seg, start, len, nseg, h->qemb_sem, inter_s, B, N, G, D, c.stream));
        QA_TRY(launch_agg_build(emb, seg, start, len, nseg, h->qemb_ac, inter_a, B, N, G, D, c.stream));
        QA_HIP(hipEventRecord(h->ev_fork, c.stream));
        QA_HIP(hipStreamWaitEvent(side, h->ev_fork, 0));
    }
    QA_TRY(mimi_pair_op(c, side, h->agg_sem, inter_s, h->agg_ac, inter_a, B, S));
    if (!c.dry) {
        hipStream_t main = c.stream;
        QA_TRY(launch_agg_gather(inter_s, start, len, nseg, agg_s, B, N, G, D, main));
        QA_TRY(launch_agg_gather(inter_a, start, len, nseg, agg_a, B, N, G, D, side));
        QA_HIP(hipEventRecord(h->ev_join, side));
        QA_HIP(hipStreamWaitEvent(main, h->ev_join, 0));
    }
    c.tap("enc.emb_agg", agg_a, (int64_t)B * G * D);
    c.tap("enc.sem_agg", agg_s, (int64_t)B * G * D);
    long long* ia = c.arena.alloc<long long>((size_t)B * G * Q);
    long long* is = c.arena.alloc<long long>((size_t)B * G * Q);
    float* rvq_ws = c.arena.alloc<float>(rvq_scratch_floats((long long)B * G, sp.codebook_size, D));
    if (!c.dry) {
        QA_TRY(launch_rvq_search(agg_a, (long long)B * G, h->cb_a, h->e2_a, Q, sp.codebook_size, D, ia, nullptr, 0, rvq_ws, c.stream));
        QA_TRY(launch_rvq_search(agg_s, (long long)B * G, h->cb_s, h->e2_s, Q, sp.codebook_size, D, is, nullptr, 0, rvq_ws, c.stream));
        QA_TRY(launch_codes_inject(ia, len, ac_out, B, N, G, Q, sp.codebook_size, c.stream));
        QA_TRY(launch_codes_inject(is, len, sc_out, B, N, G, Q, sp.codebook_size, c.stream));
    }
    return QA_OK;
}

int decode_adaptive_graph(qa_hcodec* h, Ctx& c, const long long* ac, const long long* scodes, int B, int G, int N,
                          float* wav_out) {
    const qa_hcodec_spec& sp = h->spec;
    const int Q = sp.num_quantizers, D = sp.code_dim;
    const int64_t rows = (int64_t)B * N;
    long long* ia = c.arena.alloc<long long>(rows * Q);
    long long* is = c.arena.alloc<long long>(rows * Q);
    float* cat = c.arena.alloc<float>(rows * 2 * D);
    if (!c.dry) {
        // token lengths: the reference keeps the ones extracted from the SEMANTIC codes for both streams (codec_adaptive.py:185-186)
        QA_TRY(launch_deaggregate(ac, scodes, ia, B, Q, G, N, sp.codebook_size, c.stream));
        QA_TRY(launch_deaggregate(scodes, scodes, is, B, Q, G, N, sp.codebook_size, c.stream));
        QA_TRY(launch_rvq_lookup(ia, rows, h->cb_a, Q, sp.codebook_size, D, cat, 2 * D, c.stream));
        QA_TRY(launch_rvq_lookup(is, rows, h->cb_s, Q, sp.codebook_size, D, cat + D, 2 * D, c.stream));
    }
    QA_TRY(mimi_op(c, h->bottleneck, cat, B, N));
    c.tap("dec.bottleneck", cat, rows * 2 * D);
    return decode_tail(h, c, cat, B, N, wav_out);
}

int ensure_workspace(qa_hcodec* h, size_t bytes) {
    if (bytes <= h->ws_cap) return QA_OK;
    if (h->ws) QA_HIP(hipFree(h->ws));  // synchronises with outstanding work
    h->ws = nullptr;
    h->ws_cap = 0;
    const size_t cap = bytes + bytes / 8;
    QA_HIP(hipMalloc(reinterpret_cast<void**>(&h->ws), cap));
    h->ws_cap = cap;
    return QA_OK;
}

int build(qa_hcodec* h, const HostTable& tab) {
    const qa_hcodec_spec& sp = h->spec;
    const bool v20 = sp.version == 20;
    QA_REQUIRE(sp.version == 0 || sp.version == 10 || sp.version == 15 || v20, "spec: unknown version %d", sp.version);
    QA_REQUIRE(!(v20 && sp.adaptive), "spec: H-Codec 2.0 has no adaptive frame rate");
    QA_REQUIRE(sp.n_sem_strides >= 1 && sp.n_sem_strides <= 4, "spec: bad counts");
    QA_REQUIRE(v20 || (sp.n_ratios >= 1 && sp.n_ratios <= 8 && sp.n_filters % 32 == 0), "spec: bad SEANet ladder");
    QA_REQUIRE((v20 ? sp.enc_dim : sp.dimension) % 128 == 0 && sp.dec_dim % 128 == 0, "spec: transformer widths must be multiples of 128");
    auto tr_inter = [&](int d) { return sp.tr_inter_cap > 0 ? std::min(4 * d, sp.tr_inter_cap) : 4 * d; };
    QA_REQUIRE(sp.sem_in % 32 == 0 && sp.sem_ch % 32 == 0 && sp.code_dim % 32 == 0 && sp.dec_inter % 32 == 0,
               "spec: channel counts must be multiples of 32");
    QA_REQUIRE(sp.n_fft % 2 == 0 && sp.hop > 0 && sp.n_fft > sp.hop && (sp.n_fft - sp.hop) % 2 == 0, "spec: bad STFT geometry");
    QA_REQUIRE(sp.dec_dim % sp.gn_groups == 0, "spec: dec_dim %% gn_groups != 0");
    Builder b{Folder{tab, h->store}};
    auto dw_fold_c = [&](const float** dst, const std::string& name, int k, int ch) {
        std::vector<float> w((size_t)k * ch, 0.f);
        const float* p = b.f.need(name, (int64_t)ch * k);
        if (p)
            for (int c = 0; c < ch; ++c)
                for (int j = 0; j < k; ++j) w[(size_t)j * ch + c] = p[c * k + j];
        b.raw(dst, w);
    };
    auto fold_convnext = [&](ConvNeXtW& w, const std::string& cp, int ch, int inter) {
        dw_fold_c(&w.dw, cp + ".dwconv.conv.weight", 7, ch);
        b.vec(&w.dwb, cp + ".dwconv.conv.bias", ch);
        b.vec(&w.lnw, cp + ".norm.weight", ch);
        b.vec(&w.lnb, cp + ".norm.bias", ch);
        b.vec(&w.gamma, cp + ".gamma", ch);
        w.pw1.N = inter; w.pw1.C_in = ch; w.pw1.ksize = 1;
        b.vec(&w.pw1.w, cp + ".pwconv1.linear.weight", (int64_t)inter * ch);
        b.vec(&w.pw1.b, cp + ".pwconv1.linear.bias", inter);
        w.pw2.N = ch; w.pw2.C_in = inter; w.pw2.ksize = 1;
        b.vec(&w.pw2.w, cp + ".pwconv2.linear.weight", (int64_t)inter * ch);
        b.vec(&w.pw2.b, cp + ".pwconv2.linear.bias", ch);
    };
    if (v20) {
        // --- H-Codec 2.0 encoder (HCodec-2.0/vq/codec_encoder.py:12-79)
        const int Nf = sp.n_fft, nbins = Nf / 2 + 1, blk = (Nf - sp.hop) / 2, de = sp.enc_dim;
        QA_REQUIRE(sp.hop == 2 * blk && Nf == 4 * blk && blk % 32 == 0, "spec: H-Codec 2.0 needs n_fft = 2 * hop and hop %% 64 == 0");
        QA_REQUIRE(de % 64 == 0 && sp.enc_inter % 32 == 0 && sp.frame_stride >= 1, "spec: bad H-Codec 2.0 encoder widths");
        {   // torchaudio Spectrogram(n_fft, hop, center=False, power=None): X[k] = sum_n hann[n] x[n] e^{-2 pi i k n / N}
            std::vector<float> basis((size_t)2 * nbins * Nf);
            for (int k = 0; k < nbins; ++k)
                for (int n = 0; n < Nf; ++n) {
                    const double wn = 0.5 - 0.5 * std::cos(2.0 * M_PI * n / Nf);
                    const double ang = 2.0 * M_PI * (double)(((int64_t)k * n) % Nf) / Nf;
                    basis[(size_t)k * Nf + n] = (float)(wn * std::cos(ang));
                    // DC and Nyquist bins are purely real: keep their imaginary rows at exactly +0 so that angle() of a negative
                    // real value is +pi, as torch.stft's rfft returns it (a -0 / 1e-17 there flips the phase to -pi)
                    basis[(size_t)(nbins + k) * Nf + n] = (k == 0 || 2 * k == Nf) ? 0.f : (float)(-wn * std::sin(ang));
                }
            h->stft_basis.N = 2 * nbins; h->stft_basis.C_in = blk; h->stft_basis.ksize = 4;
            b.raw(&h->stft_basis.w, basis);  // [2*nb][4][blk] == [2*nb][n_fft]
        }
        h->stft_ld = pad32(2 * nbins);
        b.conv(&h->enc_embed, "encoder.embed.conv", de, 2 * nbins, 3, false, true, de, h->stft_ld);
        b.vec(&h->enc_norm_w, "encoder.norm.weight", de);
        b.vec(&h->enc_norm_b, "encoder.norm.bias", de);
        h->enc_cnx.resize(sp.enc_convnext_layers);
        for (int i = 0; i < sp.enc_convnext_layers; ++i) fold_convnext(h->enc_cnx[i], "encoder.prior_net." + std::to_string(i), de, sp.enc_inter);
        build_transformer(b, &h->enc_tr, "encoder.post_net.1", de, sp.enc_layers, de / 64, tr_inter(de));
        b.vec(&h->enc_fnorm_w, "encoder.final_layer_norm.weight", de);
        b.vec(&h->enc_fnorm_b, "encoder.final_layer_norm.bias", de);
        b.conv(&h->enc_out20, "encoder.out.conv", sp.code_dim, de, 2 * sp.frame_stride + 1, false, true);
    } else {
    // --- SEANet encoder (seanet.py:121-187)
    const std::string em = "encoder.model.";
    {
        std::vector<float> w0((size_t)7 * sp.n_filters), b0(sp.n_filters);
        const float* v = b.f.need(em + "0.conv.conv.weight_v", (int64_t)sp.n_filters * 7);
        const float* g = b.f.need(em + "0.conv.conv.weight_g", sp.n_filters);
        const float* bi = b.f.need(em + "0.conv.conv.bias", sp.n_filters);
        if (v && g && bi) {
            for (int n = 0; n < sp.n_filters; ++n) {
                double ss = 0;
                for (int j = 0; j < 7; ++j) ss += (double)v[n * 7 + j] * v[n * 7 + j];
                const float sc = g[n] / (float)std::sqrt(ss);
                for (int j = 0; j < 7; ++j) w0[(size_t)j * sp.n_filters + n] = v[n * 7 + j] * sc;
                b0[n] = bi[n];
            }
        }
        b.raw(&h->conv0_w, w0);
        b.raw(&h->conv0_b, b0);
    }
    h->res.resize(sp.n_ratios);
    h->down.resize(sp.n_ratios);
    int C = sp.n_filters;
    for (int i = 0; i < sp.n_ratios; ++i) {
        const std::string rp = em + std::to_string(1 + 3 * i);
        const int hid = C / 2, hp = pad32(hid);
        b.conv(&h->res[i].k3, rp + ".block.1.conv.conv", hid, C, 3, true, true, hp, C);
        b.conv(&h->res[i].pw, rp + ".block.3.conv.conv", C, hid, 1, true, true, C, hp);
        b.conv(&h->res[i].sc, rp + ".shortcut.conv.conv", C, C, 1, true, true);
        b.conv(&h->down[i], em + std::to_string(3 + 3 * i) + ".conv.conv", 2 * C, C, 2 * sp.ratios[i], true, true);
        C *= 2;
    }
    QA_REQUIRE(C == sp.dimension, "spec: n_filters * 2^n_ratios = %d != dimension %d", C, sp.dimension);
    build_transformer(b, &h->enc_tr, em + std::to_string(3 * sp.n_ratios + 2), sp.dimension, sp.enc_layers, sp.enc_heads);
    b.conv(&h->enc_out, em + std::to_string(3 * sp.n_ratios + 5) + ".conv.conv", sp.dimension, sp.dimension, 4, true, true);
    QA_REQUIRE(sp.code_dim == sp.dimension, "spec: code_dim must equal dimension");
    }
    // --- semantic encoder (semantic_module.py:157-201)
    const std::string se = "semantic_encoder.";
    b.conv(&h->sem_in, se + "conv.conv", sp.sem_ch, sp.sem_in, 3, false, false);
    h->sem_blocks.resize(sp.n_sem_strides);
    for (int i = 0; i < sp.n_sem_strides; ++i) {
        auto& blk = h->sem_blocks[i];
        const std::string bp = se + "conv_blocks." + std::to_string(i);
        for (int u = 0; u < 2; ++u) {
            b.conv(&blk.u1[u], bp + ".res_units." + std::to_string(u) + ".conv1.conv", sp.sem_ch, sp.sem_ch, 3, false, false);
            b.conv(&blk.u2[u], bp + ".res_units." + std::to_string(u) + ".conv2", sp.sem_ch, sp.sem_ch, 1, false, false);
        }
        blk.stride = sp.sem_strides[i];
        b.conv(&blk.conv, bp + ".conv.conv", sp.sem_ch, sp.sem_ch, blk.stride == 1 ? 3 : 2 * blk.stride, false, true);
    }
    b.conv(&h->sem_out, se + "conv2.conv", sp.code_dim, sp.sem_ch, 3, false, false);
    // --- codebooks (vector_quantize_pytorch layout: layers.{q}._codebook.embed [1, K, D])
    {
        const int64_t kd = (int64_t)sp.codebook_size * sp.code_dim;
        std::vector<float> cba((size_t)sp.num_quantizers * kd), cbs((size_t)sp.num_quantizers * kd);
        for (int q = 0; q < sp.num_quantizers; ++q) {
            const float* a = b.f.need("quantizer.layers." + std::to_string(q) + "._codebook.embed", kd);
            const float* s = b.f.need("semantic_quantizer.layers." + std::to_string(q) + "._codebook.embed", kd);
            if (a) std::memcpy(&cba[(size_t)q * kd], a, sizeof(float) * kd);
            if (s) std::memcpy(&cbs[(size_t)q * kd], s, sizeof(float) * kd);
        }
        b.raw(&h->cb_a, cba);
        b.raw(&h->cb_s, cbs);
    }
    // --- decoder (codec_decoder.py:14-67)
    const int d = sp.dec_dim;
    if (v20) {
        b.conv(&h->dec_embed20, "decoder.embed.conv", d, 2 * sp.code_dim, sp.frame_stride + 1, false, true);
    } else {
        b.conv(&h->up, "decoder.embed.up", 2 * d, 2 * sp.code_dim, 1, false, true);
        dw_fold_c(&h->up_dw, "decoder.embed.dw.weight", 5, d);
        b.vec(&h->up_dwb, "decoder.embed.dw.bias", d);
    }
    const int ridx[4] = {0, 1, 5, 6};
    for (int i = 0; i < 4; ++i) {
        const std::string rp = "decoder.prior_net." + std::to_string(ridx[i]);
        b.vec(&h->dres[i].n1w, rp + ".norm1.weight", d);
        b.vec(&h->dres[i].n1b, rp + ".norm1.bias", d);
        b.vec(&h->dres[i].n2w, rp + ".norm2.weight", d);
        b.vec(&h->dres[i].n2b, rp + ".norm2.bias", d);
        b.conv(&h->dres[i].c1, rp + ".conv1.conv", d, d, 3, false, true);
        b.conv(&h->dres[i].c2, rp + ".conv2.conv", d, d, 3, false, true);
    }
    build_transformer(b, &h->dec_tr, "decoder.prior_net.3", d, sp.dec_layers, sp.dec_heads, tr_inter(d));
    b.vec(&h->gn_w, "decoder.prior_net.7.weight", d);
    b.vec(&h->gn_b, "decoder.prior_net.7.bias", d);
    b.vec(&h->norm_w, "decoder.norm.weight", d);
    b.vec(&h->norm_b, "decoder.norm.bias", d);
    b.vec(&h->fnorm_w, "decoder.final_layer_norm.weight", d);
    b.vec(&h->fnorm_b, "decoder.final_layer_norm.bias", d);
    h->cnx.resize(sp.convnext_layers);
    for (int i = 0; i < sp.convnext_layers; ++i) fold_convnext(h->cnx[i], "decoder.post_net." + std::to_string(i), d, sp.dec_inter);
    const int nb = sp.n_fft / 2 + 1;
    h->head.N = 2 * nb; h->head.C_in = d; h->head.ksize = 1;
    b.vec(&h->head.w, "decoder.head.out.weight", (int64_t)2 * nb * d);
    b.vec(&h->head.b, "decoder.head.out.bias", 2 * nb);
    // inverse real DFT (norm="backward") with the synthesis window folded in (spectral_ops.py:55-56):
    //   frame[n] = w[n]/N * sum_k c_k (Re_k cos(2 pi k n / N) - Im_k sin(2 pi k n / N)),  c_0 = c_{N/2} = 1, else 2
    {
        const int Nf = sp.n_fft;
        h->spec_ld = pad32(2 * nb);
        std::vector<float> win(Nf);
        if (tab.has("decoder.head.istft.window")) {
            const float* wp = b.f.need("decoder.head.istft.window", Nf);
            if (wp) std::memcpy(win.data(), wp, sizeof(float) * Nf);
        } else {
            for (int n = 0; n < Nf; ++n) win[n] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * n / Nf));
        }
        std::vector<float> basis((size_t)Nf * h->spec_ld, 0.f);
        for (int n = 0; n < Nf; ++n)
            for (int k = 0; k < nb; ++k) {
                const double ck = (k == 0 || k == Nf / 2) ? 1.0 : 2.0;
                const double ang = 2.0 * M_PI * (double)(((int64_t)k * n) % Nf) / Nf;
                basis[(size_t)n * h->spec_ld + k] = (float)(win[n] * ck * std::cos(ang) / Nf);
                basis[(size_t)n * h->spec_ld + nb + k] = (k == 0 || k == Nf / 2) ? 0.f : (float)(-win[n] * ck * std::sin(ang) / Nf);
            }
        h->basis.N = Nf; h->basis.C_in = h->spec_ld; h->basis.ksize = 1;
        b.raw(&h->basis.w, basis);
        b.raw(&h->window, win);
    }
    if (sp.adaptive) {
        QA_REQUIRE(sp.agg_heads > 0 && sp.bt_heads > 0 && sp.code_dim % sp.agg_heads == 0 && (2 * sp.code_dim) % sp.bt_heads == 0,
                   "spec: bad head counts for the adaptive stacks");
        QA_REQUIRE(sp.agg_ff % 32 == 0 && sp.bt_ff % 32 == 0 && sp.max_tokens_per_group >= 1, "spec: bad adaptive widths");
        QA_REQUIRE(sp.agg_context >= 0 && sp.bt_context >= 0, "spec: negative attention context");
        build_mimi(b, &h->agg_sem, "semantic_aggregator.transformer.transformer", sp.code_dim, sp.agg_layers, sp.agg_heads, sp.agg_ff,
                   sp.agg_causal, sp.agg_context);
        build_mimi(b, &h->agg_ac, "acoustic_aggregator.transformer.transformer", sp.code_dim, sp.agg_layers, sp.agg_heads, sp.agg_ff,
                   sp.agg_causal, sp.agg_context);
        build_mimi(b, &h->bottleneck, "bottleneck_transformer.transformer", 2 * sp.code_dim, sp.bt_layers, sp.bt_heads, sp.bt_ff,
                   sp.bt_causal, sp.bt_context);
        b.vec(&h->qemb_sem, "semantic_aggregator.query_embedding", sp.code_dim);
        b.vec(&h->qemb_ac, "acoustic_aggregator.query_embedding", sp.code_dim);
    }
    if (!b.f.ok) return b.f.status;
    QA_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->host_sync), sizeof(int) * 4));
    QA_HIP(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
    QA_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    QA_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    QA_TRY(h->store.upload());
    b.resolve();
    // |e|^2 tables
    const int QK = sp.num_quantizers * sp.codebook_size;
    QA_HIP(hipMalloc(reinterpret_cast<void**>(&h->e2_dev), sizeof(float) * 2 * QK));
    QA_TRY(launch_rvq_norms(h->cb_a, h->e2_dev, QK, sp.code_dim, nullptr));
    QA_TRY(launch_rvq_norms(h->cb_s, h->e2_dev + QK, QK, sp.code_dim, nullptr));
    QA_HIP(hipDeviceSynchronize());
    h->e2_a = h->e2_dev;
    h->e2_s = h->e2_dev + QK;
    return QA_OK;
}

}  // namespace
}  // namespace qa

// The non-dry pass of a model graph.  A call that launched the persistent LSTM recurrence (lstm.hip) waits for its stream before
// returning and, should one of that kernel's grid barriers have timed out (it needs every workgroup resident at once: the device
// was shared with another such kernel), runs the graph again on the per-step kernels - the call that hit the failure returns
// valid results, and the device stops choosing the persistent kernel by itself.
template <typename F>
static int run_graph_checked(qa_hcodec* h, Ctx& c, F&& graph) {
    void* ticket = nullptr;
    QA_TRY(lstm_call_begin(h->device, &ticket));
    {
        const int st = graph();
        if (st != QA_OK) {  // error path only: the aggregator side stream may still run out of the workspace
            (void)hipDeviceSynchronize();
            bool ignored = false;
            (void)lstm_call_end(ticket, c.stream, &ignored);
            return st;
        }
    }
    bool failed = false;
    QA_TRY(lstm_call_end(ticket, c.stream, &failed));  // host-synchronous only for a call that launched an in-launch recurrence
    if (!failed) return QA_OK;
    std::fprintf(stderr, "libquarkaudio_hip: the grid barrier of the persistent LSTM recurrence timed out on device %d (shared device?); "
                         "re-running the call on the per-step kernels\n", h->device);
    lstm_force_per_step(true);
    c.taps.clear();
    c.arena.begin(h->ws, h->ws_cap);
    const int st = graph();
    lstm_force_per_step(false);
    return st;
}

extern "C" {

int qa_hcodec_create(qa_hcodec** out, const qa_hcodec_spec* spec, const qa_tensor* tensors, int64_t n_tensors, int device) {
    if (!out || !spec || !tensors) {
        set_error("qa_hcodec_create: null argument");
        return QA_ERR_INVALID;
    }
    *out = nullptr;
    QA_HIP(hipSetDevice(device));
    std::unique_ptr<qa_hcodec> h(new qa_hcodec());
    h->spec = *spec;
    h->device = device;
    HostTable tab(tensors, n_tensors);
    const int st = build(h.get(), tab);
    if (st != QA_OK) {
        h->store.release();
        if (h->e2_dev) (void)hipFree(h->e2_dev);
        return st;
    }
    *out = h.release();
    return QA_OK;
}

void qa_hcodec_destroy(qa_hcodec* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    h->store.release();
    if (h->e2_dev) (void)hipFree(h->e2_dev);
    if (h->ws) (void)hipFree(h->ws);
    if (h->host_sync) (void)hipHostFree(h->host_sync);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    delete h;
}

int qa_hcodec_encode(qa_hcodec* h, const float* wav, int64_t B, int64_t T, const float* feat, int64_t fsb, int64_t fsc,
                     int64_t fst, int64_t n_feat, int64_t* ac, int64_t* sc, void* stream) {
    if (!h || !wav || !feat || !ac || !sc) {
        set_error("qa_hcodec_encode: null argument");
        return QA_ERR_INVALID;
    }
    int hop = 2;
    for (int i = 0; i < h->spec.n_ratios; ++i) hop *= h->spec.ratios[i];
    if (h->spec.version == 20) hop = h->spec.hop * h->spec.frame_stride;
    QA_REQUIRE(B > 0 && T > 0 && T % hop == 0, "qa_hcodec_encode: wav is [%lld, %lld]; T must be a positive multiple of %d "
               "(HCodecTokenizer.pad_wav)", (long long)B, (long long)T, hop);
    QA_REQUIRE(B * T < (1LL << 31), "qa_hcodec_encode: batch of %lld x %lld samples is too large", (long long)B, (long long)T);
    QA_REQUIRE(!h->spec.adaptive, "qa_hcodec_encode: this handle is an H-Codec 1.5 model, use qa_hcodec_encode_adaptive");
    QA_HIP(hipSetDevice(h->device));
    Ctx& c = h->ctx;
    c.stream = static_cast<hipStream_t>(stream);
    c.dry = true;
    c.arena.begin(nullptr, 0);
    QA_TRY(encode_graph(h, c, wav, (int)B, (int)T, feat, fsb, fsc, fst, (int)n_feat, (long long*)ac, (long long*)sc));
    QA_TRY(ensure_workspace(h, c.arena.peak()));
    c.dry = false;
    c.taps.clear();
    c.arena.begin(h->ws, h->ws_cap);
    return run_graph_checked(h, c, [&] { return encode_graph(h, c, wav, (int)B, (int)T, feat, fsb, fsc, fst, (int)n_feat, (long long*)ac, (long long*)sc); });
}

int qa_hcodec_decode(qa_hcodec* h, const int64_t* ac, const int64_t* sc, int64_t B, int64_t N, float* wav_out, void* stream) {
    if (!h || !ac || !sc || !wav_out) {
        set_error("qa_hcodec_decode: null argument");
        return QA_ERR_INVALID;
    }
    QA_REQUIRE(B > 0 && N > 0, "qa_hcodec_decode: codes are [%lld, Q, %lld]", (long long)B, (long long)N);
    QA_REQUIRE(B * N * (h->spec.version == 20 ? h->spec.frame_stride : 2) * (int64_t)h->spec.hop < (1LL << 31), "qa_hcodec_decode: output too large");
    QA_REQUIRE(!h->spec.adaptive, "qa_hcodec_decode: this handle is an H-Codec 1.5 model, use qa_hcodec_decode_adaptive");
    QA_HIP(hipSetDevice(h->device));
    Ctx& c = h->ctx;
    c.stream = static_cast<hipStream_t>(stream);
    c.dry = true;
    c.arena.begin(nullptr, 0);
    QA_TRY(decode_graph(h, c, (const long long*)ac, (const long long*)sc, (int)B, (int)N, wav_out));
    QA_TRY(ensure_workspace(h, c.arena.peak()));
    c.dry = false;
    c.taps.clear();
    c.arena.begin(h->ws, h->ws_cap);
    return run_graph_checked(h, c, [&] { return decode_graph(h, c, (const long long*)ac, (const long long*)sc, (int)B, (int)N, wav_out); });
}

int qa_hcodec_encode_adaptive(qa_hcodec* h, const float* wav, int64_t B, int64_t T, const float* feat, int64_t fsb, int64_t fsc,
                              int64_t fst, int64_t n_feat, int64_t* ac, int64_t* sc, int64_t* n_groups, float threshold, void* stream) {
    if (!h || !wav || !feat || !ac || !sc || !n_groups) {
        set_error("qa_hcodec_encode_adaptive: null argument");
        return QA_ERR_INVALID;
    }
    QA_REQUIRE(h->spec.adaptive, "qa_hcodec_encode_adaptive: this handle is not an H-Codec 1.5 model");
    int hop = 2;
    for (int i = 0; i < h->spec.n_ratios; ++i) hop *= h->spec.ratios[i];
    QA_REQUIRE(B > 0 && T > 0 && T % hop == 0, "qa_hcodec_encode_adaptive: wav is [%lld, %lld]; T must be a positive multiple of %d",
               (long long)B, (long long)T, hop);
    QA_REQUIRE(B * T < (1LL << 31), "qa_hcodec_encode_adaptive: batch too large");
    QA_REQUIRE(threshold >= 0.f && threshold <= 1.f, "qa_hcodec_encode_adaptive: threshold %g outside [0, 1] (codec_adaptive.py:151)", threshold);
    const float thr = threshold <= 0.f ? h->spec.threshold : threshold;  // codec_adaptive.py:158
    QA_HIP(hipSetDevice(h->device));
    Ctx& c = h->ctx;
    c.stream = static_cast<hipStream_t>(stream);
    c.dry = true;
    c.arena.begin(nullptr, 0);
    int G = 0;
    QA_TRY(encode_adaptive_graph(h, c, wav, (int)B, (int)T, feat, fsb, fsc, fst, (int)n_feat, (long long*)ac, (long long*)sc, &G, thr));
    QA_TRY(ensure_workspace(h, c.arena.peak()));
    c.dry = false;
    c.taps.clear();
    c.arena.begin(h->ws, h->ws_cap);
    QA_TRY(run_graph_checked(h, c, [&] {
        return encode_adaptive_graph(h, c, wav, (int)B, (int)T, feat, fsb, fsc, fst, (int)n_feat, (long long*)ac, (long long*)sc, &G, thr);
    }));
    *n_groups = G;
    return QA_OK;
}

int qa_hcodec_adaptive_frames(qa_hcodec* h, const int64_t* semantic_codes, int64_t B, int64_t G, int64_t* frames, void* stream) {
    if (!h || !semantic_codes || !frames) {
        set_error("qa_hcodec_adaptive_frames: null argument");
        return QA_ERR_INVALID;
    }
    QA_REQUIRE(h->spec.adaptive && B > 0 && G > 0, "qa_hcodec_adaptive_frames: bad argument");
    QA_HIP(hipSetDevice(h->device));
    QA_TRY(ensure_workspace(h, (size_t)(B + 64) * sizeof(int)));
    Ctx& c = h->ctx;
    c.stream = static_cast<hipStream_t>(stream);
    int* totals = reinterpret_cast<int*>(h->ws);
    int* tmax = totals + B;
    QA_TRY(launch_adaptive_frames((const long long*)semantic_codes, (int)B, h->spec.num_quantizers, (int)G, h->spec.codebook_size,
                                  totals, tmax, c.stream));
    int n = 0;
    QA_TRY(read_scalar(c, h, tmax, &n));
    *frames = n;
    return QA_OK;
}

int qa_hcodec_decode_adaptive(qa_hcodec* h, const int64_t* ac, const int64_t* sc, int64_t B, int64_t G, int64_t frames,
                              float* wav_out, void* stream) {
    if (!h || !ac || !sc || !wav_out) {
        set_error("qa_hcodec_decode_adaptive: null argument");
        return QA_ERR_INVALID;
    }
    QA_REQUIRE(h->spec.adaptive, "qa_hcodec_decode_adaptive: this handle is not an H-Codec 1.5 model");
    QA_REQUIRE(B > 0 && G > 0 && frames > 0 && B * frames * 2 * (int64_t)h->spec.hop < (1LL << 31), "qa_hcodec_decode_adaptive: bad shape");
    QA_HIP(hipSetDevice(h->device));
    Ctx& c = h->ctx;
    c.stream = static_cast<hipStream_t>(stream);
    c.dry = true;
    c.arena.begin(nullptr, 0);
    QA_TRY(decode_adaptive_graph(h, c, (const long long*)ac, (const long long*)sc, (int)B, (int)G, (int)frames, wav_out));
    QA_TRY(ensure_workspace(h, c.arena.peak()));
    c.dry = false;
    c.taps.clear();
    c.arena.begin(h->ws, h->ws_cap);
    return run_graph_checked(h, c, [&] { return decode_adaptive_graph(h, c, (const long long*)ac, (const long long*)sc, (int)B, (int)G, (int)frames, wav_out); });
}

int qa_hcodec_enable_taps(qa_hcodec* h, int on) {
    if (!h) {
        set_error("qa_hcodec_enable_taps: null handle");
        return QA_ERR_INVALID;
    }
    h->ctx.capture = on != 0;
    return QA_OK;
}

int64_t qa_hcodec_tap(qa_hcodec* h, const char* name, float* dst, int64_t cap, void* stream) {
    if (!h || !name) {
        set_error("qa_hcodec_tap: null argument");
        return QA_ERR_INVALID;
    }
    auto it = h->ctx.taps.find(name);
    if (it == h->ctx.taps.end()) {
        set_error("qa_hcodec_tap: no intermediate named '%s' in the last call", name);
        return QA_ERR_MISSING;
    }
    if (dst) {
        if (cap < it->second.numel) {
            set_error("qa_hcodec_tap: '%s' has %lld elements, capacity %lld", name, (long long)it->second.numel, (long long)cap);
            return QA_ERR_INVALID;
        }
        QA_HIP(hipMemcpyAsync(dst, it->second.ptr, sizeof(float) * it->second.numel, hipMemcpyDeviceToDevice,
                              static_cast<hipStream_t>(stream)));
    }
    return it->second.numel;
}


/* ---- mimi StreamingTransformer ----------------------------------------------------------------------------------------- */

int qa_mimi_create(qa_mimi** out, const qa_mimi_spec* spec, const qa_tensor* tensors, int64_t n_tensors, const char* prefix,
                   int device) {
    if (!out || !spec || !tensors || !prefix) {
        set_error("qa_mimi_create: null argument");
        return QA_ERR_INVALID;
    }
    *out = nullptr;
    const qa_mimi_spec& sp = *spec;
    QA_REQUIRE(sp.d_model > 0 && sp.num_heads > 0 && sp.d_model % sp.num_heads == 0 && sp.num_layers > 0 && sp.dim_feedforward > 0,
               "qa_mimi_create: bad sizes");
    const int hd = sp.d_model / sp.num_heads;
    QA_REQUIRE(hd == 32 || hd == 64 || hd == 96 || hd == 128, "qa_mimi_create: head_dim %d unsupported (32/64/96/128)", hd);
    QA_REQUIRE(sp.d_model % 32 == 0 && sp.dim_feedforward % 32 == 0, "qa_mimi_create: widths must be multiples of 32");
    QA_REQUIRE(sp.context >= 0 && sp.context <= MAX_POS, "qa_mimi_create: context %d outside [0, %d]", sp.context, MAX_POS);
    QA_HIP(hipSetDevice(device));
    std::unique_ptr<qa_mimi> m(new qa_mimi());
    m->spec = sp;
    m->device = device;
    HostTable tab(tensors, n_tensors);
    Builder b{Folder{tab, m->store}};
    build_mimi(b, &m->w, prefix, sp.d_model, sp.num_layers, sp.num_heads, sp.dim_feedforward, sp.causal, sp.context);
    if (!b.f.ok) return b.f.status;
    const int st = m->store.upload();
    if (st != QA_OK) {
        m->store.release();
        return st;
    }
    b.resolve();
    *out = m.release();
    return QA_OK;
}

void qa_mimi_destroy(qa_mimi* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    (void)hipDeviceSynchronize();
    m->store.release();
    if (m->ring) (void)hipFree(m->ring);
    if (m->st.rope_win) (void)hipFree(m->st.rope_win);
    if (m->ws) (void)hipFree(m->ws);
    delete m;
}

static int mimi_run(qa_mimi* m, const float* x, int B, int T, float* y, hipStream_t stream, bool streaming) {
    Ctx& c = m->ctx;
    c.stream = stream;
    const int64_t rows = (int64_t)B * T;
    for (int pass = 0; pass < 2; ++pass) {
        c.dry = pass == 0;
        c.arena.begin(c.dry ? nullptr : m->ws, c.dry ? 0 : m->ws_cap);
        const MimiTemps t = mimi_temps(c, m->w, rows);
        if (c.dry) {
            if (c.arena.peak() > m->ws_cap) {
                if (m->ws) QA_HIP(hipFree(m->ws));
                m->ws = nullptr;
                m->ws_cap = 0;
                QA_HIP(hipMalloc(reinterpret_cast<void**>(&m->ws), c.arena.peak() + c.arena.peak() / 8));
                m->ws_cap = c.arena.peak() + c.arena.peak() / 8;
            }
            continue;
        }
        if (y != x) QA_HIP(hipMemcpyAsync(y, x, sizeof(float) * rows * m->w.d, hipMemcpyDeviceToDevice, stream));
        for (size_t l = 0; l < m->w.layers.size(); ++l)
            QA_TRY(mimi_layer(c, m->w, m->w.layers[l], y, t, B, T, streaming ? &m->st : nullptr, l));
    }
    return QA_OK;
}

int qa_mimi_forward(qa_mimi* m, const float* x, int64_t B, int64_t T, float* y, void* stream) {
    if (!m || !x || !y) {
        set_error("qa_mimi_forward: null argument");
        return QA_ERR_INVALID;
    }
    QA_REQUIRE(B > 0 && T > 0 && T <= MAX_POS && B * T < (1LL << 31), "qa_mimi_forward: x is [%lld, %lld, d] (T <= %d)", (long long)B,
               (long long)T, MAX_POS);
    QA_REQUIRE(m->st.B == 0, "qa_mimi_forward: the handle is in streaming mode, use qa_mimi_stream_step");
    QA_HIP(hipSetDevice(m->device));
    return mimi_run(m, x, (int)B, (int)T, y, static_cast<hipStream_t>(stream), false);
}

int qa_mimi_stream_begin(qa_mimi* m, int64_t B) {
    if (!m) {
        set_error("qa_mimi_stream_begin: null handle");
        return QA_ERR_INVALID;
    }
    QA_REQUIRE(m->spec.causal, "qa_mimi_stream_begin: Streaming only available for causal (mimi/transformer.py:382)");
    QA_REQUIRE(m->spec.context > 0, "qa_mimi_stream_begin: Cannot create a streaming KVCache without a context to estimate capacity "
               "(mimi/transformer.py:349-353)");
    QA_REQUIRE(B > 0 && B < (1 << 20), "qa_mimi_stream_begin: batch %lld", (long long)B);
    QA_HIP(hipSetDevice(m->device));
    if (m->ring) QA_HIP(hipFree(m->ring));
    m->ring = nullptr;
    const size_t L = m->w.layers.size(), per = (size_t)B * m->spec.context * m->w.d;
    QA_HIP(hipMalloc(reinterpret_cast<void**>(&m->ring), sizeof(float) * 2 * L * per));
    QA_HIP(hipMemset(m->ring, 0, sizeof(float) * 2 * L * per));  // RingKVCache.__init__: zeros
    m->st.kc.resize(L);
    m->st.vc.resize(L);
    for (size_t l = 0; l < L; ++l) {
        m->st.kc[l] = m->ring + (2 * l) * per;
        m->st.vc[l] = m->ring + (2 * l + 1) * per;
    }
    m->st.B = (int)B;
    m->st.cap = m->spec.context;
    m->st.offset = 0;
    m->st.rope_len = 0;
    return QA_OK;
}

int qa_mimi_stream_step(qa_mimi* m, const float* x, int64_t T, float* y, void* stream) {
    if (!m || !x || !y) {
        set_error("qa_mimi_stream_step: null argument");
        return QA_ERR_INVALID;
    }
    QA_REQUIRE(m->st.B > 0, "qa_mimi_stream_step: not streaming (call qa_mimi_stream_begin)");
    QA_REQUIRE(T >= 1 && T <= m->st.cap, "qa_mimi_stream_step: a chunk of %lld frames does not fit the ring of %d (RingKVCache.complete "
               "would write one slot twice)", (long long)T, m->st.cap);
    QA_REQUIRE((int64_t)m->st.offset + T < (1LL << 31), "qa_mimi_stream_step: offset %d + %lld overflows", m->st.offset, (long long)T);
    QA_HIP(hipSetDevice(m->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    // positions beyond the static table (QA_MIMI_ROPE_WINDOW shrinks it, for tests): a rolling window of the same table
    MimiStream& st = m->st;
    const int table = (int)std::min<long long>(MAX_POS, std::max<long long>(st.cap, knob(K_MIMI_ROPE_WINDOW)));
    const long long end = (long long)st.offset + T;
    if (end <= table) {
        st.rope_len = 0;
    } else if (st.rope_len == 0 || st.offset < st.rope_base || end > (long long)st.rope_base + st.rope_len) {
        const int hd = m->w.d / m->w.heads;
        if (!st.rope_win) QA_HIP(hipMalloc(reinterpret_cast<void**>(&st.rope_win), sizeof(float) * (size_t)MAX_POS * hd));
        std::vector<float> cs;
        mimi_rope_table(hd, st.offset, table, &cs);
        QA_HIP(hipStreamSynchronize(s));  // earlier steps may still read the old window; `cs` is a stack-lifetime host buffer
        QA_HIP(hipMemcpy(st.rope_win, cs.data(), sizeof(float) * cs.size(), hipMemcpyHostToDevice));
        st.rope_base = st.offset;
        st.rope_len = table;
    }
    QA_TRY(mimi_run(m, x, m->st.B, (int)T, y, s, true));
    m->st.offset += (int)T;
    return QA_OK;
}

int qa_mimi_stream_reset(qa_mimi* m) {
    if (!m) {
        set_error("qa_mimi_stream_reset: null handle");
        return QA_ERR_INVALID;
    }
    QA_REQUIRE(m->st.B > 0, "qa_mimi_stream_reset: Trying to reset streaming, but the transformer wasn't streaming (streaming.py:118-121)");
    m->st.offset = 0;  // RingKVCache.reset(): the caches keep their contents, end_offset = 0 alone invalidates them
    m->st.rope_len = 0;
    return QA_OK;
}

int qa_mimi_stream_end(qa_mimi* m) {
    if (!m) {
        set_error("qa_mimi_stream_end: null handle");
        return QA_ERR_INVALID;
    }
    (void)hipSetDevice(m->device);
    if (m->ring) {
        (void)hipDeviceSynchronize();
        (void)hipFree(m->ring);
    }
    m->ring = nullptr;
    if (m->st.rope_win) (void)hipFree(m->st.rope_win);
    m->st = MimiStream{};
    return QA_OK;
}

int64_t qa_mimi_stream_offset(const qa_mimi* m) { return (m && m->st.B > 0) ? m->st.offset : -1; }

}  // extern "C"
