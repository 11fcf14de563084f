// ssl.cpp - SSL front-end graph behind qa_ssl_*: HuBERT / wav2vec 2.0 feature extraction as HCodecTokenizer uses it
// (QuarkAudio-HCodec/HCodec-1.0/audio_tokenizer.py:35-48; HCodec-1.5/audio_tokenizer.py:53-67).  SURVEY.md 8f-1.
//
// Layout: channel-last [B, frames, C] throughout, so every Conv1d of the feature extractor (k3/k2, stride 2, no padding),
// every Linear and the grouped positional convolution (k128, one implicit GEMM per group over a 48/64-channel slice of the
// same buffer: ldx = ldy = hidden) is a conv_gemm launch with bias / GELU / residual fused; only layer 0 (C_in = 1) and the
// per-channel GroupNorm over time have their own kernel (ssl_kernels.hip).  Attention = the codec's flash kernel over a
// fused QKV buffer (q, k, v projections concatenated at load time).
#include <cmath>
#include <memory>

#include "host_util.h"

namespace qa {
size_t ssl_conv0_scratch_bytes(int B, int T1, int C0);
int launch_ssl_conv0(const float* wav, const float* w_kc, const float* bias, const float* gamma, const float* beta, float* y,
                     void* scratch, int B, int T, int T1, int C0, int ksize, int stride, int pad, int norm_group, float eps, int act,
                     hipStream_t s);
int launch_ssl_gate(const float* hidden, const float* wab, const float* bab, const float* cst, float* gate, int B, int N, int H, int hd,
                    hipStream_t s);
int launch_ssl_accumulate(float* dst, const float* src, long long n, int first, hipStream_t s);
int launch_ssl_act(float* x, long long n, int act, hipStream_t s);
int launch_ssl_compress(const float* sum, float* out, long long n, float scale, float expo, hipStream_t s);
}  // namespace qa

using namespace qa;

namespace {
struct SslLayer {
    ConvW qkv, o, ff1, ff2;
    const float *ln1w = nullptr, *ln1b = nullptr, *ln2w = nullptr, *ln2b = nullptr;
    const float *gate_w = nullptr, *gate_b = nullptr, *gate_c = nullptr;  // WavLM: folded gru_rel_pos_linear [2][hd], [2]; const [H]
};
}  // namespace

struct qa_ssl {
    qa_ssl_spec spec{};
    int device = 0;
    WeightStore store;
    // feature extractor
    const float *conv0_w = nullptr, *conv0_b = nullptr, *gn_w = nullptr, *gn_b = nullptr;
    std::vector<ConvW> convs;                      // layers 1..n_conv-1
    std::vector<const float*> cln_w, cln_b;        // "layer" flavour: LayerNorm after every conv (index = layer)
    const float *fp_ln_w = nullptr, *fp_ln_b = nullptr;
    ConvW fp;
    std::vector<ConvW> pos;                        // one per group
    const float *enc_ln_w = nullptr, *enc_ln_b = nullptr;
    const float* relbias = nullptr;  // WavLM: [H][2R+1] relative position bias by clamped distance, R = rel_pos_max_distance
    std::vector<SslLayer> layers;
    std::vector<int> select;
    char* ws = nullptr;
    size_t ws_cap = 0;
    Ctx ctx;
};

namespace {

int conv(Ctx& c, const float* x, int64_t ldx, int B, int T_in, const ConvW& w, float* y, int64_t ldy, int T_out, int stride,
         int pad_l, int pad_r, int act, const float* res = nullptr, int64_t ldr = 0) {
    if (c.dry) return QA_OK;
    qa_conv_args a{};
    a.x = x; a.w = w.w; a.bias = w.b; a.residual = res; a.y = y;
    a.B = B; a.T_in = T_in; a.C_in = w.C_in; a.T_out = T_out; a.N = w.N;
    a.ldx = ldx; a.ldy = ldy; a.ldr = ldr;
    a.ksize = w.ksize; a.stride = stride; a.pad_left = pad_l; a.pad_right = pad_r; a.pad_mode = PAD_ZERO;
    a.act = act;
    ConvParams p;
    QA_TRY(conv_params_from_args(a, &p));
    return launch_conv_gemm(p, c.stream);
}
int linear(Ctx& c, const float* x, int64_t rows, const ConvW& w, float* y, int act = ACT_NONE, const float* res = nullptr) {
    return conv(c, x, w.C_in, 1, (int)rows, w, y, w.N, (int)rows, 1, 0, 0, act, res, w.N);
}
int layernorm(Ctx& c, const float* x, const float* w, const float* b, float* y, int64_t rows, int C, float eps) {
    if (c.dry) return QA_OK;
    return launch_layernorm(x, w, b, y, rows, C, eps, c.stream);
}

int64_t frames_of(const qa_ssl_spec& sp, int64_t T) {
    int64_t L = T + 2 * (int64_t)sp.pad;
    for (int i = 0; i < sp.n_conv; ++i) {
        if (L < sp.conv_kernel[i]) return -1;
        L = (L - sp.conv_kernel[i]) / sp.conv_stride[i] + 1;
    }
    return L;
}

int build(qa_ssl* h, const HostTable& tab) {
    const qa_ssl_spec& sp = h->spec;
    QA_REQUIRE(sp.n_conv >= 2 && sp.n_conv <= 8, "ssl spec: n_conv = %d", sp.n_conv);
    const int d = sp.hidden, H = sp.n_heads, I = sp.intermediate;
    QA_REQUIRE(H > 0 && d % H == 0 && (d / H == 32 || d / H == 64 || d / H == 96 || d / H == 128), "ssl spec: head_dim %d unsupported",
               H > 0 ? d / H : 0);
    QA_REQUIRE(d % 32 == 0 && I % 32 == 0, "ssl spec: hidden / intermediate must be multiples of 32");
    QA_REQUIRE(sp.pos_groups > 0 && d % sp.pos_groups == 0 && (d / sp.pos_groups) % 16 == 0 && d / sp.pos_groups > 32,
               "ssl spec: %d channels per positional-conv group unsupported", sp.pos_groups > 0 ? d / sp.pos_groups : 0);
    for (int i = 0; i < sp.n_conv; ++i)
        QA_REQUIRE(sp.conv_dim[i] % 32 == 0 && sp.conv_kernel[i] >= 1 && sp.conv_stride[i] >= 1, "ssl spec: conv layer %d", i);
    QA_REQUIRE(sp.n_select >= 0 && sp.n_select <= 32, "ssl spec: n_select");
    if (sp.n_select == 0)
        for (int i = 0; i <= sp.n_layers; ++i) h->select.push_back(i);
    else
        for (int i = 0; i < sp.n_select; ++i) {
            QA_REQUIRE(sp.select[i] >= 0 && sp.select[i] <= sp.n_layers, "ssl spec: hidden state %d does not exist", sp.select[i]);
            h->select.push_back(sp.select[i]);
        }

    WeightStore& st = h->store;
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
    // Conv1d weight [N, C, k] (PyTorch) -> library layout [N][k][C]
    auto convw = [&](ConvW* w, const std::string& prefix, int N, int C, int k, bool bias) {
        w->N = N; w->C_in = C; w->ksize = k;
        const float* src = tab.get(prefix + ".weight", (int64_t)N * C * k);
        if (!src) {
            ok = false;
            return;
        }
        std::vector<float> t((size_t)N * k * C);
        for (int n = 0; n < N; ++n)
            for (int cc = 0; cc < C; ++cc)
                for (int j = 0; j < k; ++j) t[((size_t)n * k + j) * C + cc] = src[((size_t)n * C + cc) * k + j];
        pend.push_back({&w->w, st.add(t)});
        if (bias) vec(&w->b, prefix + ".bias", N);
    };
    auto linw = [&](ConvW* w, const std::string& prefix, int N, int C) {
        w->N = N; w->C_in = C; w->ksize = 1;
        vec(&w->w, prefix + ".weight", (int64_t)N * C);
        vec(&w->b, prefix + ".bias", N);
    };

    // ---- feature extractor
    const int C0 = sp.conv_dim[0], k0 = sp.conv_kernel[0];
    {
        const float* src = tab.get("feature_extractor.conv_layers.0.conv.weight", (int64_t)C0 * k0);
        if (src) {
            std::vector<float> t((size_t)k0 * C0);
            for (int cc = 0; cc < C0; ++cc)
                for (int j = 0; j < k0; ++j) t[(size_t)j * C0 + cc] = src[(size_t)cc * k0 + j];
            pend.push_back({&h->conv0_w, st.add(t)});
        } else {
            ok = false;
        }
        if (sp.conv_bias) vec(&h->conv0_b, "feature_extractor.conv_layers.0.conv.bias", C0);
    }
    h->cln_w.assign(sp.n_conv, nullptr);
    h->cln_b.assign(sp.n_conv, nullptr);
    if (sp.feat_norm_layer) {
        for (int i = 0; i < sp.n_conv; ++i) {
            const std::string pre = "feature_extractor.conv_layers." + std::to_string(i) + ".layer_norm.";
            vec(&h->cln_w[i], pre + "weight", sp.conv_dim[i]);
            vec(&h->cln_b[i], pre + "bias", sp.conv_dim[i]);
        }
    } else {
        vec(&h->gn_w, "feature_extractor.conv_layers.0.layer_norm.weight", C0);
        vec(&h->gn_b, "feature_extractor.conv_layers.0.layer_norm.bias", C0);
    }
    h->convs.resize(sp.n_conv - 1);
    for (int i = 1; i < sp.n_conv; ++i)
        convw(&h->convs[i - 1], "feature_extractor.conv_layers." + std::to_string(i) + ".conv", sp.conv_dim[i], sp.conv_dim[i - 1],
              sp.conv_kernel[i], sp.conv_bias != 0);
    const int CL = sp.conv_dim[sp.n_conv - 1];
    // ---- feature projection
    vec(&h->fp_ln_w, "feature_projection.layer_norm.weight", CL);
    vec(&h->fp_ln_b, "feature_projection.layer_norm.bias", CL);
    linw(&h->fp, "feature_projection.projection", d, CL);
    // ---- positional convolution: weight_norm(dim = 2) folded, then one [cg][k][cg] filter bank per group
    {
        const int G = sp.pos_groups, cg = d / G, k = sp.pos_kernel;
        const std::string pre = "encoder.pos_conv_embed.conv.";
        std::vector<float> wfull((size_t)d * cg * k);
        const int64_t n = (int64_t)d * cg * k;
        const float *g = nullptr, *v = nullptr;
        if (tab.has(pre + "parametrizations.weight.original0")) {
            g = tab.get(pre + "parametrizations.weight.original0", k);
            v = tab.get(pre + "parametrizations.weight.original1", n);
        } else if (tab.has(pre + "weight_g")) {
            g = tab.get(pre + "weight_g", k);
            v = tab.get(pre + "weight_v", n);
        } else {
            v = tab.get(pre + "weight", n);
        }
        if (!v || ((tab.has(pre + "parametrizations.weight.original0") || tab.has(pre + "weight_g")) && !g)) {
            ok = false;
        } else {
            std::vector<double> scale(k, 1.0);
            if (g)
                for (int j = 0; j < k; ++j) {  // norm over (out, in) for every kernel position
                    double ss = 0.0;
                    for (int64_t e = 0; e < (int64_t)d * cg; ++e) ss += (double)v[e * k + j] * v[e * k + j];
                    scale[j] = (double)g[j] / std::sqrt(ss);
                }
            const float* bias = tab.get(pre + "bias", d);
            if (!bias) ok = false;
            h->pos.resize(G);
            for (int gi = 0; gi < G && bias; ++gi) {
                std::vector<float> t((size_t)cg * k * cg);
                for (int o = 0; o < cg; ++o)
                    for (int ci = 0; ci < cg; ++ci)
                        for (int j = 0; j < k; ++j)
                            t[((size_t)o * k + j) * cg + ci] = (float)(v[((size_t)(gi * cg + o) * cg + ci) * k + j] * scale[j]);
                ConvW& w = h->pos[gi];
                w.N = cg; w.C_in = cg; w.ksize = k;
                pend.push_back({&w.w, st.add(t)});
                pend.push_back({&w.b, st.add(bias + (size_t)gi * cg, cg)});
            }
        }
    }
    vec(&h->enc_ln_w, "encoder.layer_norm.weight", d);
    vec(&h->enc_ln_b, "encoder.layer_norm.bias", d);
    // ---- encoder layers
    h->layers.resize(sp.n_layers);
    for (int i = 0; i < sp.n_layers; ++i) {
        SslLayer& L = h->layers[i];
        const std::string pre = "encoder.layers." + std::to_string(i) + ".";
        {  // q, k, v projections concatenated -> one [3d, d] GEMM
            std::vector<float> w((size_t)3 * d * d), b((size_t)3 * d);
            const char* nm[3] = {"q_proj", "k_proj", "v_proj"};
            for (int j = 0; j < 3; ++j) {
                const float* ws_ = tab.get(pre + "attention." + nm[j] + ".weight", (int64_t)d * d);
                const float* bs_ = tab.get(pre + "attention." + nm[j] + ".bias", d);
                if (!ws_ || !bs_) {
                    ok = false;
                    break;
                }
                std::memcpy(w.data() + (size_t)j * d * d, ws_, sizeof(float) * (size_t)d * d);
                std::memcpy(b.data() + (size_t)j * d, bs_, sizeof(float) * d);
            }
            L.qkv.N = 3 * d; L.qkv.C_in = d; L.qkv.ksize = 1;
            pend.push_back({&L.qkv.w, st.add(w)});
            pend.push_back({&L.qkv.b, st.add(b)});
        }
        linw(&L.o, pre + "attention.out_proj", d, d);
        vec(&L.ln1w, pre + "layer_norm.weight", d);
        vec(&L.ln1b, pre + "layer_norm.bias", d);
        linw(&L.ff1, pre + "feed_forward.intermediate_dense", I, d);
        linw(&L.ff2, pre + "feed_forward.output_dense", d, I);
        vec(&L.ln2w, pre + "final_layer_norm.weight", d);
        vec(&L.ln2b, pre + "final_layer_norm.bias", d);
        if (sp.rel_pos_buckets > 0) {  // gate = f(sum of 4 outputs): fold rows 0..3 and 4..7 of the 8 x hd projection
            const int hd = d / H;
            const float* gw = tab.get(pre + "attention.gru_rel_pos_linear.weight", (int64_t)8 * hd);
            const float* gb = tab.get(pre + "attention.gru_rel_pos_linear.bias", 8);
            if (!gw || !gb) {
                ok = false;
            } else {
                std::vector<float> w2((size_t)2 * hd, 0.f), b2(2, 0.f);
                for (int r = 0; r < 8; ++r) {
                    for (int e = 0; e < hd; ++e) w2[(size_t)(r / 4) * hd + e] += gw[(size_t)r * hd + e];
                    b2[r / 4] += gb[r];
                }
                pend.push_back({&L.gate_w, st.add(w2)});
                pend.push_back({&L.gate_b, st.add(b2)});
            }
            vec(&L.gate_c, pre + "attention.gru_rel_pos_const", H);
        }
    }
    if (sp.rel_pos_buckets > 0) {
        // WavLMAttention.compute_bias / _relative_positions_bucket, tabulated by relative distance r = key - query (float32
        // arithmetic like the reference).  For |r| >= max_distance the bucket is saturated, so clamping r is exact.
        QA_REQUIRE(sp.rel_pos_buckets % 4 == 0 && sp.rel_pos_max_distance > sp.rel_pos_buckets / 4, "ssl spec: relative position buckets");
        const float* emb = tab.get("encoder.layers.0.attention.rel_attn_embed.weight", (int64_t)sp.rel_pos_buckets * H);
        if (!emb) {
            ok = false;
        } else {
            const int R = sp.rel_pos_max_distance, nb = sp.rel_pos_buckets / 2, max_exact = nb / 2;
            std::vector<float> t((size_t)H * (2 * R + 1));
            const float denom = (float)std::log((double)sp.rel_pos_max_distance / max_exact);
            for (int r = -R; r <= R; ++r) {
                int bucket = r > 0 ? nb : 0;
                const int a = r < 0 ? -r : r;
                if (a < max_exact) {
                    bucket += a;
                } else {
                    float v = std::log((float)a / (float)max_exact);
                    v = v / denom;
                    v = v * (float)(nb - max_exact);
                    long long big = (long long)((float)max_exact + v);
                    if (big > nb - 1) big = nb - 1;
                    bucket += (int)big;
                }
                for (int hh = 0; hh < H; ++hh) t[(size_t)hh * (2 * R + 1) + (r + R)] = emb[(size_t)bucket * H + hh];
            }
            pend.push_back({&h->relbias, st.add(t)});
        }
    }
    if (!ok) return QA_ERR_INVALID;
    QA_TRY(st.upload());
    for (auto& pv : pend) *pv.first = st.ptr(pv.second);
    return QA_OK;
}

int forward_graph(qa_ssl* h, Ctx& c, const float* wav, int B, int T, float* feats) {
    const qa_ssl_spec& sp = h->spec;
    const int d = sp.hidden, H = sp.n_heads, hd = d / H, I = sp.intermediate;
    const float eps = sp.layer_norm_eps;
    // ---- feature extractor (HubertFeatureEncoder): conv -> [GroupNorm | LayerNorm] -> GELU
    int L = (int)((T + 2 * (int64_t)sp.pad - sp.conv_kernel[0]) / sp.conv_stride[0] + 1);
    int C = sp.conv_dim[0];
    float* x = c.arena.alloc<float>((size_t)B * L * C);
    {
        const size_t mark = c.arena.mark();
        char* scratch = c.arena.alloc<char>(ssl_conv0_scratch_bytes(B, L, C));
        if (!c.dry)
            QA_TRY(launch_ssl_conv0(wav, h->conv0_w, h->conv0_b, h->gn_w, h->gn_b, x, scratch, B, T, L, C, sp.conv_kernel[0],
                                    sp.conv_stride[0], sp.pad, sp.feat_norm_layer ? 0 : 1, 1e-5f, sp.feat_norm_layer ? ACT_NONE : ACT_GELU,
                                    c.stream));
        c.arena.release(mark);
        if (sp.feat_norm_layer) {
            QA_TRY(layernorm(c, x, h->cln_w[0], h->cln_b[0], x, (int64_t)B * L, C, 1e-5f));
            if (!c.dry) QA_TRY(launch_ssl_act(x, (long long)B * L * C, ACT_GELU, c.stream));
        }
    }
    c.tap("ssl.conv0", x, (int64_t)B * L * C);
    for (int i = 1; i < sp.n_conv; ++i) {
        const ConvW& w = h->convs[i - 1];
        const int Lo = (L - w.ksize) / sp.conv_stride[i] + 1;
        QA_REQUIRE(Lo >= 1, "ssl: input too short at conv layer %d", i);
        float* y = c.arena.alloc<float>((size_t)B * Lo * w.N);
        QA_TRY(conv(c, x, C, B, L, w, y, w.N, Lo, sp.conv_stride[i], 0, 0, sp.feat_norm_layer ? ACT_NONE : ACT_GELU));
        if (sp.feat_norm_layer) {
            QA_TRY(layernorm(c, y, h->cln_w[i], h->cln_b[i], y, (int64_t)B * Lo, w.N, 1e-5f));
            if (!c.dry) QA_TRY(launch_ssl_act(y, (long long)B * Lo * w.N, ACT_GELU, c.stream));
        }
        x = y;
        L = Lo;
        C = w.N;
    }
    c.tap("ssl.extract", x, (int64_t)B * L * C);
    const int N = L;
    const int64_t rows = (int64_t)B * N;
    // ---- feature projection: LayerNorm -> Linear
    float* t0 = c.arena.alloc<float>((size_t)rows * std::max(C, d));
    float* hcur = c.arena.alloc<float>((size_t)rows * d);
    float* hnext = c.arena.alloc<float>((size_t)rows * d);
    float* tmp = c.arena.alloc<float>((size_t)rows * d);
    float* qkv = c.arena.alloc<float>((size_t)rows * 3 * d);
    float* att = c.arena.alloc<float>((size_t)rows * d);
    float* ffu = c.arena.alloc<float>((size_t)rows * I);
    float* acc = c.arena.alloc<float>((size_t)rows * d);
    const bool rel = sp.rel_pos_buckets > 0;
    float* gate = rel ? c.arena.alloc<float>((size_t)rows * H) : nullptr;
    QA_TRY(layernorm(c, x, h->fp_ln_w, h->fp_ln_b, t0, rows, C, eps));
    QA_TRY(linear(c, t0, rows, h->fp, hcur));
    // ---- encoder front: h = h + GELU(pos_conv(h))  (HubertPositionalConvEmbedding; the even kernel's extra output frame is
    // never computed), then LayerNorm for the post-LN flavour
    {
        const int G = sp.pos_groups, cg = d / G, k = sp.pos_kernel;
        for (int g = 0; g < G; ++g)
            QA_TRY(conv(c, hcur + (size_t)g * cg, d, B, N, h->pos[g], hnext + (size_t)g * cg, d, N, 1, k / 2, k - 1 - k / 2, ACT_GELU,
                        hcur + (size_t)g * cg, d));
        if (!sp.stable_layer_norm) {
            QA_TRY(layernorm(c, hnext, h->enc_ln_w, h->enc_ln_b, hcur, rows, d, eps));
        } else {
            std::swap(hcur, hnext);
        }
    }
    c.tap("ssl.hidden0", hcur, rows * d);
    int n_acc = 0;
    auto maybe_accumulate = [&](int index, const float* hs) -> int {
        for (int sidx : h->select)
            if (sidx == index) {
                if (!c.dry) QA_TRY(launch_ssl_accumulate(acc, hs, (long long)rows * d, n_acc == 0, c.stream));
                ++n_acc;
            }
        return QA_OK;
    };
    QA_TRY(maybe_accumulate(0, hcur));
    const float scale = 1.0f / std::sqrt((float)hd);
    for (int i = 0; i < sp.n_layers; ++i) {
        const SslLayer& Lw = h->layers[i];
        if (!sp.stable_layer_norm) {  // HubertEncoderLayer: x = LN(x + Attn(x)); x = LN(x + FFN(x))
            QA_TRY(linear(c, hcur, rows, Lw.qkv, qkv));
            if (!c.dry) {
                if (rel) QA_TRY(launch_ssl_gate(hcur, Lw.gate_w, Lw.gate_b, Lw.gate_c, gate, B, N, H, hd, c.stream));
                QA_TRY(launch_attention(qkv, 3 * d, qkv + d, qkv + 2 * d, 3 * d, att, d, B, N, N, (long long)N * 3 * d, H, hd, scale, 0,
                                        c.stream, gate, rel ? h->relbias : nullptr, sp.rel_pos_max_distance));
            }
            QA_TRY(linear(c, att, rows, Lw.o, tmp, ACT_NONE, hcur));
            QA_TRY(layernorm(c, tmp, Lw.ln1w, Lw.ln1b, hnext, rows, d, eps));
            QA_TRY(linear(c, hnext, rows, Lw.ff1, ffu, ACT_GELU));
            QA_TRY(linear(c, ffu, rows, Lw.ff2, tmp, ACT_NONE, hnext));
            QA_TRY(layernorm(c, tmp, Lw.ln2w, Lw.ln2b, hcur, rows, d, eps));
            QA_TRY(maybe_accumulate(i + 1, hcur));
        } else {  // HubertEncoderLayerStableLayerNorm: x = x + Attn(LN(x)); x = x + FFN(LN(x)); final LN after the last layer
            QA_TRY(layernorm(c, hcur, Lw.ln1w, Lw.ln1b, tmp, rows, d, eps));
            QA_TRY(linear(c, tmp, rows, Lw.qkv, qkv));
            if (!c.dry) {
                if (rel) QA_TRY(launch_ssl_gate(tmp, Lw.gate_w, Lw.gate_b, Lw.gate_c, gate, B, N, H, hd, c.stream));
                QA_TRY(launch_attention(qkv, 3 * d, qkv + d, qkv + 2 * d, 3 * d, att, d, B, N, N, (long long)N * 3 * d, H, hd, scale, 0,
                                        c.stream, gate, rel ? h->relbias : nullptr, sp.rel_pos_max_distance));
            }
            QA_TRY(linear(c, att, rows, Lw.o, hcur, ACT_NONE, hcur));
            QA_TRY(layernorm(c, hcur, Lw.ln2w, Lw.ln2b, tmp, rows, d, eps));
            QA_TRY(linear(c, tmp, rows, Lw.ff1, ffu, ACT_GELU));
            QA_TRY(linear(c, ffu, rows, Lw.ff2, hcur, ACT_NONE, hcur));
            if (i == sp.n_layers - 1) {
                QA_TRY(layernorm(c, hcur, h->enc_ln_w, h->enc_ln_b, tmp, rows, d, eps));
                QA_TRY(maybe_accumulate(i + 1, tmp));
            } else {
                QA_TRY(maybe_accumulate(i + 1, hcur));
            }
        }
    }
    QA_REQUIRE(n_acc > 0, "ssl: no hidden state selected");
    if (!c.dry) QA_TRY(launch_ssl_compress(acc, feats, (long long)rows * d, 1.0f / (float)n_acc, sp.compress_exponent, c.stream));
    return QA_OK;
}

int ensure_ws(qa_ssl* h, size_t bytes) {
    if (bytes <= h->ws_cap) return QA_OK;
    if (h->ws) QA_HIP(hipFree(h->ws));
    h->ws = nullptr;
    h->ws_cap = 0;
    const size_t cap = bytes + bytes / 16;
    QA_HIP(hipMalloc(reinterpret_cast<void**>(&h->ws), cap));
    h->ws_cap = cap;
    return QA_OK;
}

}  // namespace

extern "C" {

int qa_ssl_create(qa_ssl** out, const qa_ssl_spec* spec, const qa_tensor* tensors, int64_t n_tensors, int device) {
    if (!out || !spec || !tensors) {
        set_error("qa_ssl_create: null argument");
        return QA_ERR_INVALID;
    }
    *out = nullptr;
    QA_HIP(hipSetDevice(device));
    std::unique_ptr<qa_ssl> h(new qa_ssl());
    h->spec = *spec;
    h->device = device;
    HostTable tab(tensors, n_tensors);
    const int st = build(h.get(), tab);
    if (st != QA_OK) {
        h->store.release();
        return st;
    }
    *out = h.release();
    return QA_OK;
}

void qa_ssl_destroy(qa_ssl* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    h->store.release();
    if (h->ws) (void)hipFree(h->ws);
    delete h;
}

int64_t qa_ssl_frames(const qa_ssl* h, int64_t T) {
    if (!h) {
        set_error("qa_ssl_frames: null handle");
        return QA_ERR_INVALID;
    }
    const int64_t n = frames_of(h->spec, T);
    if (n < 1) {
        set_error("qa_ssl_frames: %lld samples are too short for the feature extractor", (long long)T);
        return QA_ERR_INVALID;
    }
    return n;
}

int qa_ssl_forward(qa_ssl* h, const float* wav, int64_t B, int64_t T, float* feats, void* stream) {
    if (!h || !wav || !feats) {
        set_error("qa_ssl_forward: null argument");
        return QA_ERR_INVALID;
    }
    const int64_t N = frames_of(h->spec, T);
    QA_REQUIRE(B > 0 && N >= 1, "qa_ssl_forward: wav is [%lld, %lld]: too short for the feature extractor", (long long)B, (long long)T);
    const int64_t L0 = (T + 2 * (int64_t)h->spec.pad - h->spec.conv_kernel[0]) / h->spec.conv_stride[0] + 1;
    QA_REQUIRE(B * L0 < (1LL << 31) && L0 * h->spec.conv_dim[0] < (1LL << 31), "qa_ssl_forward: batch of %lld x %lld samples is too large",
               (long long)B, (long long)T);
    QA_HIP(hipSetDevice(h->device));
    Ctx& c = h->ctx;
    c.stream = static_cast<hipStream_t>(stream);
    c.dry = true;
    c.arena.begin(nullptr, 0);
    QA_TRY(forward_graph(h, c, wav, (int)B, (int)T, feats));
    QA_TRY(ensure_ws(h, c.arena.peak()));
    c.dry = false;
    c.taps.clear();
    c.arena.begin(h->ws, h->ws_cap);
    return forward_graph(h, c, wav, (int)B, (int)T, feats);
}

}  // extern "C"
