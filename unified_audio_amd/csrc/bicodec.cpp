// bicodec.cpp - BiCodec.detokenize on the HIP kernels (host orchestration, C-ABI handle): semantic tokens [B, T] + global tokens
// [B, token_num] -> waveform [B, T * prod(rates)], the stage the reference's UniSE test path ends with
// (QuarkAudio-UniSE/model/bicodec/bicodec.py:182-199, called at model/model.py:193,223).  SURVEY.md 8f-2.
//
//   z_q      FactorizedVectorQuantize.detokenize  (vq/factorized_vector_quantize.py:154-172)  one gather: codebook x out_project folded
//   d_vector SpeakerEncoder.detokenize            (speaker/speaker_encoder.py:111-116)        gather of (FSQ code x project_out) + Linear
//   prenet   Decoder.forward, ratios [1, 1]       (encoder_decoder/feat_decoder.py:79-96)     linears / k7 convs on conv_gemm, AdaLN kernel
//   x += d   bicodec.py:196
//   decoder  WaveGenerator.forward                (encoder_decoder/wave_generator.py:61-91)   every conv on conv_gemm:
//            * ConvTranspose1d(k, stride s, padding (k - s) / 2) as s polyphase stride-1 convolutions of ceil((k - j0) / s) taps that
//              write interleaved rows of the output (row stride s * C_out): no zero-stuffed input, no scatter;
//            * dilated k7 convolutions through the tap table (dilation 1 / 3 / 9);
//            * Snake never costs a pass over HBM: it is the epilogue activation of the producing convolution, and where a tensor is
//              needed both raw (residual skip) and activated (next unit's first conv) the epilogue writes both (ConvParams::y2).
// Layout: channel-last [B, frames, C] everywhere; weights arrive with the reference's state_dict keys (weight_g / weight_v folded here).
#include <memory>

#include "host_util.h"

namespace qa {
int launch_gather_rows(const long long* tok, const float* table, float* out, long long n, int V, int D, hipStream_t s);
int launch_gather_global(const long long* tok, const float* table, float* out, int B, int N, int V, int L, hipStream_t s);
int launch_adaln(const float* x, const float* scale, const float* shift, long long ld_cond, float* y, int B, int T, int C, float eps,
                 hipStream_t s);
int launch_add_rowvec(float* x, const float* v, int B, int T, int C, hipStream_t s);
int launch_skinny_gemm(const float* x, long long ldx, const float* w, const float* bias, const float* gate, long long ldg,
                       const float* res, long long ldr, float* y, long long ldy, int M, int N, int K, int act, hipStream_t s,
                       float rms_eps, int dual);
}  // namespace qa

using namespace qa;

namespace {

struct VocosLayerW {
    const float *dw = nullptr, *dwb = nullptr, *lnw = nullptr, *lnb = nullptr, *gamma = nullptr;  // lnw == nullptr: AdaLN
    ConvW pw1, pw2;
};
struct VocosW {
    ConvW embed;
    const float *nw = nullptr, *nb = nullptr, *fw = nullptr, *fb = nullptr;
    std::vector<VocosLayerW> layers;
};
struct UnitW {
    const float *a1 = nullptr, *a2 = nullptr;  // Snake alphas in front of the k7 and the k1 convolution
    ConvW c7, c1;
    int dilation = 1;
};
struct GenBlockW {
    const float* a_in = nullptr;    // Snake in front of the ConvTranspose1d
    std::vector<ConvW> phase;       // stride polyphase filters, output phase phi
    std::vector<int> pad_left;      // per phase
    int stride = 1, c_in = 0, c_out = 0;
    UnitW unit[3];
};

}  // namespace

struct qa_bicodec {
    qa_bicodec_spec spec{};
    int device = 0;
    int n_glob = 0, hop = 1, n_ada = 0;
    WeightStore store;
    const float *sem_table = nullptr, *glob_table = nullptr;
    ConvW project, ada, linear_pre, linear_out, gen_in, gen_out;
    VocosW down[2], backbone;
    std::vector<GenBlockW> blocks;
    const float* a_final = nullptr;
    char* ws = nullptr;
    size_t ws_cap = 0;
    Ctx ctx;
};

namespace {

// ---------------------------------------------------------------- weight folding (host)
struct Loader {
    const HostTable& tab;
    WeightStore& st;
    bool ok = true;
    std::vector<std::pair<const float**, size_t>> pend;

    const float* need(const std::string& name, int64_t n) {
        const float* p = tab.get(name, n);
        if (!p) ok = false;
        return p;
    }
    void vec(const float** dst, const std::string& name, int64_t n) {
        const float* p = need(name, n);
        if (p) pend.push_back({dst, st.add(p, (size_t)n)});
    }
    void raw(const float** dst, const std::vector<float>& v) { pend.push_back({dst, st.add(v)}); }
    // torch.nn.utils.weight_norm (dim 0) or a plain .weight: returns the [d0][rest] tensor
    bool weight(const std::string& p, int64_t d0, int64_t rest, std::vector<float>* out) {
        out->assign((size_t)(d0 * rest), 0.f);
        if (tab.has(p + ".weight")) {
            const float* w = need(p + ".weight", d0 * rest);
            if (!w) return false;
            std::memcpy(out->data(), w, sizeof(float) * (size_t)(d0 * rest));
            return true;
        }
        const float* v = need(p + ".weight_v", d0 * rest);
        const float* g = need(p + ".weight_g", d0);
        if (!v || !g) return false;
        for (int64_t i = 0; i < d0; ++i) {
            double ss = 0.0;
            for (int64_t j = 0; j < rest; ++j) ss += (double)v[i * rest + j] * v[i * rest + j];
            const float scale = g[i] / (float)std::sqrt(ss);
            for (int64_t j = 0; j < rest; ++j) (*out)[(size_t)(i * rest + j)] = v[i * rest + j] * scale;
        }
        return true;
    }
    // Conv1d / Linear weight [N][C_in][k] -> library layout [N][k][C_in] (scaled by `gain`), bias [N]
    void conv(ConvW* dst, const std::string& p, int N, int C_in, int k, bool bias = true, float gain = 1.f) {
        std::vector<float> w, r((size_t)N * k * C_in);
        if (weight(p, N, (int64_t)C_in * k, &w))
            for (int n = 0; n < N; ++n)
                for (int c = 0; c < C_in; ++c)
                    for (int j = 0; j < k; ++j) r[((size_t)n * k + j) * C_in + c] = w[((size_t)n * C_in + c) * k + j] * gain;
        dst->N = N; dst->C_in = C_in; dst->ksize = k;
        raw(&dst->w, r);
        if (bias) vec(&dst->b, p + ".bias", N);
    }
    void resolve() {
        for (auto& pv : pend) *pv.first = st.ptr(pv.second);
    }
};

void build_vocos(Loader& L, VocosW* v, const std::string& p, int C, int I, int n_layers, bool ada, float embed_gain) {
    L.conv(&v->embed, p + ".embed", C, C, 7, true, embed_gain);
    if (!ada) {
        L.vec(&v->nw, p + ".norm.weight", C);
        L.vec(&v->nb, p + ".norm.bias", C);
    }
    v->layers.resize(n_layers);
    for (int i = 0; i < n_layers; ++i) {
        VocosLayerW& w = v->layers[i];
        const std::string q = p + ".convnext." + std::to_string(i);
        // depthwise filter [C][1][7] -> [7][C]
        const float* dw = L.need(q + ".dwconv.weight", (int64_t)C * 7);
        std::vector<float> kc((size_t)7 * C, 0.f);
        if (dw)
            for (int c = 0; c < C; ++c)
                for (int j = 0; j < 7; ++j) kc[(size_t)j * C + c] = dw[(size_t)c * 7 + j];
        L.raw(&w.dw, kc);
        L.vec(&w.dwb, q + ".dwconv.bias", C);
        if (!ada) {
            L.vec(&w.lnw, q + ".norm.weight", C);
            L.vec(&w.lnb, q + ".norm.bias", C);
        }
        L.vec(&w.gamma, q + ".gamma", C);
        L.conv(&w.pw1, q + ".pwconv1", I, C, 1);
        L.conv(&w.pw2, q + ".pwconv2", C, I, 1);
    }
    L.vec(&v->fw, p + ".final_layer_norm.weight", C);
    L.vec(&v->fb, p + ".final_layer_norm.bias", C);
}

int build(qa_bicodec* h, const HostTable& tab) {
    const qa_bicodec_spec& sp = h->spec;
    const int Ld = sp.latent_dim, C = sp.vocos_dim, I = sp.vocos_inter;
    QA_REQUIRE(sp.n_levels >= 1 && sp.n_levels <= 8 && sp.n_rates >= 1 && sp.n_rates <= 8, "bicodec spec: bad level / rate count");
    QA_REQUIRE(Ld % 32 == 0 && C % 32 == 0 && I % 32 == 0 && sp.gen_channels % (32 << sp.n_rates) == 0,
               "bicodec spec: latent %d, vocos %d / %d must be multiples of 32 and gen_channels %d of 32 * 2^n_rates", Ld, C, I, sp.gen_channels);
    QA_REQUIRE((sp.spk_latent_dim * sp.token_num) % 32 == 0 && sp.codebook_dim >= 1 && sp.codebook_size >= 1, "bicodec spec: bad quantizer shape");
    h->n_glob = 1;
    h->hop = 1;
    for (int i = 0; i < sp.n_levels; ++i) h->n_glob *= sp.levels[i];
    for (int i = 0; i < sp.n_rates; ++i) {
        QA_REQUIRE(sp.rates[i] >= 1 && sp.kernel_sizes[i] >= sp.rates[i] && (sp.kernel_sizes[i] - sp.rates[i]) % 2 == 0,
                   "bicodec spec: ConvTranspose1d kernel %d / stride %d: length-exact up-sampling needs k - s even", sp.kernel_sizes[i], sp.rates[i]);
        h->hop *= sp.rates[i];
    }
    Loader L{tab, h->store};
    // ---- semantic tokens: table[v] = out_project(codebook[v]) + bias   (factorized_vector_quantize.py:154-172; k = 1 conv = Linear)
    {
        std::vector<float> w;
        const float* cb = L.need("quantizer.codebook.weight", (int64_t)sp.codebook_size * sp.codebook_dim);
        const float* b = L.need("quantizer.out_project.bias", Ld);
        std::vector<float> t((size_t)sp.codebook_size * Ld, 0.f);
        if (L.weight("quantizer.out_project", Ld, sp.codebook_dim, &w) && cb && b)
            for (int v = 0; v < sp.codebook_size; ++v)
                for (int n = 0; n < Ld; ++n) {
                    float acc = 0.f;  // fp32 products in k order, then the bias, like the convolution
                    for (int k = 0; k < sp.codebook_dim; ++k) acc = std::fmaf(cb[(size_t)v * sp.codebook_dim + k], w[(size_t)n * sp.codebook_dim + k], acc);
                    t[(size_t)v * Ld + n] = acc + b[n];
                }
        L.raw(&h->sem_table, t);
    }
    // ---- global tokens: table[v] = project_out(fsq_code(v)) + bias   (residual_fsq.py:112-156, finite_scalar_quantization.py:165-183)
    {
        const int Ls = sp.spk_latent_dim, nl = sp.n_levels;
        const float* w = L.need("speaker_encoder.quantizer.project_out.weight", (int64_t)Ls * nl);
        const float* b = L.need("speaker_encoder.quantizer.project_out.bias", Ls);
        std::vector<float> t((size_t)h->n_glob * Ls, 0.f);
        if (w && b)
            for (int v = 0; v < h->n_glob; ++v) {
                float code[8];
                int rem = v;
                for (int d = 0; d < nl; ++d) {  // least-significant digit first: basis = cumprod([1] + levels[:-1])
                    const int digit = rem % sp.levels[d], half = sp.levels[d] / 2;
                    rem /= sp.levels[d];
                    code[d] = (float)(digit - half) / (float)half;
                }
                for (int n = 0; n < Ls; ++n) {
                    float acc = 0.f;
                    for (int d = 0; d < nl; ++d) acc = std::fmaf(code[d], w[(size_t)n * nl + d], acc);
                    t[(size_t)v * Ls + n] = acc + b[n];
                }
            }
        L.raw(&h->glob_table, t);
    }
    L.conv(&h->project, "speaker_encoder.project", Ld, sp.spk_latent_dim * sp.token_num, 1);
    // ---- prenet (feat_decoder.py:48-96).  SamplingBlock with both scales 1 returns 3 x (samper.py:78-95): folded into the embed filters
    L.conv(&h->linear_pre, "prenet.linear_pre", C, Ld, 1);
    for (int i = 0; i < 2; ++i) build_vocos(L, &h->down[i], "prenet.downsample." + std::to_string(i) + ".1", C, I, 2, false, 3.0f);
    build_vocos(L, &h->backbone, "prenet.vocos_backbone", C, I, sp.vocos_layers, true, 1.0f);
    L.conv(&h->linear_out, "prenet.linear", Ld, C, 1);
    // every AdaLayerNorm's scale / shift Linear stacked into one [n_ada * 2 * C, latent] matrix: one GEMV per call
    h->n_ada = sp.vocos_layers + 1;
    {
        std::vector<float> w((size_t)h->n_ada * 2 * C * Ld, 0.f), b((size_t)h->n_ada * 2 * C, 0.f);
        for (int a = 0; a < h->n_ada; ++a) {
            const std::string p = a == 0 ? "prenet.vocos_backbone.norm" : "prenet.vocos_backbone.convnext." + std::to_string(a - 1) + ".norm";
            const char* part[2] = {".scale", ".shift"};
            for (int k = 0; k < 2; ++k) {
                const float* wp = L.need(p + part[k] + ".weight", (int64_t)C * Ld);
                const float* bp = L.need(p + part[k] + ".bias", C);
                if (wp) std::memcpy(&w[((size_t)a * 2 + k) * C * Ld], wp, sizeof(float) * (size_t)C * Ld);
                if (bp) std::memcpy(&b[((size_t)a * 2 + k) * C], bp, sizeof(float) * C);
            }
        }
        h->ada.N = h->n_ada * 2 * C; h->ada.C_in = Ld; h->ada.ksize = 1;
        L.raw(&h->ada.w, w);
        L.raw(&h->ada.b, b);
    }
    // ---- wave generator (wave_generator.py:61-91)
    int ch = sp.gen_channels;
    L.conv(&h->gen_in, "decoder.model.0", ch, Ld, 7);
    h->blocks.resize(sp.n_rates);
    for (int i = 0; i < sp.n_rates; ++i) {
        GenBlockW& g = h->blocks[i];
        const std::string p = "decoder.model." + std::to_string(i + 1) + ".block";
        const int cin = ch, cout = ch / 2, k = sp.kernel_sizes[i], s = sp.rates[i], pad = (k - s) / 2;
        g.stride = s; g.c_in = cin; g.c_out = cout;
        L.vec(&g.a_in, p + ".0.alpha", cin);
        // ConvTranspose1d weight [C_in][C_out][k] (weight_norm over dim 0 = C_in).  y[q s + phi] = sum_m x[q + c0 - m] W[:, :, j0 + m s]
        // with j0 = (phi + pad) mod s, c0 = (phi + pad) div s: a stride-1 convolution with taps jj = 0 .. n-1 <-> m = n-1-jj,
        // pad_left = n - 1 - c0 (n = number of taps of the phase), right padding c0.
        std::vector<float> w;
        const bool have = L.weight(p + ".1", cin, (int64_t)cout * k, &w);
        g.phase.resize(s);
        g.pad_left.resize(s);
        for (int phi = 0; phi < s; ++phi) {
            const int j0 = (phi + pad) % s, c0 = (phi + pad) / s, n = (k - j0 + s - 1) / s;
            QA_REQUIRE(n >= 1 && n - 1 - c0 >= 0, "bicodec: ConvTranspose1d phase %d of k=%d s=%d has no causal tap layout", phi, k, s);
            std::vector<float> r((size_t)cout * n * cin, 0.f);
            if (have)
                for (int o = 0; o < cout; ++o)
                    for (int jj = 0; jj < n; ++jj) {
                        const int j = j0 + (n - 1 - jj) * s;
                        for (int c = 0; c < cin; ++c) r[((size_t)o * n + jj) * cin + c] = w[((size_t)c * cout + o) * k + j];
                    }
            ConvW& cw = g.phase[phi];
            cw.N = cout; cw.C_in = cin; cw.ksize = n;
            L.raw(&cw.w, r);
            L.vec(&cw.b, p + ".1.bias", cout);
            g.pad_left[phi] = n - 1 - c0;
        }
        const int dil[3] = {1, 3, 9};
        for (int j = 0; j < 3; ++j) {
            UnitW& u = g.unit[j];
            const std::string up = p + "." + std::to_string(j + 2) + ".block";
            u.dilation = dil[j];
            L.vec(&u.a1, up + ".0.alpha", cout);
            L.conv(&u.c7, up + ".1", cout, cout, 7);
            L.vec(&u.a2, up + ".2.alpha", cout);
            L.conv(&u.c1, up + ".3", cout, cout, 1);
        }
        ch = cout;
    }
    L.vec(&h->a_final, "decoder.model." + std::to_string(sp.n_rates + 1) + ".alpha", ch);
    L.conv(&h->gen_out, "decoder.model." + std::to_string(sp.n_rates + 2), 1, ch, 7);
    if (!L.ok) return QA_ERR_MISSING;
    QA_TRY(h->store.upload());
    L.resolve();
    return QA_OK;
}

// ---------------------------------------------------------------- graph helpers
struct ConvOpt {
    int stride = 1, pad_left = 0, pad_right = 0, dilation = 1, act = ACT_NONE, post_act = ACT_NONE;
    const float *gamma = nullptr, *res = nullptr, *alpha = nullptr, *alpha2 = nullptr;
    float* y2 = nullptr;
    int64_t ldr = 0, ldy2 = 0;
};

int conv(Ctx& c, const float* x, int64_t ldx, int B, int T_in, const ConvW& w, float* y, int64_t ldy, int T_out, const ConvOpt& o) {
    if (c.dry) return QA_OK;
    qa_conv_args a{};
    a.x = x; a.w = w.w; a.bias = w.b; a.gamma = o.gamma; a.residual = o.res; a.y = y;
    a.B = B; a.T_in = T_in; a.C_in = w.C_in; a.T_out = T_out; a.N = w.N;
    a.ldx = ldx; a.ldy = ldy; a.ldr = o.ldr ? o.ldr : w.N; a.ldg = w.N;
    a.ksize = w.ksize; a.stride = o.stride; a.pad_left = o.pad_left; a.pad_right = o.pad_right; a.pad_mode = PAD_ZERO;
    a.act = o.act; a.post_act = o.post_act;
    ConvParams p;
    // conv_params_from_args checks the window span for a dense kernel: hand it the dense-equivalent paddings, then set the dilation
    if (o.dilation > 1) {
        a.pad_left = o.pad_left + (w.ksize - 1) * (o.dilation - 1);
        QA_TRY(conv_params_from_args(a, &p));
        p.pad_left = o.pad_left;
        const int max_pad = o.pad_left > o.pad_right ? o.pad_left : o.pad_right;
        p.Lp = (p.T_in <= max_pad) ? max_pad + 1 : p.T_in;
    } else {
        QA_TRY(conv_params_from_args(a, &p));
    }
    p.dilation = o.dilation;
    p.alpha = o.alpha; p.y2 = o.y2; p.alpha2 = o.alpha2; p.ldy2 = o.ldy2;
    return launch_conv_gemm(p, c.stream);
}

// per_item: one row per batch item (d-vector, AdaLN conditions) - always the weight-streaming skinny GEMM, 32 rows per launch, so that an
// item's arithmetic does not depend on how many items share the call (a row-count rule - skinny up to 32 rows, implicit GEMM above -
// made batches of more than 32 segments differ from smaller ones by 1e-5: found by the pipelined UniSE driver at 64 segments per batch)
int linear(Ctx& c, const float* x, int64_t rows, const ConvW& w, float* y, int act = ACT_NONE, const float* res = nullptr,
           const float* gamma = nullptr, bool per_item = false) {
    if (c.dry) return QA_OK;
    if (per_item && w.C_in % 256 == 0 && !gamma) {
        for (int64_t r0 = 0; r0 < rows; r0 += 32) {
            const int n = (int)std::min<int64_t>(32, rows - r0);
            QA_TRY(launch_skinny_gemm(x + r0 * w.C_in, w.C_in, w.w, w.b, nullptr, 0, res ? res + r0 * w.N : nullptr, w.N, y + r0 * w.N, w.N, n, w.N,
                                      w.C_in, act, c.stream, 0.f, 0));
        }
        return QA_OK;
    }
    ConvOpt o;
    o.act = act; o.res = res; o.gamma = gamma;
    return conv(c, x, w.C_in, 1, (int)rows, w, y, w.N, (int)rows, o);
}

// VocosBackbone.forward (blocks/vocos.py:323-335) in place on x [B, T, C]; t1 [rows, C], u [rows, I] scratch.
// cond: AdaLN scale / shift rows of this backbone ([B, n_ada * 2 * C], entry a at offset a * 2 * C), or nullptr for plain LayerNorm
int vocos(Ctx& c, const VocosW& v, float* x, float* t1, float* u, int B, int T, int C, const float* cond, int64_t ld_cond) {
    const int64_t rows = (int64_t)B * T;
    ConvOpt same7;
    same7.pad_left = 3; same7.pad_right = 3;
    QA_TRY(conv(c, x, C, B, T, v.embed, t1, C, T, same7));
    if (!c.dry) {
        if (cond) QA_TRY(launch_adaln(t1, cond, cond + C, ld_cond, x, B, T, C, 1e-6f, c.stream));
        else QA_TRY(launch_layernorm(t1, v.nw, v.nb, x, rows, C, 1e-6f, c.stream));
    }
    for (size_t i = 0; i < v.layers.size(); ++i) {
        const VocosLayerW& w = v.layers[i];
        if (!c.dry) {
            if (cond) {
                const float* sc = cond + (int64_t)(i + 1) * 2 * C;
                QA_TRY(launch_dwconv(x, w.dw, w.dwb, nullptr, nullptr, u, B, T, C, 7, 0.f, c.stream));  // u doubles as [rows, C] scratch
                QA_TRY(launch_adaln(u, sc, sc + C, ld_cond, t1, B, T, C, 1e-6f, c.stream));
            } else {
                QA_TRY(launch_dwconv(x, w.dw, w.dwb, w.lnw, w.lnb, t1, B, T, C, 7, 1e-6f, c.stream));
            }
        }
        QA_TRY(linear(c, t1, rows, w.pw1, u, ACT_GELU));
        QA_TRY(linear(c, u, rows, w.pw2, x, ACT_NONE, x, w.gamma));
    }
    if (!c.dry) {
        QA_TRY(launch_layernorm(x, v.fw, v.fb, t1, rows, C, 1e-6f, c.stream));
        QA_HIP(hipMemcpyAsync(x, t1, sizeof(float) * (size_t)rows * C, hipMemcpyDeviceToDevice, c.stream));
    }
    return QA_OK;
}

int detokenize_graph(qa_bicodec* h, Ctx& c, const long long* sem, const long long* glob, int B, int T, float* wav_out) {
    const qa_bicodec_spec& sp = h->spec;
    const int Ld = sp.latent_dim, C = sp.vocos_dim, I = sp.vocos_inter;
    const int64_t rows = (int64_t)B * T;
    // ---- tokens -> z_q [B, T, latent], d_vector [B, latent]
    float* zq = c.arena.alloc<float>(rows * Ld);
    float* gflat = c.arena.alloc<float>((size_t)B * sp.spk_latent_dim * sp.token_num);
    float* dvec = c.arena.alloc<float>((size_t)B * Ld);
    float* cond = c.arena.alloc<float>((size_t)B * h->ada.N);
    if (!c.dry) {
        QA_TRY(launch_gather_rows(sem, h->sem_table, zq, rows, sp.codebook_size, Ld, c.stream));
        QA_TRY(launch_gather_global(glob, h->glob_table, gflat, B, sp.token_num, h->n_glob, sp.spk_latent_dim, c.stream));
    }
    QA_TRY(linear(c, gflat, B, h->project, dvec, ACT_NONE, nullptr, nullptr, true));
    QA_TRY(linear(c, dvec, B, h->ada, cond, ACT_NONE, nullptr, nullptr, true));
    c.tap("z_q", zq, rows * Ld);
    c.tap("d_vector", dvec, (int64_t)B * Ld);
    // ---- prenet
    float* x = c.arena.alloc<float>(rows * C);
    float* t1 = c.arena.alloc<float>(rows * C);
    float* u = c.arena.alloc<float>(rows * I);
    QA_TRY(linear(c, zq, rows, h->linear_pre, x));
    for (int i = 0; i < 2; ++i) QA_TRY(vocos(c, h->down[i], x, t1, u, B, T, C, nullptr, 0));
    c.tap("prenet.down", x, rows * C);
    QA_TRY(vocos(c, h->backbone, x, t1, u, B, T, C, cond, h->ada.N));
    c.tap("prenet.backbone", x, rows * C);
    float* px = zq;  // z_q is dead: reuse it for the prenet output [B, T, latent]
    QA_TRY(linear(c, x, rows, h->linear_out, px));
    if (!c.dry) QA_TRY(launch_add_rowvec(px, dvec, B, T, Ld, c.stream));
    c.tap("prenet.out", px, rows * Ld);
    // ---- wave generator
    int ch = sp.gen_channels, Tc = T;
    float* s_in = c.arena.alloc<float>((size_t)B * Tc * ch);  // snake(conv0(x)): the only form block 0 consumes
    {
        ConvOpt o;
        o.pad_left = 3; o.pad_right = 3; o.act = ACT_SNAKE; o.alpha = h->blocks[0].a_in;
        QA_TRY(conv(c, px, Ld, B, T, h->gen_in, s_in, ch, T, o));
    }
    for (size_t bi = 0; bi < h->blocks.size(); ++bi) {
        const GenBlockW& g = h->blocks[bi];
        const int s = g.stride, co = g.c_out, To = Tc * s;
        const size_t n = (size_t)B * To * co;
        float* raw0 = c.arena.alloc<float>(n);
        float* raw1 = c.arena.alloc<float>(n);
        float* snk = c.arena.alloc<float>(n);
        float* act = c.arena.alloc<float>(n);
        // ConvTranspose1d: phase phi writes rows q * s + phi (row stride s * co) of the raw tensor and of its Snake (unit 0's alpha)
        for (int phi = 0; phi < s; ++phi) {
            ConvOpt o;
            o.pad_left = g.pad_left[phi];
            o.pad_right = g.phase[phi].ksize - 1 - g.pad_left[phi];
            o.y2 = snk + (size_t)phi * co; o.alpha2 = g.unit[0].a1; o.ldy2 = (int64_t)s * co;
            QA_TRY(conv(c, s_in, g.c_in, B, Tc, g.phase[phi], raw0 + (size_t)phi * co, (int64_t)s * co, Tc, o));
        }
        float *cur = raw0, *nxt = raw1;
        const bool last_block = bi + 1 == h->blocks.size();
        const float* a_next_block = last_block ? h->a_final : h->blocks[bi + 1].a_in;
        for (int j = 0; j < 3; ++j) {
            const UnitW& un = g.unit[j];
            ConvOpt o7;  // Snake (input, already applied) -> dilated k7 -> Snake (epilogue)
            o7.pad_left = 3 * un.dilation; o7.pad_right = 3 * un.dilation; o7.dilation = un.dilation;
            o7.act = ACT_SNAKE; o7.alpha = un.a2;
            QA_TRY(conv(c, snk, co, B, To, un.c7, act, co, To, o7));
            ConvOpt o1;  // k1 + skip; the sum leaves raw (for the next skip) and activated (for the next convolution)
            o1.res = cur; o1.ldr = co;
            if (j < 2) {
                o1.y2 = snk; o1.alpha2 = g.unit[j + 1].a1; o1.ldy2 = co;
                QA_TRY(conv(c, act, co, B, To, un.c1, nxt, co, To, o1));
            } else {  // the next consumer (next block's ConvTranspose1d / the output conv) only reads the activated sum
                o1.post_act = ACT_SNAKE; o1.alpha = a_next_block;
                QA_TRY(conv(c, act, co, B, To, un.c1, nxt, co, To, o1));
            }
            std::swap(cur, nxt);
        }
        c.tap("gen.block" + std::to_string(bi), cur, (int64_t)n);  // NOTE: the last unit's tensor is stored ACTIVATED (see above)
        s_in = cur;
        ch = co;
        Tc = To;
    }
    {
        ConvOpt o;
        o.pad_left = 3; o.pad_right = 3; o.act = ACT_TANH;
        QA_TRY(conv(c, s_in, ch, B, Tc, h->gen_out, wav_out, 1, Tc, o));
    }
    return QA_OK;
}

int ensure_ws(qa_bicodec* h, size_t bytes) {
    if (bytes <= h->ws_cap) return QA_OK;
    if (h->ws) QA_HIP(hipFree(h->ws));
    h->ws = nullptr;
    h->ws_cap = 0;
    const size_t cap = bytes + bytes / 8;
    QA_HIP(hipMalloc(reinterpret_cast<void**>(&h->ws), cap));
    h->ws_cap = cap;
    return QA_OK;
}

}  // namespace

extern "C" {

int qa_bicodec_create(qa_bicodec** out, const qa_bicodec_spec* spec, const qa_tensor* tensors, int64_t n_tensors, int device) {
    if (!out || !spec || !tensors) {
        set_error("qa_bicodec_create: null argument");
        return QA_ERR_INVALID;
    }
    *out = nullptr;
    QA_HIP(hipSetDevice(device));
    std::unique_ptr<qa_bicodec> h(new qa_bicodec());
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

void qa_bicodec_destroy(qa_bicodec* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    h->store.release();
    if (h->ws) (void)hipFree(h->ws);
    delete h;
}

int64_t qa_bicodec_hop(const qa_bicodec* h) { return h ? h->hop : QA_ERR_INVALID; }

int qa_bicodec_detokenize(qa_bicodec* h, const int64_t* semantic_tokens, const int64_t* global_tokens, int64_t B, int64_t T, float* wav_out,
                          void* stream) {
    if (!h || !semantic_tokens || !global_tokens || !wav_out) {
        set_error("qa_bicodec_detokenize: null argument");
        return QA_ERR_INVALID;
    }
    QA_REQUIRE(B > 0 && T > 0, "qa_bicodec_detokenize: tokens are [%lld, %lld]", (long long)B, (long long)T);
    QA_REQUIRE(B * T * (int64_t)h->hop * 32 < (1LL << 31), "qa_bicodec_detokenize: batch too large (split it)");
    QA_HIP(hipSetDevice(h->device));
    Ctx& c = h->ctx;
    c.stream = static_cast<hipStream_t>(stream);
    c.dry = true;
    c.arena.begin(nullptr, 0);
    QA_TRY(detokenize_graph(h, c, (const long long*)semantic_tokens, (const long long*)global_tokens, (int)B, (int)T, wav_out));
    QA_TRY(ensure_ws(h, c.arena.peak()));
    c.dry = false;
    c.taps.clear();
    c.arena.begin(h->ws, h->ws_cap);
    return detokenize_graph(h, c, (const long long*)semantic_tokens, (const long long*)global_tokens, (int)B, (int)T, wav_out);
}

int qa_bicodec_enable_taps(qa_bicodec* h, int on) {
    if (!h) {
        set_error("qa_bicodec_enable_taps: null handle");
        return QA_ERR_INVALID;
    }
    h->ctx.capture = on != 0;
    return QA_OK;
}

int64_t qa_bicodec_tap(qa_bicodec* h, const char* name, float* dst, int64_t cap, void* stream) {
    if (!h || !name) {
        set_error("qa_bicodec_tap: null argument");
        return QA_ERR_INVALID;
    }
    auto it = h->ctx.taps.find(name);
    if (it == h->ctx.taps.end()) {
        set_error("qa_bicodec_tap: no intermediate named '%s' in the last call", name);
        return QA_ERR_MISSING;
    }
    if (dst) {
        if (cap < it->second.numel) {
            set_error("qa_bicodec_tap: '%s' has %lld elements, capacity %lld", name, (long long)it->second.numel, (long long)cap);
            return QA_ERR_INVALID;
        }
        QA_HIP(hipMemcpyAsync(dst, it->second.ptr, sizeof(float) * it->second.numel, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    }
    return it->second.numel;
}

}  // extern "C"
