/*
 * quarkaudio.h - C-ABI of libquarkaudio_hip.so: the MI355X (gfx950) hot path of alibaba/unified-audio
 * (QuarkAudio): H-Codec encode -> RVQ -> decode, and the UniSE decoder-only AR-LM generate loop.
 *
 * The reference is 100 % Python and has no FFI of its own; the drop-in boundary is the tensor-in /
 * tensor-out method pair a maintainer would rebind (see INTEGRATION.md for the ctypes stub):
 *
 *   qa_hcodec_encode   <-> Codec.encode(x, feat)            QuarkAudio-HCodec/HCodec-1.0/vq/codec.py:166-175
 *   qa_hcodec_decode   <-> Codec.decode(ac, sc)             QuarkAudio-HCodec/HCodec-1.0/vq/codec.py:178-187
 *   qa_hcodec_create   <-> Codec(...) + load_state_dict     QuarkAudio-HCodec/HCodec-1.0/audio_tokenizer.py:23-26
 *   qa_rvq_search      <-> ResidualVQ.forward (eval)        call sites vq/codec.py:171-172 (algorithm: vq/core_vq.py:223-231,394-404)
 *   qa_rvq_lookup      <-> ResidualVQ.get_output_from_indices  call sites vq/codec.py:183-184 (vq/core_vq.py:406-412)
 *   qa_lm_create       <-> LLM_SFT(...) + load_state_dict   QuarkAudio-UniSE/model/llm/llm_sft.py:13-33, model/model.py:82-91
 *   qa_lm_generate     <-> LLM_SFT.generate(...)            QuarkAudio-UniSE/model/llm/llm_sft.py:93-195
 *   qa_bicodec_detokenize <-> BiCodec.detokenize(...)       QuarkAudio-UniSE/model/bicodec/bicodec.py:182-199
 *
 * Conventions
 *   - every function returns 0 on success or a negative qa_status; nothing throws across the ABI;
 *     qa_last_error() returns a thread-local, NUL-terminated description of the last failure.
 *   - all data pointers are DEVICE pointers owned by the caller (e.g. torch tensor.data_ptr()),
 *     contiguous unless strides are passed, except the weight table of *_create which is HOST memory
 *     in the reference's own state_dict layout (the library folds weight-norm, re-lays filters out for
 *     its kernels and keeps its own device copy).
 *   - work is enqueued on `stream` (a hipStream_t; NULL = the default stream) and is asynchronous;
 *     a handle owns its workspaces and must not be used from two streams / threads at once.
 *     EXCEPTION (qa_hcodec_encode / _decode and their adaptive forms): a call that launched one of the in-launch
 *     LSTM recurrences (knobs QA_LSTM_XCD - the default for d = 512 / 768 -, QA_LSTM_TEAM, QA_LSTM_PERSISTENT) waits
 *     for `stream` on the host before it returns, because it must read that launch's own error word and, if a bounded
 *     barrier spin ran out (device shared with another persistent kernel), re-run itself on the per-step kernels: such
 *     a call is host-synchronous and cannot be issued under a caller's stream capture.  QA_LSTM_XCD=0 QA_LSTM_TEAM=0
 *     QA_LSTM_PERSISTENT=0 restores fully asynchronous calls (launch-per-step recurrence).  The error word and the
 *     launch count are per CALL (a ticket taken by the calling thread), so handles that share a device from
 *     different threads never collect each other's failure.
 *   - there is no CPU fallback: on a machine without a gfx950 device every compute entry point fails.
 */
#ifndef QUARKAUDIO_H_
#define QUARKAUDIO_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QA_VERSION 103 /* 0.1.2 */

typedef enum qa_status {
    QA_OK = 0,
    QA_ERR_INVALID = -1,   /* bad argument / shape (the reference raises AssertionError / RuntimeError) */
    QA_ERR_HIP = -2,       /* a HIP runtime call failed (no device, launch failure, out of memory) */
    QA_ERR_MISSING = -3,   /* a tensor the architecture needs is not in the weight table (KeyError) */
    QA_ERR_UNSUPPORTED = -4
} qa_status;

/* One named fp32 tensor of a state_dict.  `data` is HOST memory, row-major, `numel` elements. */
typedef struct qa_tensor {
    const char* name;
    const float* data;
    int64_t numel;
} qa_tensor;

/* Architecture constants of an H-Codec model (hard-coded in the reference: vq/codec.py:30-136). */
typedef struct qa_hcodec_spec {
    int32_t n_filters;        /* 32            codec.py:33  */
    int32_t n_ratios;         /* 4 */
    int32_t ratios[8];        /* encoder order 2,4,5,8  (codec.py:33 reversed by seanet.py:114) */
    int32_t dimension;        /* 512 */
    int32_t enc_heads;        /* 8   seanet.py:166 */
    int32_t enc_layers;       /* 2   seanet.py:167 */
    int32_t sem_in;           /* 768 codec.py:122 */
    int32_t sem_ch;           /* 768 codec.py:123 */
    int32_t n_sem_strides;    /* 2 */
    int32_t sem_strides[4];   /* 2,1 codec.py:126 */
    int32_t code_dim;         /* 512 */
    int32_t codebook_size;    /* 1024 */
    int32_t num_quantizers;   /* 4 */
    int32_t dec_dim;          /* 768 codec.py:44 */
    int32_t dec_inter;        /* 2304 */
    int32_t dec_heads;        /* 8 */
    int32_t dec_layers;       /* 2 */
    int32_t convnext_layers;  /* 12 */
    int32_t n_fft;            /* 1280 */
    int32_t hop;              /* 320 */
    int32_t gn_groups;        /* 32 */
    /* H-Codec 1.5 adaptive frame rate (QuarkAudio-HCodec/HCodec-1.5/conf/config_adaptive_v3.yaml:65-111); 0 = H-Codec 1.0 */
    int32_t adaptive;
    int32_t agg_layers;       /* 32  aggregators.*.num_layers (d_model = code_dim) */
    int32_t agg_heads;        /* 8 */
    int32_t agg_ff;           /* 2048 */
    int32_t bt_layers;        /* 32  transformer_kwargs.num_layers (bottleneck, d_model = 2*code_dim) */
    int32_t bt_heads;         /* 8 */
    int32_t bt_ff;            /* 2048 */
    int32_t max_tokens_per_group; /* 8 */
    float threshold;          /* 0.6 manual_threshold */
    /* H-Codec 2.0 (QuarkAudio-HCodec/HCodec-2.0/conf/large_12.5hz_config.yaml); version 0 / 10 = SEANet family (1.0, 1.5) */
    int32_t version;          /* 20 = STFT-domain ConvNeXt encoder, repeat-interleave decoder embed */
    int32_t enc_dim;          /* 1536 */
    int32_t enc_inter;        /* 4608 */
    int32_t enc_convnext_layers; /* 24 */
    int32_t frame_stride;     /* 4 = int(50 / target_frame_rate) */
    int32_t tr_inter_cap;     /* 4096: transformer MLP width = min(4*d, cap); 0 = 4*d */
    /* causal variant (all versions; 2.0: `causal` of encoder_config / decoder_config, codec_encoder.py:23, codec_decoder.py:25):
     * every SConv1d pads
     * (k_eff - stride, extra) instead of splitting it (encoder_modules/conv.py:203-206), vq/conv.py's Conv1d /
     * ConvTranspose1d pad (k - 1, 0) (vq/conv.py:44-47,76-79) and both Transformers apply the tril mask
     * (encoder_modules/transformer.py:470-475).  vq/codec.py:31 ships causal=False. */
    int32_t causal;
    /* H-Codec 1.5: `causal` / `context_frames` of the two aggregators and `causal` / `context` of the bottleneck transformer
     * (config_adaptive_v3.yaml:84,87,93,96,103,105; the YAML ships causal: false, where the context is ignored,
     * mimi/transformer.py:403-413).  causal = 1: key j visible to query i iff 0 <= i - j < context (context 0 = unbounded). */
    int32_t agg_causal, agg_context, bt_causal, bt_context;
} qa_hcodec_spec;

typedef struct qa_hcodec qa_hcodec;

int qa_version(void);
const char* qa_last_error(void);
/* number of visible HIP devices (0 if none / runtime unavailable); never fails */
int qa_device_count(void);

/* ---- H-Codec ------------------------------------------------------------------------------------ */

int qa_hcodec_create(qa_hcodec** out, const qa_hcodec_spec* spec, const qa_tensor* tensors, int64_t n_tensors,
                     int device);
void qa_hcodec_destroy(qa_hcodec* h);

/* Codec.encode.  wav: [B, T] fp32, T a multiple of the encoder hop (2*prod(ratios); HCodecTokenizer.pad_wav
 * guarantees it, audio_tokenizer.py:50-53).  feat: fp32 [B, sem_in, N50] addressed through element strides
 * (so the [B, N50, sem_in] tensor the SSL model returns can be passed without the transpose copy the reference
 * makes at audio_tokenizer.py:59).  Outputs: int64 [B, num_quantizers, N25] each, contiguous. */
int qa_hcodec_encode(qa_hcodec* h, const float* wav, int64_t B, int64_t T,
                     const float* feat, int64_t feat_stride_b, int64_t feat_stride_c, int64_t feat_stride_t,
                     int64_t n_feat_frames, int64_t* acoustic_codes, int64_t* semantic_codes, void* stream);

/* Codec.decode.  codes: int64 [B, num_quantizers, N]; wav_out: fp32 [B, 2*N*hop]. */
int qa_hcodec_decode(qa_hcodec* h, const int64_t* acoustic_codes, const int64_t* semantic_codes, int64_t B,
                     int64_t N, float* wav_out, void* stream);

/* H-Codec 1.5 (spec.adaptive != 0): Codec.encode / Codec.decode of QuarkAudio-HCodec/HCodec-1.5/vq/codec_adaptive.py:150-199.
 * The number of groups G is data dependent (the reference syncs the host too: modeling_flexicodec_new.py:910), so
 *   - encode writes int64 [B, nq, G] length-injected codes (code' = (len-1)*codebook_size + code) compactly into buffers of
 *     capacity B*nq*N25 elements and returns G through *n_groups (the call synchronises `stream` once);
 *   - qa_hcodec_adaptive_frames returns max_b sum_g len[b,g] (= N25 of the clip) for a batch of codes, so the caller can size
 *     wav_out = [B, frames * 2 * hop] before qa_hcodec_decode_adaptive.
 * threshold: the per-call similarity threshold of Codec.encode(..., threshold=t) (codec_adaptive.py:150-158): 0 = the model's
 *   manual_threshold (spec.threshold), otherwise in (0, 1]. */
int qa_hcodec_encode_adaptive(qa_hcodec* h, const float* wav, int64_t B, int64_t T,
                              const float* feat, int64_t feat_stride_b, int64_t feat_stride_c, int64_t feat_stride_t,
                              int64_t n_feat_frames, int64_t* acoustic_codes, int64_t* semantic_codes, int64_t* n_groups,
                              float threshold, void* stream);
int qa_hcodec_adaptive_frames(qa_hcodec* h, const int64_t* semantic_codes, int64_t B, int64_t G, int64_t* frames, void* stream);
int qa_hcodec_decode_adaptive(qa_hcodec* h, const int64_t* acoustic_codes, const int64_t* semantic_codes, int64_t B,
                              int64_t G, int64_t frames, float* wav_out, void* stream);

/* Test hook: when enabled, encode/decode snapshot their named intermediates (costs copies; off by default). */
int qa_hcodec_enable_taps(qa_hcodec* h, int on);
/* Test hook: copy the named snapshot of the LAST encode/decode (still in the handle's workspace) into
 * `dst` (device, fp32, capacity `cap` elements).  Returns the element count or a negative status.  Layout is
 * the library's: time-major, channel-last ([B, frames, channels]).  Names: see DESIGN.md "taps". */
int64_t qa_hcodec_tap(qa_hcodec* h, const char* name, float* dst, int64_t cap, void* stream);

/* ---- kernel-level entry points (used by the parity tests and available to integrators) ------------- */

/* Residual nearest-codebook search.  x: [n_vec, D]; codebooks: [Q, K, D]; indices out: int64 [n_vec, Q];
 * quantized_out (nullable): [n_vec, D] = sum_q E_q[idx_q].  Ties resolve to the lowest index. */
int qa_rvq_search(const float* x, int64_t n_vec, const float* codebooks, int32_t Q, int32_t K, int32_t D,
                  int64_t* indices, float* quantized_out, void* stream);
/* indices: int64 [n_vec, Q] -> out [n_vec, D] = sum_q E_q[idx_q].  idx == -1 is a DROPPED code and contributes a zero vector, as in the
 * third-party ResidualVQ.get_output_from_indices the reference calls (vq/codec.py:183-184; upstream masks -1, the quantize-dropout
 * convention).  Other out-of-range indices are NOT detected on device (clamped for memory safety); the caller guarantees
 * -1 <= idx < K - qa_codes_check_async(codes, n, -1, K, ...) is the check (the reference would raise IndexError). */
int qa_rvq_lookup(const int64_t* indices, int64_t n_vec, const float* codebooks, int32_t Q, int32_t K, int32_t D,
                  float* out, void* stream);

/* Range check of integer codes before a decode (the reference's F.embedding raises IndexError on the host, or trips a device-side
 * assert on a GPU): *bad = number of entries of codes[0..n) outside [0, limit).  One tiny kernel + one 4-byte copy; synchronises
 * `stream`.  The decode entry points themselves never synchronise and clamp indices for memory safety. */
int qa_codes_check(const int64_t* codes, int64_t n, int64_t limit, int64_t* bad, void* stream);
/* The same check without a host synchronisation: *bad_count_dev (DEVICE int64, zeroed by the caller) += number of entries of
 * codes[0..n) outside [lo, limit).  The caller reads the counter after whatever synchronisation it performs anyway (the Python mirror
 * reads it behind the decode call's own: one host round trip per decode instead of two). */
int qa_codes_check_async(const int64_t* codes, int64_t n, int64_t lo, int64_t limit, int64_t* bad_count_dev, void* stream);

/* torchaudio.transforms.Resample(orig_freq, new_freq) with its defaults (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99),
 * the 48 kHz -> 16 kHz step in front of HuBERT in H-Codec 2.0 (HCodec-2.0/audio_tokenizer.py:44,51).
 * qa_resample_length = ceil(new * T / orig) samples per clip; wav [B, T] -> out [B, qa_resample_length] (device). */
int64_t qa_resample_length(int64_t T, int32_t orig_freq, int32_t new_freq);
int qa_resample(const float* wav, int64_t B, int64_t T, int32_t orig_freq, int32_t new_freq, float* out, void* stream);

/* Implicit-GEMM Conv1d over channel-last activations (covers nn.Linear with ksize = 1):
 *   y[b, t, n] = post( res[b,t,n] + gamma[n] * act( bias[n] + sum_{j,c} pro(x[b, src(t,j), c]) * w[n, j, c] ) )
 * with src(t,j) = t*stride - pad_left + j resolved by pad_mode (0 zero, 1 reflect as SConv1d does).
 * Exposed for kernel-level parity tests.  See DESIGN.md for the full argument contract. */
typedef struct qa_conv_args {
    const float* x;        /* [B, T_in, C_in] */
    const float* w;        /* [N, ksize, C_in]  (library layout) */
    const float* bias;     /* [N] or NULL */
    const float* gamma;    /* [N] or NULL */
    const float* residual; /* [B, T_out, N] (row stride ldr) or NULL */
    const float* gate;     /* [B, T_out, N] (row stride ldg) or NULL: y = silu(gate) * (acc + bias) */
    float* y;              /* [B, T_out, N] with row stride ldy */
    int64_t B, T_in, C_in, T_out, N;
    int64_t ldx;           /* row stride of x in floats (>= C_in) */
    int64_t ldy, ldr, ldg;
    int32_t ksize, stride, pad_left, pad_right, pad_mode;
    int32_t prologue;      /* 0 none, 1 ELU applied to x on load */
    int32_t act;           /* 0 none, 1 ELU, 2 GELU(erf), 3 SiLU */
    int32_t post_act;      /* applied after the residual add: 0 none, 1 ELU */
    int32_t in_rep;        /* 0/1 none; r > 1: x is read as x.repeat_interleave(r) along frames (zero padding only) */
} qa_conv_args;
int qa_conv1d_cl(const qa_conv_args* args, void* stream);
/* RMSNorm (mode 1: transformer.py:77-96 of H-Codec 1.0, LlamaRMSNorm) / LayerNorm (mode 2: nn.LayerNorm, biased variance) over the last axis of
 * x [rows, C] (C % 4 == 0, C <= 2048); w [C], b [C] or NULL.  Exposed for kernel-level parity tests. */
int qa_rownorm(const float* x, const float* w, const float* b, float* y, int64_t rows, int32_t C, float eps, int32_t mode, void* stream);
/* Depthwise Conv1d over channel-last x [B, T, C] with zero padding (pad_left frames in front, ksize - 1 - pad_left behind; -1 = ksize / 2: the "same"
 * padding of vq/conv.py:33-56), optionally followed by LayerNorm over C (ConvNeXtBlock: dwconv k7 -> LayerNorm, vq/conv.py:200-203):
 *   y[b, t, c] = LN_c( bias[c] + sum_j w[j, c] * x[b, t + j - pad_left, c] ).  w in the library's [ksize, C] layout (the reference's [C, 1, ksize]
 * transposed); ln_w / ln_b [C] or both NULL.  C % 4 == 0, C <= 2048, ksize odd.  Exposed for kernel-level parity tests. */
int qa_dwconv_cl(const float* x, const float* w_kc, const float* bias, const float* ln_w, const float* ln_b, float* y, int64_t B, int64_t T, int32_t C,
                 int32_t ksize, int32_t pad_left, float eps, void* stream);

/* ---- host logic exposed for CPU tests (no device needed) ------------------------------------------------
 * SConv1d geometry of the reference (encoder_modules/conv.py:54-61,195-211, non-causal): for an input of L frames,
 * kernel k (dilation 1) and stride s it yields T_out = ceil(L/s) and the (left, right + extra) reflect paddings. */
int qa_sconv_geometry(int64_t L, int32_t ksize, int32_t stride, int64_t* T_out, int32_t* pad_left, int32_t* pad_right);
/* Source frame read by padded position r of an L-frame signal: index in [0, L) or -1 for "reads zero".  pad_mode 0 zero,
 * 1 reflect with the short-input rule of pad1d (encoder_modules/conv.py:79-96); max_pad = max(pad_left, pad_right). */
int64_t qa_resolve_frame(int64_t r, int64_t L, int32_t max_pad, int32_t pad_mode);

/* ---- measurement hook (bench.py) ------------------------------------------------------------------------
 * Between qa_profile_begin() and qa_profile_end() every implicit-GEMM launch is bracketed by HIP events recorded on
 * the stream it is launched on.  qa_profile_end fills out[cfg*4 + {0,1,2,3}] = {algorithmic FLOPs, elapsed ms, launches,
 * algorithmic bytes (input frames + weights + outputs + fused residual / gate reads, each once)} for the five tile
 * configurations cfg = 0 (128x32), 1 (128x64), 2 (128x128), 3 (64x128), 4 (64x64); n_out >= 20.  Not thread-safe; process-wide. */
int qa_profile_begin(void);
int qa_profile_end(double* out, int32_t n_out);
/* qa_profile_begin_ex(mask): bit 0 = the implicit-GEMM launches (= qa_profile_begin), bit 1 = the byte-bound kernels (norms, depthwise
 * conv, RoPE, ISTFT, RVQ look-up / pick, the fused SEANet front ...), each recorded with its ALGORITHMIC bytes (every input and output
 * element once).  qa_profile_end_hbm (call it before qa_profile_end) fills out[kind*3 + {0,1,2}] = {bytes, elapsed ms, launches} for
 * kind in [0, qa_profile_hbm_kinds()); n_out >= 3 * kinds; qa_profile_hbm_name(kind) names the kernel.  bytes / ms = achieved GB/s
 * against the HBM roofline (bench.py's `roofline_hbm`). */
int qa_profile_begin_ex(int32_t mask);
int qa_profile_hbm_kinds(void);
const char* qa_profile_hbm_name(int32_t kind);
int qa_profile_end_hbm(double* out, int32_t n_out);
/* qa_set_serial(1) (or QA_SERIAL=1 in the environment) collapses the library's internal streams onto the caller's, so that a
 * profiler sees every kernel alone on the device; results are bit-identical either way.  Process-wide. */
int qa_set_serial(int32_t on);

/* Diagnostics of the in-launch LSTM recurrences of `device` (tests): out[0] recurrences launched so far, out[1] model-graph calls with
 * one possibly still in flight (not yet behind a host synchronisation), out[2] launches that took the per-step kernels because ANOTHER
 * call's recurrence was in flight (the co-residency ticket: two handles driving one device never starve each other's grid barrier),
 * out[3] 1 once a barrier time-out has degraded the device to the per-step kernels. */
int qa_debug_lstm_stats(int32_t device, int64_t* out4);

/* ---- tuning knobs -------------------------------------------------------------------------------------------
 * Every A/B switch of the library is one row of a table (csrc/knobs.h; INTEGRATION.md lists them): an integer whose initial
 * value is the environment variable of the same name (e.g. QA_LSTM_PERSISTENT, QA_LM_MLP_FUSED) and which qa_set_knob()
 * overrides at run time.  Knobs are read by the host-side launch code at launch (QA_LM_MLP_FUSED: at qa_lm_create) time.  No knob changes results beyond fp32 summation order; none selects a CPU path.  Process-wide. */
int qa_knob_count(void);
int qa_knob_info(int32_t index, const char** name, int64_t* value, int64_t* default_value, const char** doc);
int qa_set_knob(const char* name, int64_t value);
int qa_get_knob(const char* name, int64_t* value);

/* ---- SSL front-end ---------------------------------------------------------------------------
 * HCodecTokenizer.extract_wav2vec2_features (QuarkAudio-HCodec/HCodec-1.0/audio_tokenizer.py:35-48, HCodec-1.5/audio_tokenizer.py:53-67):
 * and UniSE's Model.extract_semantic_features (QuarkAudio-UniSE/model/model.py:38-51, microsoft/wavlm-base-plus, no compression):
 * zero-pad `pad` samples each side, run the HuBERT / wav2vec 2.0 / WavLM model (transformers HubertModel / Wav2Vec2Model / WavLMModel: 7-layer
 * Conv1d feature extractor, feature projection, grouped positional convolution, post-LN or stable-LN encoder), average the
 * selected hidden states, compress sign * |x|^e.  Weights: the HF state_dict (keys feature_extractor.conv_layers.*,
 * feature_projection.*, encoder.pos_conv_embed.conv.{parametrizations.weight.original0/1 | weight_g/weight_v | weight}, ...). */
typedef struct qa_ssl_spec {
    int32_t n_conv;             /* 7 */
    int32_t conv_dim[8];        /* 512 x 7 */
    int32_t conv_kernel[8];     /* 10,3,3,3,3,2,2 */
    int32_t conv_stride[8];     /* 5,2,2,2,2,2,2 */
    int32_t conv_bias;          /* 0 HuBERT base; 1 wav2vec2-large / XLSR */
    int32_t feat_norm_layer;    /* 0 = "group" (GroupNorm on layer 0 only), 1 = "layer" (LayerNorm after every conv) */
    int32_t hidden;             /* 768 / 1024 */
    int32_t n_layers;           /* 12 / 24 */
    int32_t n_heads;            /* 12 / 16 */
    int32_t intermediate;       /* 3072 / 4096 */
    int32_t stable_layer_norm;  /* 0 post-LN (base), 1 pre-LN + final LN (large / XLSR) */
    int32_t pos_kernel;         /* 128 */
    int32_t pos_groups;         /* 16 */
    int32_t pad;                /* 160 (F.pad(wavs, (160, 160))) */
    int32_t n_select;           /* number of hidden states averaged; 0 = all n_layers + 1 (torch.stack(hidden_states).mean) */
    int32_t select[32];         /* their indices into hidden_states (1.5: 11, 14, 16) */
    float layer_norm_eps;       /* 1e-5 */
    float compress_exponent;    /* 0.3; <= 0: return the plain average (UniSE, QuarkAudio-UniSE/model/model.py:38-51) */
    int32_t rel_pos_buckets;    /* 0 HuBERT / wav2vec2; 320 WavLM: gated relative position bias (WavLMAttention), keys
                                   encoder.layers.0.attention.rel_attn_embed.weight, ...attention.gru_rel_pos_{linear,const} */
    int32_t rel_pos_max_distance; /* 800 */
} qa_ssl_spec;
typedef struct qa_ssl qa_ssl;
int qa_ssl_create(qa_ssl** out, const qa_ssl_spec* spec, const qa_tensor* tensors, int64_t n_tensors, int device);
void qa_ssl_destroy(qa_ssl* h);
/* frames the model yields for T samples (after padding); negative status when T is too short */
int64_t qa_ssl_frames(const qa_ssl* h, int64_t T);
/* wav float32 [B, T] (device) -> feats float32 [B, frames, hidden] (device, channel-last: what qa_hcodec_encode takes as `feat`
 * with strides (frames*hidden, 1, hidden)) */
int qa_ssl_forward(qa_ssl* h, const float* wav, int64_t B, int64_t T, float* feats, void* stream);

/* ---- mimi StreamingTransformer: causal / context windows and the streaming state --------------------------
 * Replaces StreamingTransformer (QuarkAudio-HCodec/HCodec-1.5/adaptive/model_blocks/mimi/transformer.py:605-698) in the
 * configuration H-Codec 1.5 instantiates it with (:722-736): positional_embedding "rope", norm "layer_norm", gating "none",
 * LayerScale, no biases.  Weights: `<prefix>.layers.N.{self_attn.in_proj_weight, self_attn.out_proj.weight, norm1.*, norm2.*,
 * linear1.weight, linear2.weight, layer_scale_1.scale, layer_scale_2.scale}`.
 *   qa_mimi_forward       forward() outside streaming: no mask unless causal; causal: key j visible to query i iff
 *                         0 <= i - j < context (context 0 = unbounded)                          (transformer.py:403-413)
 *   qa_mimi_stream_begin  `with model.streaming(B)` / streaming_forever(B): one RingKVCache of capacity `context` per layer
 *                         (:212-241,345-370); needs causal = 1 and context > 0 like the reference (:349-353,382)
 *   qa_mimi_stream_step   forward() inside streaming on a chunk x [B, T, d], T <= context: RoPE at the running offset, the
 *                         chunk's keys / values written to slots (offset + t) % context BEFORE the queries attend, positions
 *                         and validity of the slots as RingKVCache.complete() computes them (:243-281) - including its
 *                         treatment of the slot at the write cursor, which leaves context - 1 visible keys per query
 *   qa_mimi_stream_reset  reset_streaming(): offsets to zero, cache contents kept but invisible (:239-241)
 *   qa_mimi_stream_end    leaving the context manager (streaming.py:100-105)
 * x and y are fp32 [B, T, d_model] device buffers (y may alias x). */
typedef struct qa_mimi_spec {
    int32_t d_model;          /* 512 / 1024 */
    int32_t num_heads;        /* 8 */
    int32_t num_layers;       /* 32 */
    int32_t dim_feedforward;  /* 2048 */
    int32_t causal;           /* config_adaptive_v3.yaml ships false */
    int32_t context;          /* 16 */
} qa_mimi_spec;
typedef struct qa_mimi qa_mimi;
int qa_mimi_create(qa_mimi** out, const qa_mimi_spec* spec, const qa_tensor* tensors, int64_t n_tensors, const char* prefix,
                   int device);
void qa_mimi_destroy(qa_mimi* m);
int qa_mimi_forward(qa_mimi* m, const float* x, int64_t B, int64_t T, float* y, void* stream);
int qa_mimi_stream_begin(qa_mimi* m, int64_t B);
int qa_mimi_stream_step(qa_mimi* m, const float* x, int64_t T, float* y, void* stream);
int qa_mimi_stream_reset(qa_mimi* m);
int qa_mimi_stream_end(qa_mimi* m);
/* tokens seen since begin / reset; -1 outside streaming */
int64_t qa_mimi_stream_offset(const qa_mimi* m);

/* ---- BiCodec detokenizer ------------------------------------------------------------------------
 * BiCodec.detokenize(semantic_tokens, global_tokens) (QuarkAudio-UniSE/model/bicodec/bicodec.py:182-199), the stage
 * Model.test_step ends with (model/model.py:193,223): codebook look-up + out_project, FSQ look-up + speaker projection (d-vector),
 * the AdaLN-Vocos prenet, and the Snake / ConvTranspose1d / dilated-residual wave generator.  Shapes of the Spark-TTS BiCodec
 * checkpoint (its config.yaml is not part of the reference tree) are the defaults of unified_audio_amd.BiCodecSpec.
 * Weights: the `BiCodec.state_dict()` entries quantizer.codebook / quantizer.out_project, speaker_encoder.quantizer.project_out,
 * speaker_encoder.project, prenet.*, decoder.* (weight_g / weight_v or plain weight). */
typedef struct qa_bicodec_spec {
    int32_t latent_dim;        /* 1024  quantizer.input_dim = prenet in / out = d-vector width = decoder.input_channel */
    int32_t codebook_size;     /* 8192 */
    int32_t codebook_dim;      /* 8 */
    int32_t spk_latent_dim;    /* 128 */
    int32_t token_num;         /* 32 global tokens */
    int32_t n_levels;          /* 6 */
    int32_t levels[8];         /* 4,4,4,4,4,4 (FSQ) */
    int32_t vocos_dim;         /* 384 */
    int32_t vocos_inter;       /* 2048 */
    int32_t vocos_layers;      /* 12 */
    int32_t gen_channels;      /* 1536 */
    int32_t n_rates;           /* 4 */
    int32_t rates[8];          /* 8,5,4,2 */
    int32_t kernel_sizes[8];   /* 16,11,8,4 */
} qa_bicodec_spec;
typedef struct qa_bicodec qa_bicodec;
int qa_bicodec_create(qa_bicodec** out, const qa_bicodec_spec* spec, const qa_tensor* tensors, int64_t n_tensors, int device);
void qa_bicodec_destroy(qa_bicodec* h);
/* samples per semantic token = prod(rates) (320) */
int64_t qa_bicodec_hop(const qa_bicodec* h);
/* semantic_tokens int64 [B, T], global_tokens int64 [B, token_num] (the reference's [B, 1, token_num], contiguous) -> wav_out
 * fp32 [B, T * hop] (the reference returns [B, 1, T * hop]).  Out-of-range tokens are clamped for memory safety only: validate
 * with qa_codes_check where the reference's F.embedding would raise. */
int qa_bicodec_detokenize(qa_bicodec* h, const int64_t* semantic_tokens, const int64_t* global_tokens, int64_t B, int64_t T,
                          float* wav_out, void* stream);
/* test hooks, as qa_hcodec_enable_taps / qa_hcodec_tap: z_q, d_vector, prenet.down, prenet.backbone, prenet.out, gen.block{i} */
int qa_bicodec_enable_taps(qa_bicodec* h, int on);
int64_t qa_bicodec_tap(qa_bicodec* h, const char* name, float* dst, int64_t cap, void* stream);

/* ---- UniSE AR-LM ------------------------------------------------------------------------------------ */

typedef struct qa_lm_spec {
    int32_t hidden;        /* 512   QuarkAudio-UniSE/conf/config.yaml:131-146 */
    int32_t n_layers;      /* 12 */
    int32_t n_heads;       /* 8 */
    int32_t intermediate;  /* 2048 = 4*hidden (llm.py:69) */
    int32_t global_size;   /* 4096 */
    int32_t semantic_size; /* 8192 */
    int32_t feats_dim;     /* 768 */
    int32_t num_tasks;     /* 3 */
    float rope_theta;      /* 10000 */
    float rms_eps;         /* 1e-6 */
} qa_lm_spec;

typedef struct qa_lm qa_lm;

int qa_lm_create(qa_lm** out, const qa_lm_spec* spec, const qa_tensor* tensors, int64_t n_tensors, int device);
void qa_lm_destroy(qa_lm* lm);

/* LLM_SFT.generate with do_sample=False (the reference's test path, model/model.py:173): greedy decoding of
 * `global_length`+1 global tokens (last one discarded) then `semantic_length` semantic tokens.
 *   task        0 se, 1 tse, 2 rtse (config.yaml:132-136)
 *   enroll_feats [B, n_enroll, feats_dim] or NULL (SE prompt); mix_feats [B, n_mix, feats_dim]
 *   global_ids  int64 [B, global_length]; semantic_ids int64 [B, semantic_length]  (offsets already subtracted)
 * top_k / top_p / temperature are accepted for signature parity; with greedy decoding they cannot change the
 * argmax (llm.py:262-286) and are validated only (0 < temperature <= 1, llm.py:278). */
int qa_lm_generate(qa_lm* lm, int32_t task, const float* enroll_feats, int64_t n_enroll, const float* mix_feats,
                   int64_t n_mix, int64_t B, int32_t global_length, int32_t semantic_length, float temperature,
                   int32_t top_k, float top_p, int64_t* global_ids, int64_t* semantic_ids, void* stream);

/* LLM_SFT.generate with do_sample=True (the signature default, llm_sft.py:106): every step filters the logits of the active
 * vocabulary slice as CustomLlamaModel.sample_logits does (llm.py:253-288: top-k threshold with ties kept, nucleus filter on the
 * un-tempered logits that keeps the token crossing top_p, / temperature, softmax) and draws one token per sequence.  The
 * reference draws from torch's global generator; here the draw is a Philox4x32-10 uniform keyed by (seed, sequence, step), so the
 * same seed reproduces the same streams and the DISTRIBUTION equals the reference's (the streams cannot). */
int qa_lm_generate_sampled(qa_lm* lm, int32_t task, const float* enroll_feats, int64_t n_enroll, const float* mix_feats,
                           int64_t n_mix, int64_t B, int32_t global_length, int32_t semantic_length, float temperature,
                           int32_t top_k, float top_p, uint64_t seed, int64_t* global_ids, int64_t* semantic_ids, void* stream);

/* Kernel-level entry point of the sampler (parity / distribution tests): CustomLlamaModel.sample_logits (llm.py:253-288) on
 * logits [B, width] (row stride ld, device).  out_index int64 [B] (device).  do_sample = 0: arg-max (first maximum).
 * Synchronises `stream`. */
int qa_sample_logits(const float* logits, int64_t B, int64_t width, int64_t ld, int32_t top_k, float top_p, float temperature,
                     int32_t do_sample, uint64_t seed, int64_t* out_index, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QUARKAUDIO_H_ */
