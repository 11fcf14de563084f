// kernels.h - host-side launchers of every HIP kernel in libquarkaudio_hip.
#pragma once
#include <algorithm>

#include "common.h"

namespace qa {

// ew.hip
// pad_left < 0: the non-causal split of SConv1d (left = pad_total - pad_total / 2); causal SConv1d passes ksize - 1
int launch_conv_in(const float* x, const float* w_kc, const float* bias, float* y, int B, int T, int Cout, int ksize,
                   hipStream_t s, int pad_left = -1);
// seanet_front.hip: conv0 + SEANetResnetBlock + the ELU in front of the strided conv, one launch, `a` written once
bool seanet_front_supported(int C, int hid, int L);
int launch_seanet_front(const float* wav, const float* w0, const float* b0, const float* w3, const float* b3, const float* wsc,
                        const float* bsc, const float* wpw, const float* bpw, float* a, int B, int L, int C, int hid, int causal,
                        hipStream_t s);
int launch_rmsnorm(const float* x, const float* w, float* y, long long rows, int C, float eps, hipStream_t s);
int launch_layernorm(const float* x, const float* w, const float* b, float* y, long long rows, int C, float eps,
                     hipStream_t s);
// pad_left < 0: "same" (ksize / 2 each side); the causal Conv1d of vq/conv.py:44-47 passes ksize - 1
int launch_dwconv(const float* x, const float* w_kc, const float* bias, const float* lnw, const float* lnb, float* y,
                  int B, int T, int C, int ksize, float eps, hipStream_t s, int pad_left = -1);
size_t groupnorm_scratch_bytes(int B, int T, int G);
int launch_groupnorm(const float* x, const float* w, const float* bias, float* y, double* scratch, int B, int T, int C,
                     int G, float eps, int swish, hipStream_t s);
int launch_rope(float* qkv, const float* cos_sin, int B, int N, int H, int hd, long long ld, int pos0, hipStream_t s,
                int interleaved = 0);
int launch_align(const float* sem, int B, int T, int D, float thr, int max_tokens, int* seg, int* start, int* len,
                 int* nseg, int* gmax, hipStream_t s);
int launch_agg_build(const float* feats, const int* seg, const int* start, const int* len, const int* nseg, const float* qemb,
                     float* out, int B, int T, int G, int D, hipStream_t s);
int launch_agg_gather(const float* x, const int* start, const int* len, const int* nseg, float* out, int B, int T, int G,
                      int D, hipStream_t s);
int launch_codes_inject(const long long* idx, const int* len, long long* dst, int B, int T, int G, int Q, int K,
                        hipStream_t s);
int launch_adaptive_frames(const long long* codes, int B, int Q, int G, int K, int* totals, int* tmax, hipStream_t s);
int launch_deaggregate(const long long* codes, const long long* len_codes, long long* out, int B, int Q, int G, int T, int K,
                       hipStream_t s);
int launch_to_channel_last(const float* x, long long sb, long long sc, long long st, float* y, int B, int C, int T,
                           hipStream_t s);
int launch_codes_to_bqn(const long long* src, long long* dst, int B, int N, int Q, hipStream_t s);
int launch_codes_from_bqn(const long long* src, long long* dst, int B, int N, int Q, hipStream_t s);
int launch_stft_post(const float* ri, float* out, long long rows, int nb, int ldi, int ldo, hipStream_t s);
int launch_istft_spec(const float* y, float* S, long long rows, int nb, int ldy, int ldS, hipStream_t s);
int launch_istft_ola(const float* frames, const float* win, float* out, int B, int T, int n_fft, int hop, hipStream_t s);

int launch_codes_check(const long long* codes, long long n, long long limit, unsigned long long* bad, hipStream_t s);
int launch_codes_count(const long long* codes, long long n, long long lo, long long limit, unsigned long long* bad, hipStream_t s);
int launch_resample(const float* wav, const float* taps, float* out, int B, long long T, long long T_out, int orig, int nw, int width,
                    int ktaps, hipStream_t s);

// attention.hip : softmax(Q K^T * scale) V over a fused [B*N, 3*H*hd] QKV buffer (RoPE already applied)
//   causal = 0: full attention over the N keys of the same batch item (codec transformers)
//   causal = 1: key j visible to query i iff j <= i + (n_keys - n_q) (LM prefill / decode over a KV cache)
//   gate [B, H, n_q] + relbias [H, 2R+1] (optional): score(i, j) += gate[b,h,i] * relbias[h][clamp(j - i, -R, R) + R]
//   (WavLM gated relative position bias)
int launch_attention(const float* q, long long ldq, const float* k, const float* v, long long ldkv, float* out,
                     long long ldo, int B, int n_q, int n_keys, long long kv_batch_stride, int H, int hd, float scale,
                     int causal, hipStream_t s, const float* gate = nullptr, const float* relbias = nullptr, int R = 0,
                     int context = 0, int q_pos0 = 0, int ring_end = 0);
// RingKVCache.complete() write (mimi/transformer.py:243-250): rows t = 0..T-1 of k / v (row stride ld, batch stride T * ld) go to
// slot (pos0 + t) % cap of the caches [B, cap, d]
int launch_ring_append(const float* k, const float* v, long long ld, float* kc, float* vc, int B, int T, int d, int cap, int pos0,
                       hipStream_t s);

// lstm.hip : one nn.LSTM layer (batch_first, zero initial state) given the precomputed input projection
//   xw [B, T, 4d] = x W_ih^T + b_ih + b_hh with the 4d axis permuted to (unit, gate) order,
//   w_hh [4d, d] rows permuted the same way.  h_out [B, T, d].  c_state [B, d] scratch.
// eager = true: the T step kernels are launched one by one on `s` (no hipGraph replay, no persistent kernel) - for CU-masked streams,
// whose mask a replayed graph is not known to inherit
int launch_lstm(const float* xw, const float* w_hh_ug, float* h_out, float* c_state, int B, int T, int d,
                hipStream_t s, bool eager = false);
// persistent-recurrence bookkeeping for the model graphs: launches so far on `dev`; wait for `s` and report (and clear) a barrier
// time-out; make this thread's next launch_lstm calls take the per-step kernels
int lstm_call_begin(int dev, void** ticket);                  // open a model-graph call: own error word + launch count (thread-local)
int lstm_call_end(void* ticket, hipStream_t s, bool* failed);  // syncs `s` only if a recurrence of the call is still in flight
void lstm_call_note_sync();                                    // the graph synchronised the call's stream itself: collect now
void lstm_force_per_step(bool on);

// rvq.hip
// scratch: rvq_scratch_floats(n_vec, K, D) floats of device workspace (residuals, distance products, norms of one chunk)
size_t rvq_scratch_floats(long long n_vec, int K, int D);
int launch_rvq_search(const float* x, long long n_vec, const float* codebooks, const float* e2, int Q, int K, int D,
                      long long* indices, float* quantized, long long ldq, float* scratch, hipStream_t s);
int launch_rvq_norms(const float* codebooks, float* e2, int QK, int D, hipStream_t s);
int launch_rvq_lookup(const long long* indices, long long n_vec, const float* codebooks, int Q, int K, int D, float* out,
                      long long ldo, hipStream_t s);

}  // namespace qa
