// attention.hip - fp32 flash-style attention on the f32 matrix cores.
//
// Reference semantics: softmax(Q K^T * head_dim^-0.5, fp32) V, no mask for the codec transformers
// (QuarkAudio-HCodec/HCodec-1.0/vq/encoder_modules/transformer.py:158-180) and a causal mask over a KV cache for the
// UniSE Llama layers (QuarkAudio-UniSE/model/llm/llm.py:182-211).  RoPE has already been applied to q / k.
//
// Causal windows and the streaming ring (mimi/transformer.py:212-281,403-413; encoder_modules/transformer.py:437-447):
//   causal = 1            key j visible to query i iff j <= i + off                       (off = n_keys - n_q)
//   context > 0           ... and (i + off) - j < context   (StreamingMultiheadAttention `context`, Transformer `left_context`)
//   ring_end > 0          the keys are the slots of a RingKVCache of capacity n_keys whose write cursor stands at ring_end
//                         (= tokens seen so far, this chunk included): slot s holds position p(s) as RingKVCache.complete()
//                         computes it (-1 = never written), query i sits at q_pos0 + i, visible iff p >= 0 and
//                         0 <= pos_q - p < context.  Slots overwritten by later tokens of the same chunk are gone, as in the
//                         reference (it writes the whole chunk before it attends).
//
// One wave64 owns 32 queries and walks the keys 32 at a time; both products run TRANSPOSED on
// v_mfma_f32_32x32x2_f32 so that every softmax statistic is per lane (no cross-lane row reductions):
//   S^T[key, q] = sum_d K[key, d] Q[q, d]     A = K tile (LDS, ds_read_b128), B = Q (registers)
//   O^T[d,  q] += sum_key V[key, d] P[q, key] A = V tile (LDS),               B = P = exp(S - m) (registers)
// The 32x32 accumulator layout gives lane (q = lane & 31, h = lane >> 5) the keys (r&3) + 8*(r>>2) + 4*h for r = 0..15,
// which is exactly the k-slot order used for the second product, so P never leaves the registers.
#include "kernels.h"

#ifdef QA_ATT_TIMING  // tuning builds only (tools/variants.py -DQA_ATT_TIMING=1, read by tools/att_timing.py): shader cycles per phase of the key-tile loop
__device__ unsigned long long g_qa_att_timing[8];  // [0] barriers + LDS stores (incl. the wait for the prefetched tile), [1] S = K Q^T, [2] softmax, [3] O += V P, [4] whole kernel, [5] waves, [6] tiles
#define QA_ATT_TICK(i)                                      \
    {                                                       \
        const long long now_ = __builtin_readcyclecounter(); \
        tacc[i] += now_ - tlast;                            \
        tlast = now_;                                       \
    }
extern "C" int qa_debug_att_timing(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qa_att_timing), sizeof(g_qa_att_timing)) != hipSuccess) return -1;  // out[8]
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_qa_att_timing), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define QA_ATT_TICK(i)
#endif

namespace qa {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// BIAS: WavLM's gated relative position bias (transformers WavLMAttention.forward): score(i, j) += gate[b, head, i] *
// relbias[head][clamp(j - i, -R, R) + R]; the bucket function saturates below R, so the clamp is exact.
template <int HD, bool BIAS>
__global__ __launch_bounds__(256, 2) void attention_kernel(const float* __restrict__ q, long long ldq,
                                                        const float* __restrict__ k, const float* __restrict__ v,
                                                        long long ldkv, long long kv_bstride, float* __restrict__ out,
                                                        long long ldo, int n_q, int n_keys, float scale, int causal,
                                                        const float* __restrict__ gate, const float* __restrict__ relbias, int R,
                                                        int context, int q_pos0, int ring_end, int dbg_arg) {
#if defined(QA_ATT_DBG) && QA_ATT_DBG == 0
    constexpr int dbg = 0;
    (void)dbg_arg;
#else
    const int dbg = dbg_arg;
#endif
    constexpr int LD = HD + 4;
    constexpr int DT = HD / 32;
    constexpr int NG = HD / 8;
    __shared__ __attribute__((aligned(16))) float sK[32 * LD];
    __shared__ __attribute__((aligned(16))) float sV[32 * LD];

    // `wave` through readfirstlane: every wave-level test below (tile skips, mask tests) is then a SCALAR branch.  As a VGPR value
    // hipcc lowers them to exec-masked regions, and exec-masked VMEM next to register-staged prefetches is where ROCm 7.2 mis-tracks
    // outstanding loads (see fetch()).
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hh = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int q_blk0 = blockIdx.x * 128;
    const int qi = q_blk0 + wave * 32 + ql;
    const int off = n_keys - n_q;  // causal: key j visible iff j <= qi + off

    // Q fragment: lane (q, h) keeps d = 8g + 4h + e  ->  qreg[4g + e].  The softmax runs in base 2: scale * log2(e) is folded into
    // Q once, so that a score needs no multiply and an exponential is the single v_exp_f32 instruction (ocml's expf is ~10
    // instructions, and a vector instruction issued beside the other waves' MFMAs costs ~30 cycles, conv_gemm.hip)
    constexpr float LOG2E = 1.4426950408889634f;
    const float qs = scale * LOG2E;
    float qreg[HD / 2];
    {
        const int qrow = qi < n_q ? qi : n_q - 1;
        const float* qp = q + ((long long)b * n_q + qrow) * ldq + head * HD + 4 * hh;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const float4 t = *reinterpret_cast<const float4*>(qp + 8 * g);
            qreg[4 * g + 0] = t.x * qs; qreg[4 * g + 1] = t.y * qs; qreg[4 * g + 2] = t.z * qs; qreg[4 * g + 3] = t.w * qs;
        }
    }

    float gate_q = 0.f;
    const float* rb = nullptr;
    if (BIAS) {
        gate_q = LOG2E * gate[((long long)b * gridDim.y + head) * n_q + (qi < n_q ? qi : n_q - 1)];
        rb = relbias + (long long)head * (2 * R + 1) + R;
    }

    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const float* kb = k + (long long)b * kv_bstride + head * HD;
    const float* vb = v + (long long)b * kv_bstride + head * HD;

    const bool ring = ring_end > 0;
    const bool lin_causal = causal && !ring;
    int last_key = n_keys - 1, first_key = 0;
    if (lin_causal) {
        const int q_last = min(q_blk0 + 127, n_q - 1);
        last_key = min(last_key, q_last + off);
        if (context > 0) first_key = max(0, q_blk0 + off - context + 1);
    }
    const int n_tiles = last_key / 32 + 1, kt0 = first_key / 32;
    const int wave_last_key = lin_causal ? min(n_keys - 1, min(q_blk0 + wave * 32 + 31, n_q - 1) + off) : n_keys - 1;
    const int wave_first_key = (lin_causal && context > 0) ? max(0, q_blk0 + wave * 32 + off - context + 1) : 0;
    const int ring_idx = ring ? ring_end % n_keys : 0;  // RingKVCache.complete(): end_index

    // K / V tiles go global -> registers -> LDS; the loads of tile kt + 1 are issued right after tile kt is in LDS and stay in
    // flight under its 64 MFMAs (the first version loaded synchronously: one exposed L2 / HBM round trip per 32 keys)
    // Rows past n_keys re-read the last valid row instead of being predicated off: their scores are masked to -inf below (need_mask
    // is set on any tile that reaches n_keys) and their probability is exactly 0, so a finite stand-in row changes nothing - and
    // every load of the kernel is UNCONDITIONAL.  The round-2 form (`if (key < n_keys) load; else zeros`) put the prefetch into an
    // exec-masked region; with it hipcc (ROCm 7.2) re-used the staging registers while the loads were still in flight for
    // HD = 64 at n_keys = 1500: half of all outputs differed from run to run (tools/diag_attention.py on a build of that form,
    // profiles/r03_attention_determinism.txt).  Caught by the at-size parity test of H-Codec 2.0 (tests/test_at_size_gpu.py).
    constexpr int NLD = HD / 32;  // float4 per thread per operand: 32 keys x HD floats over 256 threads
    // native vector type, NOT float4: with float4 staging arrays hipcc funnels the unconditional loads through ONE temporary register
    // quad into AGPRs for HD >= 96 (global_load; s_waitcnt vmcnt(0); v_accvgpr_write - eight serialized round trips per tile: 110 -> 177 us
    // per launch at HD = 128); ext_vector_type values stay in VGPRs and the loads stay in flight
    f32x4 kreg[NLD], vreg[NLD];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + 256 * j;
            const int row = i / (HD / 4), c4 = (i % (HD / 4)) * 4;
            const int key = min(kt * 32 + row, n_keys - 1);
            kreg[j] = *reinterpret_cast<const f32x4*>(kb + (long long)key * ldkv + c4);
            vreg[j] = *reinterpret_cast<const f32x4*>(vb + (long long)key * ldkv + c4);
        }
    };
#ifdef QA_ATT_TIMING
    long long tacc[4] = {0, 0, 0, 0};
    long long tlast = __builtin_readcyclecounter();
    const long long tbegin = tlast;
    long long n_done = 0;
#endif
    fetch(kt0);
    for (int kt = kt0; kt < n_tiles; ++kt) {
        QA_ATT_TICK(3)  // (the tail of the previous tile's PV phase; the first time: the Q prologue, negligible)
        __syncthreads();  // the previous tile is no longer read
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + 256 * j;
            const int row = i / (HD / 4), c4 = (i % (HD / 4)) * 4;
            *reinterpret_cast<f32x4*>(sK + row * LD + c4) = kreg[j];
            *reinterpret_cast<f32x4*>(sV + row * LD + c4) = vreg[j];
        }
        __syncthreads();
        if (kt + 1 < n_tiles) fetch(kt + 1);
        if (dbg & 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        QA_ATT_TICK(0)
        // wave-uniform skips: the whole tile is masked for this wave, or the wave owns no query at all (the last query block of a
        // sequence that is not a multiple of 128: at N = 283 three of the four waves of block 3 would multiply clamped rows)
        if (!(dbg & 8) && (kt * 32 > wave_last_key || kt * 32 + 31 < wave_first_key || q_blk0 + wave * 32 >= n_q)) continue;

        // S^T = K Q^T
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float* kp = sK + ql * LD + 4 * hh;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(kp + 8 * g);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qreg[4 * g + 0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qreg[4 * g + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qreg[4 * g + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qreg[4 * g + 3], s, 0, 0, 0);
        }
        QA_ATT_TICK(1)
        // online softmax in base 2 (per lane = per query; the two halves of the wave hold interleaved key groups).  Masks are
        // evaluated only on tiles that can contain a hidden key for some query of this wave (wave-uniform test).
        const int q_first = q_blk0 + wave * 32, q_last = q_first + 31;
        const bool need_mask = (dbg & 16) || ring || kt * 32 + 31 >= n_keys ||
                               (lin_causal && (kt * 32 + 31 > q_first + off || (context > 0 && kt * 32 < q_last + off - context + 1)));
        float tmax = -INFINITY;
        if (BIAS || need_mask) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                bool ok = key < n_keys;
                if (ring) {
                    const int delta = key - ring_idx;
                    const int pos = key >= ring_end ? -1 : (delta <= 0 ? ring_end + delta : ring_end + delta - n_keys);
                    const int dq = q_pos0 + qi - pos;
                    ok = ok && pos >= 0 && dq >= 0 && dq < context;
                } else if (causal) {
                    ok = ok && key <= qi + off && (context <= 0 || qi + off - key < context);
                }
                float sc = s[r];
                if (BIAS) sc += gate_q * rb[max(-R, min(R, key - qi))];
                s[r] = ok ? sc : -INFINITY;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);  // m_run = -inf -> 0
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __builtin_amdgcn_exp2f(s[r] - m_use);
            psum += s[r];
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if ((dbg & 1) || __any(alpha != 1.f)) {  // the running maximum moved for some query of the wave: rescale (rare after the first tiles)
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        }
        QA_ATT_TICK(2)
        // O^T += V^T P^T ; k-slot (step st, half h) <-> key (st&3) + 8*(st>>2) + 4*h
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int key = (st & 3) + 8 * (st >> 2) + 4 * hh;
            const float* vp = sV + key * LD + ql;
#pragma unroll
            for (int t = 0; t < DT; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32 * t], s[st], o[t], 0, 0, 0);
        }
        if (dbg & 2) __syncthreads();
#ifdef QA_ATT_TIMING
        ++n_done;
#endif
    }
#ifdef QA_ATT_TIMING
    QA_ATT_TICK(3)
    if (lane == 0) {
        for (int i = 0; i < 4; ++i) atomicAdd(&g_qa_att_timing[i], (unsigned long long)tacc[i]);
        atomicAdd(&g_qa_att_timing[4], (unsigned long long)(tlast - tbegin));
        atomicAdd(&g_qa_att_timing[5], 1ULL);
        atomicAdd(&g_qa_att_timing[6], (unsigned long long)n_done);
    }
#endif

    if (qi < n_q) {
        // a query with no visible key (possible in the ring mode: a chunk as long as the ring overwrites everything its first query
        // could see, and the slot at the write cursor is masked) yields 0, as torch >= 2.5's scaled_dot_product_attention does for
        // a fully masked row - the convention the oracle and the reference-generated golden were produced under
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        float* op = out + ((long long)b * n_q + qi) * ldo + head * HD + 4 * hh;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float4 w;
                w.x = o[t][4 * c + 0] * inv; w.y = o[t][4 * c + 1] * inv;
                w.z = o[t][4 * c + 2] * inv; w.w = o[t][4 * c + 3] * inv;
                *reinterpret_cast<float4*>(op + 32 * t + 8 * c) = w;
            }
    }
}

// gate [B, H, n_q] and relbias [H, 2R+1] (both optional, together): gated relative position bias, see attention_kernel
int launch_attention(const float* q, long long ldq, const float* k, const float* v, long long ldkv, float* out,
                     long long ldo, int B, int n_q, int n_keys, long long kv_batch_stride, int H, int hd, float scale,
                     int causal, hipStream_t s, const float* gate, const float* relbias, int R, int context, int q_pos0,
                     int ring_end) {
    QA_REQUIRE(n_q > 0 && n_keys > 0 && (!causal || ring_end > 0 || n_keys >= n_q), "attention: n_q=%d n_keys=%d", n_q, n_keys);
    QA_REQUIRE(context >= 0 && (context == 0 || causal) && (ring_end <= 0 || (causal && context > 0 && !gate)),
               "attention: a context window needs causal=1; the ring mode needs causal=1 and context > 0");
    QA_REQUIRE((ldq % 4) == 0 && (ldkv % 4) == 0 && (ldo % 4) == 0, "attention: strides must be multiples of 4");
    QA_REQUIRE((gate == nullptr) == (relbias == nullptr) && (!gate || (R >= 0 && !causal && n_q == n_keys)),
               "attention: gate and relbias come together, for non-causal self-attention");
    dim3 grid((unsigned)ceil_div(n_q, 128), H, B);
    const int dbg = (int)knob(K_ATT_DEBUG);
#define QA_ATT(HD)                                                                                                          \
    if (gate)                                                                                                                \
        hipLaunchKernelGGL((attention_kernel<HD, true>), grid, dim3(256), 0, s, q, ldq, k, v, ldkv, kv_batch_stride, out, ldo, n_q, \
                           n_keys, scale, causal, gate, relbias, R, context, q_pos0, ring_end, dbg);                             \
    else                                                                                                                     \
        hipLaunchKernelGGL((attention_kernel<HD, false>), grid, dim3(256), 0, s, q, ldq, k, v, ldkv, kv_batch_stride, out, ldo, n_q, \
                           n_keys, scale, causal, nullptr, nullptr, 0, context, q_pos0, ring_end, dbg)
    switch (hd) {
        case 32: QA_ATT(32); break;
        case 64: QA_ATT(64); break;
        case 96: QA_ATT(96); break;
        case 128: QA_ATT(128); break;
        default: qa::set_error("attention: head_dim=%d unsupported (32/64/96/128)", hd); return QA_ERR_UNSUPPORTED;
    }
#undef QA_ATT
    QA_LAUNCH_CHECK();
    return QA_OK;
}

}  // namespace qa

// test / diagnostic hook (not part of the public header): the attention kernel alone on caller-provided buffers
extern "C" int qa_debug_attention(const float* q, long long ldq, const float* k, const float* v, long long ldkv, float* out, long long ldo,
                                  int B, int n_q, int n_keys, long long kv_bstride, int H, int hd, float scale, int causal, void* stream) {
    return qa::launch_attention(q, ldq, k, v, ldkv, out, ldo, B, n_q, n_keys, kv_bstride, H, hd, scale, causal,
                                static_cast<hipStream_t>(stream));
}
