// knobs.h - every tuning / A-B switch of libquarkaudio_hip in ONE table (INTEGRATION.md lists the same rows).
//
// A knob is an integer.  Its initial value comes from the environment variable of the same name (read once, at the first
// use of any knob); qa_set_knob() changes it at run time, so tests and A/B sessions flip a switch inside one process
// instead of re-spawning with another environment.  Kernels never read knobs: only the host-side launch code does, at
// launch (or handle-creation) time.
#pragma once
#include <cstdint>

namespace qa {

#define QA_KNOB_TABLE(X)                                                                                                           \
    X(SERIAL, "QA_SERIAL", 0, "1: no internal stream concurrency (every kernel alone on the device; = qa_set_serial)")           \
    X(GEMM_CFG, "QA_GEMM_CFG", -1, "force the conv_gemm tile: 0 = 128x32, 1 = 128x64, 2 = 128x128, 3 = 64x128, 4 = 64x64 (-1: cost model)")             \
    X(GEMM_BK16, "QA_GEMM_BK16", 1 << 30, "largest K that takes the BK = 16 K-chunk variant")                                     \
    X(GEMM_BK16_MIN_TILES, "QA_GEMM_BK16_MIN_TILES", 384, "fewest tiles of a launch that take BK = 16")                          \
    X(GEMM_LINEAR, "QA_GEMM_LINEAR", 1, "table-free K loop for ksize-1 layers")                                                  \
    X(GEMM_XCD, "QA_GEMM_XCD", 1, "XCD-aware tile order")                                                                        \
    X(GEMM_PANEL, "QA_GEMM_PANEL", 8, "conv_gemm tile order: column panels of this many tiles, row tiles fastest inside a panel (0: column tiles fastest over the whole row; 8: +4 % on N >= 4096 shapes, +1.2 % on H-Codec 2.0)") \
    X(ATT_DEBUG, "QA_ATT_DEBUG", 0, "attention_kernel debug bits: 1 always rescale, 2 extra barrier per tile, 4 wait for the prefetch at once") \
    X(DWCONV_STRIP, "QA_DWCONV_STRIP", 1, "depthwise conv (+ LayerNorm) as row strips with the tap window in registers (0: one wave per output row; same bits)") \
    X(SEANET_FUSED, "QA_SEANET_FUSED", 1, "fused conv0 + first SEANet residual block")                                           \
    X(MIMI_ROPE_WINDOW, "QA_MIMI_ROPE_WINDOW", 8192, "mimi streaming: positions covered by the RoPE table before the rolling window takes over (tests shrink it)") \
    X(LSTM_GRAPH, "QA_LSTM_GRAPH", 1, "replay the T step launches of an LSTM call from a cached hipGraph")                       \
    X(LSTM_PERSISTENT, "QA_LSTM_PERSISTENT", -1, "persistent recurrence kernel: -1 auto (d >= 1536), 0 off, 1 on for every supported width") \
    X(LSTM_XCD, "QA_LSTM_XCD", 1, "XCD-local LSTM recurrence for d = 512 / 768 (one launch, W_hh in the registers of every XCD's 32 CUs, sequences dealt to the XCDs, 32-member step barrier per XCD): 0 off (the per-step kernels), 1 agent-scope hand-off forms, 2 XCD-local forms (h stores that stay in the XCD's L2; H-Codec 1.0: 42.7 against 43.3 ms)") \
    X(LSTM_TEAM, "QA_LSTM_TEAM", 1, "team recurrence for d = 1024 (one launch: 4 teams of 64 workgroups, W_hh resident in registers, 8 sequences per team, agent-scope hand-offs; H-Codec 1.5 decoder: 134.1 -> 128.3 ms per step, profiles/r04_lstm_team_ab.txt): 0 = the per-step kernels") \
    X(LSTM_SPIN_LIMIT, "QA_LSTM_SPIN_LIMIT", 1 << 21, "persistent recurrence: polls of a barrier word before the barrier is declared broken") \
    X(LSTM_FAULT, "QA_LSTM_FAULT", 0, "1 (tests): the persistent kernel's barrier waits for a workgroup that does not exist, like a starved launch") \
    X(LM_GRAPH, "QA_LM_GRAPH", 0, "1: replay one captured decode step per token")                                                \
    X(LM_MLP_FUSED, "QA_LM_MLP_FUSED", 1, "decode step: gate/up + SwiGLU + down of 16 activation columns per workgroup in one launch emitting K-slice partials, summed by a reduce launch (0: separate gate/up and down launches)") \
    X(LM_PF, "QA_LM_PF", 7, "decode step: cross-launch L2 weight prefetch planes (bit mask; lm_decode.h PfArgs): 1 qkv launch -> o_proj weights, 2 o_proj launch -> fused-MLP weights, 4 MLP launch -> next layer's qkv weights (last layer: the head slice), 8 head launch -> layer 0 qkv weights") \
    X(LM_ROWSPLIT, "QA_LM_ROWSPLIT", 3, "decode step at 9 .. 16 sequences: two 8-row groups instead of one 16-row group in the qkv launch (bit 1) and the fused-MLP launch (bit 2)") \
    X(LM_CHAINS, "QA_LM_CHAINS", 0, "generate: number of concurrent chains on internal streams (0: ceil(B / 64) - one chain serves up to 64 sequences, two row groups of 32 per launch; a count that would put more than 64 sequences into a chain is raised)")

enum Knob {
#define QA_KNOB_ENUM(id, name, def, doc) K_##id,
    QA_KNOB_TABLE(QA_KNOB_ENUM)
#undef QA_KNOB_ENUM
        K_COUNT
};

long long knob(Knob k);
void knob_set(Knob k, long long v);

}  // namespace qa
