"""TEST INFRASTRUCTURE ONLY - token-stream golden vectors for the UniSE AR-LM, produced by the reference's OWN
`LLM_SFT.generate` (QuarkAudio-UniSE/model/llm/llm_sft.py:93-195, do_sample=False as in model/model.py:173) run in this
container through oracle/ref_llm_shim.py on seeded weights / inputs (unified_audio_amd/synth.py generators, so the GPU box
regenerates the same tensors from the seeds).  Stored per case: the reference's global / semantic ids, and the top-2 logit
gap of every step (from the oracle teacher-forced on the reference's stream) so that a GPU implementation can be audited
with the near-tie protocol.

Run in the build container:  python -m oracle.gen_golden_lm
"""
from __future__ import annotations

import os

import numpy as np
import torch

from oracle import llm_ref as L
from oracle import ref_llm_shim as S

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
SMALL = dict(hidden=256, n_layers=2, n_heads=4, global_size=96, semantic_size=160, feats_dim=64, num_tasks=3)
CASES = {  # name -> (spec kwargs or None for the UniSE shape, weight seed, task, B, n_mix, n_enroll, S, G)
    "lm_small_se": (SMALL, 21, "se", 3, 9, 0, 12, 5),
    "lm_small_tse": (SMALL, 21, "tse", 3, 9, 7, 12, 5),
    "lm_small_rtse": (SMALL, 22, "rtse", 2, 6, 11, 7, 32),
    "lm_unise_se": (None, 33, "se", 4, 50, 0, 50, 32),        # 12 x 512, vocab 12291 (conf/config.yaml:131-146)
    "lm_unise_tse": (None, 34, "tse", 2, 30, 40, 30, 32),
    # BASELINE configs[2] / configs[3] AT SIZE: 16 segments x 5 s (prompt 252, KV 535) and the per-GPU TSE share of 8
    # segments with a 250-frame enrollment (prompt 503, KV 786): the shapes bench.py's tokens/s are quoted on
    "lm_config3_se_b16": (None, 35, "se", 16, 250, 0, 250, 32),
    "lm_config4_tse_b8": (None, 36, "tse", 8, 250, 250, 250, 32),
    # RANGE STRESS: synth.stress_lm_state_dict - output head x 0.05 (near-degenerate top-2 gaps: the stored `gaps` are what a
    # GPU stream is audited with), q / k projections x 3 (softmax rows dominated by one key)
    "lm_small_stress": (SMALL, 23, "tse", 3, 9, 7, 24, 8, "stress"),
}


def case_tensors(name):
    kw, seed, task, B, n_mix, n_enr, S_len, G = CASES[name][:8]
    spec = L.LMSpec(**kw) if kw else L.SPEC_UNISE
    sd = L.lm_state_dict(seed, spec)
    if "stress" in CASES[name][8:]:
        from unified_audio_amd.synth import stress_lm_state_dict

        sd = stress_lm_state_dict(sd)
    mix = L.synth_feats(seed + 100, B, n_mix, spec.feats_dim)
    enr = L.synth_feats(seed + 200, B, n_enr, spec.feats_dim) if n_enr else None
    return spec, sd, task, mix, enr, S_len, G


def reference_generate(spec, sd, task, mix, enr, S_len, G, seed=None, **kw):
    model = S.load_state(S.load_reference_llm(spec), sd)
    mel = torch.zeros(mix.shape[0], S_len, 80)  # only mix_mel.size(1) is read (llm_sft.py:108)
    if seed is not None:
        torch.manual_seed(seed)  # after construction: the random init of the module consumes the global RNG too
    with torch.no_grad():
        return model.generate(task, None if enr is None else mel, enr, mel, mix, global_length=G, **kw)


def main(names=None):
    os.makedirs(GOLDEN, exist_ok=True)
    for name in (names or CASES):
        spec, sd, task, mix, enr, S_len, G = case_tensors(name)
        g_ref, s_ref = reference_generate(spec, sd, task, mix, enr, S_len, G, do_sample=False)
        g_o, s_o, toks_o, _ = L.generate(sd, task, enr, mix, S_len, G, spec)
        assert torch.equal(g_o, g_ref) and torch.equal(s_o, s_ref), name  # the pin itself (also a test)
        _, _, _, gaps = L.generate(sd, task, enr, mix, S_len, G, spec, forced=toks_o)
        np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), global_ids=g_ref.numpy().astype(np.int16),
                            semantic_ids=s_ref.numpy().astype(np.int16), discarded=toks_o[:, G].numpy().astype(np.int16),
                            gaps=gaps.numpy().astype(np.float32))
        print(name, tuple(g_ref.shape), tuple(s_ref.shape), "min gap %.2e" % float(gaps.min()))


if __name__ == "__main__":
    import sys

    main(sys.argv[1:] or None)
