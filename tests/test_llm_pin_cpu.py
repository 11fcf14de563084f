"""Pins oracle/llm_ref.py (generate loop, prompt assembly, range masks, sample_logits) to the reference's OWN
LLM_SFT (QuarkAudio-UniSE/model/llm/llm_sft.py:93-195, llm.py:253-288): live through oracle/ref_llm_shim.py where
/root/reference is mounted, and everywhere through the token-stream goldens that reference run produced
(oracle/gen_golden_lm.py -> tests/golden/lm_*.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import gen_golden_lm as G
from oracle import llm_ref as L
from oracle import ref_llm_shim as S

HERE = os.path.dirname(os.path.abspath(__file__))
live = pytest.mark.skipif(not S.reference_available(), reason="/root/reference is only mounted in the build container")
TINY = L.LMSpec(hidden=64, n_layers=2, n_heads=2, global_size=40, semantic_size=50, feats_dim=32, num_tasks=3)


@pytest.mark.parametrize("name", list(G.CASES))
def test_oracle_reproduces_reference_token_goldens(name):
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    spec, sd, task, mix, enr, S_len, Gl = G.case_tensors(name)
    gi, si, toks, gaps = L.generate(sd, task, enr, mix, S_len, Gl, spec)
    assert np.array_equal(gi.numpy(), g["global_ids"].astype(np.int64))
    assert np.array_equal(si.numpy(), g["semantic_ids"].astype(np.int64))
    assert np.array_equal(toks[:, Gl].numpy(), g["discarded"].astype(np.int64))
    np.testing.assert_allclose(gaps.numpy(), g["gaps"], rtol=0, atol=5e-4)


@live
@pytest.mark.parametrize("task,n_enr", [("se", 0), ("tse", 5), ("rtse", 3)])
def test_greedy_generate_equals_reference(task, n_enr):
    sd = L.lm_state_dict(5, TINY)
    mix = L.synth_feats(1, 2, 6, TINY.feats_dim)
    enr = L.synth_feats(2, 2, n_enr, TINY.feats_dim) if n_enr else None
    g_r, s_r = G.reference_generate(TINY, sd, task, mix, enr, 6, 4, do_sample=False)
    g, s, _, _ = L.generate(sd, task, enr, mix, 6, 4, TINY)
    assert torch.equal(g, g_r) and torch.equal(s, s_r)
    assert g_r.dtype == torch.int64 and g_r.shape == (2, 4) and s_r.shape == (2, 6)


@live
@pytest.mark.parametrize("top_k,top_p,temperature", [(50, 0.95, 0.8), (0, 0.9, 1.0), (5, 1.0, 0.5), (0, 1.0, 1.0)])
def test_sampled_generate_equals_reference_under_the_same_seed(top_k, top_p, temperature):
    """do_sample=True (the signature default, llm_sft.py:106): same torch ops in the same order consume the same RNG
    stream, so the restatement must reproduce the reference's sampled tokens exactly."""
    sd = L.lm_state_dict(6, TINY)
    mix = L.synth_feats(3, 3, 5, TINY.feats_dim)
    g_r, s_r = G.reference_generate(TINY, sd, "se", mix, None, 9, 6, seed=1234, do_sample=True, top_k=top_k, top_p=top_p,
                                    temperature=temperature)
    torch.manual_seed(1234)
    g, s, _, _ = L.generate(sd, "se", None, mix, 9, 6, TINY, do_sample=True, top_k=top_k, top_p=top_p, temperature=temperature)
    assert torch.equal(g, g_r) and torch.equal(s, s_r)


@live
def test_sample_logits_equals_reference_incl_ties_and_inplace_mutation():
    model = S.load_reference_llm(TINY)
    gen = torch.Generator().manual_seed(0)
    logits = torch.randn(6, TINY.vocab, generator=gen) * 3
    logits[:, :3] = float("-inf")
    logits[0, 10:14] = logits[0].max() + 1.0  # 4-way tie at the top
    logits[1, 3:] = 0.25                       # everything tied: top-k keeps all (strict '<', llm.py:263)
    for kw in (dict(top_k=50, top_p=0.95, temperature=0.8), dict(top_k=3, top_p=0.5, temperature=0.3),
               dict(top_k=0, top_p=0.7, temperature=1.0)):
        for do_sample in (False, True):
            a = logits.clone()
            torch.manual_seed(7)
            want = model.sample_logits(a, do_sample=do_sample, **kw)
            torch.manual_seed(7)
            got = L.sample_logits(logits, do_sample=do_sample, **kw)
            assert torch.equal(got, want), (kw, do_sample)
            # the reference filters `logits` in place (llm.py:264,275); its surviving set is the oracle's support
            keep_ref = torch.isfinite(a)
            keep = L.sampling_distribution(logits, **kw) > 0
            assert torch.equal(keep, keep_ref)
    with pytest.raises(AssertionError):
        L.sample_logits(logits, temperature=1.5)
    with pytest.raises(AssertionError):
        model.sample_logits(logits.clone(), temperature=1.5)
