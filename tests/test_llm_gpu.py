"""UniSE AR-LM generate through the C-ABI against the CPU oracle (greedy token streams, integer output)."""
import pytest
import torch

from oracle import llm_ref as L

pytestmark = pytest.mark.gpu

SMALL = L.LMSpec(hidden=256, n_layers=2, n_heads=4, global_size=96, semantic_size=160, feats_dim=64, num_tasks=3)


def _model(spec, seed, device):
    import unified_audio_amd as qa

    sd = L.lm_state_dict(seed, spec)
    cfg = dict(global_size=spec.global_size, semantic_size=spec.semantic_size, hidden_size=spec.hidden,
               num_layers=spec.n_layers, num_attention_heads=spec.n_heads)
    lm = qa.LLM_SFT(num_tasks=spec.num_tasks, feats_dim=spec.feats_dim, llm_base_config=cfg, device=device)
    return sd, lm.load_state_dict({"dnn." + k: v for k, v in sd.items()})  # Lightning checkpoint prefix (model.py:82-91)


def _audit(sd, spec, task, enr, mix, S, G, gids, sids, tol=2e-4):
    """Integer output: the HIP stream must equal the oracle's wherever the oracle's decision is not a near-tie.  The
    oracle is re-run teacher-forced on the HIP tokens, so every step is judged on identical context."""
    B = mix.shape[0]
    g_o, s_o, toks_o, _ = L.generate(sd, task, enr, mix, S, G, spec)
    free_match = min((gids == g_o).float().mean().item(), (sids == s_o).float().mean().item())
    # rebuild the raw stream the HIP path fed back: G kept globals, one discarded global, S semantics
    forced = torch.cat([gids + spec.global_offset, toks_o[:, G:G + 1], sids + spec.semantic_offset], dim=1)
    _, _, toks_f, gaps = L.generate(sd, task, enr, mix, S, G, spec, forced=forced)
    keep = torch.ones_like(forced, dtype=torch.bool)
    keep[:, G] = False  # the discarded 33rd global token is not observable
    wrong = (toks_f != forced) & keep
    assert not (wrong & (gaps > tol)).any(), f"{int((wrong & (gaps > tol)).sum())} decisive mismatches"
    return free_match, int(wrong.sum())


@pytest.mark.parametrize("task,use_enroll", [("se", False), ("tse", True)])
def test_small_lm_generate_matches_oracle(qa_lib, gpu_device, task, use_enroll):
    sd, lm = _model(SMALL, 21, gpu_device)
    B, Nm, Ne, S, G = 3, 9, 7, 12, 5
    mix = L.synth_feats(1, B, Nm, SMALL.feats_dim)
    enr = L.synth_feats(2, B, Ne, SMALL.feats_dim) if use_enroll else None
    mel = torch.zeros(B, S, 80)
    gids, sids = lm.generate(task, mel if use_enroll else None, None if enr is None else enr.to(gpu_device), mel,
                             mix.to(gpu_device), global_length=G, do_sample=False)
    assert gids.shape == (B, G) and sids.shape == (B, S) and gids.dtype == torch.int64
    free_match, near_ties = _audit(sd, SMALL, task, enr, mix, S, G, gids.cpu(), sids.cpu())
    print("free-running agreement", free_match, "near-tie flips", near_ties)
    assert near_ties > 0 or free_match == 1.0  # a stream can only leave the oracle's at an audited near-tie


def test_unise_lm_full_size_generate(qa_lib, gpu_device):
    """The real UniSE LM shape (12 x 512, vocab 12291), SE prompt of 2 + 50 frames, 33 + 50 steps, B = 4."""
    spec = L.SPEC_UNISE
    sd, lm = _model(spec, 33, gpu_device)
    B, Nm, S = 4, 50, 50
    mix = L.synth_feats(5, B, Nm)
    gids, sids = lm.generate("se", None, None, torch.zeros(B, S, 80), mix.to(gpu_device), do_sample=False)
    assert int(gids.min()) >= 0 and int(gids.max()) < 4096 and int(sids.min()) >= 0 and int(sids.max()) < 8192
    free_match, near_ties = _audit(sd, spec, "se", None, mix, S, 32, gids.cpu(), sids.cpu())
    print("free-running agreement", free_match, "near-tie flips", near_ties)
    assert near_ties > 0 or free_match == 1.0


@pytest.mark.parametrize("name", ["lm_small_se", "lm_small_tse", "lm_small_rtse", "lm_unise_se", "lm_unise_tse"])
def test_generate_matches_reference_token_goldens(qa_lib, gpu_device, name):
    """Token streams produced by the reference's OWN LLM_SFT.generate (oracle/gen_golden_lm.py): the HIP stream must be
    identical up to the first step whose top-2 logit gap (stored with the golden) is below fp32 noise; whatever follows such
    a step is audited teacher-forced against the oracle like every other stream."""
    import os

    import numpy as np

    from oracle import gen_golden_lm as GG

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    spec, sd, task, mix, enr, S, G = GG.case_tensors(name)
    _, lm = _model(spec, GG.CASES[name][1], gpu_device)
    mel = torch.zeros(mix.shape[0], S, 80)
    gids, sids = lm.generate(task, None if enr is None else mel, None if enr is None else enr.to(gpu_device), mel,
                             mix.to(gpu_device), global_length=G, do_sample=False)
    got = torch.cat([gids.cpu(), sids.cpu()], dim=1).numpy()
    want = np.concatenate([g["global_ids"], g["semantic_ids"]], axis=1).astype(np.int64)
    gaps = np.delete(g["gaps"], G, axis=1)  # drop the discarded 33rd global step: columns now line up with got / want
    for b in range(got.shape[0]):
        diff = np.nonzero(got[b] != want[b])[0]
        if diff.size:
            assert gaps[b, diff[0]] <= 2e-4, f"sequence {b} leaves the reference stream at step {diff[0]} (gap {gaps[b, diff[0]]:.2e})"
    _audit(sd, spec, task, enr, mix, S, G, gids.cpu(), sids.cpu())


def test_generate_argument_errors(qa_lib, gpu_device):
    import unified_audio_amd as qa

    sd, lm = _model(SMALL, 21, gpu_device)
    mix = L.synth_feats(1, 1, 4, SMALL.feats_dim).to(gpu_device)
    mel = torch.zeros(1, 4, 80)
    with pytest.raises(KeyError):
        lm.generate("asr", None, None, mel, mix, do_sample=False)
    with pytest.raises(qa.QuarkAudioError):
        lm.generate("se", None, None, mel, mix, temperature=1.5, do_sample=False)  # llm.py:278 assert
    with pytest.raises(qa.QuarkAudioError):
        lm.generate("se", None, None, mel, mix, do_sample=True)
