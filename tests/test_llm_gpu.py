"""UniSE AR-LM generate through the C-ABI against the CPU oracle (greedy token streams, integer output)."""
import os

import pytest
import torch

from oracle import llm_ref as L

pytestmark = pytest.mark.gpu

SMALL = L.LMSpec(hidden=256, n_layers=2, n_heads=4, global_size=96, semantic_size=160, feats_dim=64, num_tasks=3)


def _model(spec, seed, device, sd=None):
    import unified_audio_amd as qa

    sd = sd if sd is not None else L.lm_state_dict(seed, spec)
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


def golden_stream_parity(name, device, audit=True, verbose=print):
    """One case of oracle/gen_golden_lm.py through the HIP path (also called by __graft_entry__.smoke() for the config-3 case)."""
    import os

    import numpy as np

    from oracle import gen_golden_lm as GG

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    spec, sd, task, mix, enr, S, G = GG.case_tensors(name)
    _, lm = _model(spec, GG.CASES[name][1], device, sd=sd)  # the case's own weights (the stress case transforms them)
    mel = torch.zeros(mix.shape[0], S, 80)
    gids, sids = lm.generate(task, None if enr is None else mel, None if enr is None else enr.to(device), mel,
                             mix.to(device), global_length=G, do_sample=False)
    got = torch.cat([gids.cpu(), sids.cpu()], dim=1).numpy()
    want = np.concatenate([g["global_ids"], g["semantic_ids"]], axis=1).astype(np.int64)
    gaps = np.delete(g["gaps"], G, axis=1)  # drop the discarded 33rd global step: columns now line up with got / want
    left = 0
    for b in range(got.shape[0]):
        diff = np.nonzero(got[b] != want[b])[0]
        if diff.size:
            left += 1
            assert gaps[b, diff[0]] <= 2e-4, f"sequence {b} leaves the reference stream at step {diff[0]} (gap {gaps[b, diff[0]]:.2e})"
    verbose(f"{name}: {got.shape[0]} sequences x {got.shape[1]} tokens, {left} left the reference stream at a near-tie, "
            f"smallest gap of the golden {float(g['gaps'].min()):.2e}")
    if audit or left:
        _audit(sd, spec, task, enr, mix, S, G, gids.cpu(), sids.cpu())
    return left


@pytest.mark.parametrize("name", ["lm_small_se", "lm_small_tse", "lm_small_rtse", "lm_unise_se", "lm_unise_tse", "lm_small_stress"])
def test_generate_matches_reference_token_goldens(qa_lib, gpu_device, name):
    """Token streams produced by the reference's OWN LLM_SFT.generate (oracle/gen_golden_lm.py): the HIP stream must be
    identical up to the first step whose top-2 logit gap (stored with the golden) is below fp32 noise; whatever follows such
    a step is audited teacher-forced against the oracle like every other stream."""
    golden_stream_parity(name, gpu_device)


@pytest.mark.parametrize("name", ["lm_config3_se_b16", "lm_config4_tse_b8"])
def test_generate_at_baseline_size_matches_reference_goldens(qa_lib, gpu_device, name):
    """BASELINE configs[2] / configs[3] AT SIZE - the shapes bench.py's tokens/s are quoted on: full UniSE width, 16 segments
    x 5 s (prompt 252, 33 + 250 steps, KV 535: three-way key split of the decode attention) and the per-GPU TSE share of 8
    segments with a 250-frame enrollment (prompt 503, KV 786: four-way split), against token streams of the reference's own
    LLM_SFT.generate.  The goldens' smallest top-2 gap is 4.6e-4 / 3.2e-4, above the 2e-4 near-tie bar: the streams must be
    IDENTICAL (the teacher-forced oracle audit only runs if one is not)."""
    left = golden_stream_parity(name, gpu_device, audit=False)
    assert left == 0


def test_spec_the_decode_step_cannot_tile_is_refused_with_the_reason(qa_lib, gpu_device):
    """r05: the round-1 per-op decode step (QA_LM_UNFUSED) is gone; a spec whose widths the fused step does not tile (hidden not a
    multiple of 256) is refused at load time with the reason instead of silently taking a slower path."""
    import unified_audio_amd as qa

    spec = L.LMSpec(hidden=128, n_layers=1, n_heads=2, global_size=64, semantic_size=96, feats_dim=64, num_tasks=3)
    with pytest.raises(qa.QuarkAudioError, match="hidden % 256"):
        _model(spec, 5, gpu_device)


def test_generate_argument_errors(qa_lib, gpu_device):
    import unified_audio_amd as qa

    sd, lm = _model(SMALL, 21, gpu_device)
    mix = L.synth_feats(1, 1, 4, SMALL.feats_dim).to(gpu_device)
    mel = torch.zeros(1, 4, 80)
    with pytest.raises(KeyError):
        lm.generate("asr", None, None, mel, mix, do_sample=False)
    with pytest.raises(qa.QuarkAudioError):
        lm.generate("se", None, None, mel, mix, temperature=1.5, do_sample=False)  # llm.py:278 assert


# ------------------------------------------------------------------------------------------- fused step / graph replay / sampling

def _gen(lm, spec, device, B, Nm, S, G, **kw):
    mix = L.synth_feats(41, B, Nm, spec.feats_dim)
    mel = torch.zeros(B, S, 80)
    g, s = lm.generate("se", None, None, mel, mix.to(device), global_length=G, **kw)
    torch.cuda.synchronize()
    return g.cpu(), s.cpu()


@pytest.mark.parametrize("B", [6, 40])
def test_generate_under_a_callers_stream_capture(qa_lib, gpu_device, knob, B):
    """batches above 32 sequences default to concurrent chains on INTERNAL streams that replay step graphs captured by the
    library - inside a caller's own stream capture that would fail where the plain launches work.  qa_lm_generate now asks
    hipStreamIsCapturing and, when the caller is capturing, issues everything as plain launches on the caller's stream (one chain, no
    capture of its own): the call can be recorded into a caller's hipGraph (torch.cuda.graph) and the replayed graph produces the eager
    call's tokens."""
    spec = L.SPEC_UNISE
    _, lm = _model(spec, 35, gpu_device)
    mix = L.synth_feats(43, B, 12, spec.feats_dim).to(gpu_device)
    mel = torch.zeros(B, 10, 80)
    eager = lm.generate("se", None, None, mel, mix, global_length=6, do_sample=False)
    torch.cuda.synchronize()
    eager = (eager[0].clone(), eager[1].clone())
    knob("QA_LM_CHAINS", 1)  # the layout a capturing caller gets: one eager call sizes the workspace for it (no allocation under capture)
    one = lm.generate("se", None, None, mel, mix, global_length=6, do_sample=False)
    torch.cuda.synchronize()
    assert torch.equal(one[0], eager[0]) and torch.equal(one[1], eager[1])  # chains are a schedule, not a result
    knob("QA_LM_CHAINS", 0)
    side = torch.cuda.Stream(gpu_device)
    side.wait_stream(torch.cuda.current_stream(gpu_device))
    with torch.cuda.stream(side):  # one plain call on the capture stream first, as torch's graph recipe prescribes
        lm.generate("se", None, None, mel, mix, global_length=6, do_sample=False)
    torch.cuda.current_stream(gpu_device).wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = lm.generate("se", None, None, mel, mix, global_length=6, do_sample=False)
    out[0].zero_()
    out[1].zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[0], eager[0]) and torch.equal(out[1], eager[1])
    again = lm.generate("se", None, None, mel, mix, global_length=6, do_sample=False)  # and the handle is usable eagerly afterwards
    torch.cuda.synchronize()
    assert torch.equal(again[0], eager[0]) and torch.equal(again[1], eager[1])


def test_graph_replay_equals_eager_launches_and_is_deterministic(qa_lib, gpu_device, knob):
    """One captured step replayed per token (QA_LM_GRAPH, the default) against the same kernels launched eagerly: identical
    integers; repeated calls re-use the captured graphs and the workspace and stay identical; a second shape re-captures."""
    spec = L.SPEC_UNISE
    _, lm = _model(spec, 33, gpu_device)
    outs = []
    for graph in ("1", "0", "1"):
        knob("QA_LM_GRAPH", int(graph))
        outs.append(_gen(lm, spec, gpu_device, 5, 20, 24, 32, do_sample=False))
    for g, s in outs[1:]:
        assert torch.equal(g, outs[0][0]) and torch.equal(s, outs[0][1])
    knob("QA_LM_GRAPH", 1)
    a = _gen(lm, spec, gpu_device, 3, 11, 9, 4, do_sample=False)   # other shape: new workspace layout, new graphs
    knob("QA_LM_GRAPH", 0)
    b = _gen(lm, spec, gpu_device, 3, 11, 9, 4, do_sample=False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_batch_17_to_32_uses_two_row_tiles(qa_lib, gpu_device):
    """M = 17..32 sequences run the MT = 2 instances of the fused GEMVs: every sequence must produce what it produces alone."""
    spec = SMALL
    sd, lm = _model(spec, 21, gpu_device)
    B, Nm, S, G = 19, 6, 7, 3
    mix = L.synth_feats(5, B, Nm, spec.feats_dim)
    mel = torch.zeros(B, S, 80)
    g, s = lm.generate("se", None, None, mel, mix.to(gpu_device), global_length=G, do_sample=False)
    _audit(sd, spec, "se", None, mix, S, G, g.cpu(), s.cpu())
    for i in (0, 16, 18):
        g1, s1 = lm.generate("se", None, None, mel[i:i + 1], mix[i:i + 1].to(gpu_device), global_length=G, do_sample=False)
        assert torch.equal(g1[0], g[i]) and torch.equal(s1[0], s[i])


def test_narrow_tile_kernels_agree(qa_lib, gpu_device):
    """The 4x4x1-MFMA narrow-tile GEMV kernels (NT = 8 / 4) must produce the oracle's tokens: a spec whose every GEMV uses narrow
    tiles.  (The 16x16x4 form of those tiles, QA_LM_MFMA16, measured equal in r02 and is gone since r05.)"""
    spec = L.LMSpec(hidden=256, n_layers=3, n_heads=4, global_size=64, semantic_size=96, feats_dim=64, num_tasks=3)
    sd, lm = _model(spec, 23, gpu_device)
    B, Nm, S, G = 4, 7, 10, 6
    mix = L.synth_feats(6, B, Nm, spec.feats_dim)
    g, s = lm.generate("se", None, None, torch.zeros(B, S, 80), mix.to(gpu_device), global_length=G, do_sample=False)
    _audit(sd, spec, "se", None, mix, S, G, g.cpu(), s.cpu())


def _tv(p, q):
    return 0.5 * float((p - q).abs().sum())


@pytest.mark.parametrize("top_k,top_p,temperature", [(50, 0.95, 0.8), (0, 0.9, 1.0), (5, 1.0, 0.5), (0, 1.0, 1.0), (3, 0.5, 0.3)])
def test_sample_logits_distribution_matches_reference_filters(qa_lib, gpu_device, top_k, top_p, temperature):
    """Device sampler vs the EXACT categorical distribution of the reference's sample_logits (oracle, pinned to llm.py:253-288):
    the support must be identical (top-k with ties kept, nucleus keeps the crossing token) and the empirical distribution of
    8192 independent draws must be within sampling noise."""
    import unified_audio_amd as qa

    V, n = 700, 8192
    gen = torch.Generator().manual_seed(5)
    row = torch.randn(V, generator=gen) * 2.5
    row[10:13] = row.max() + 0.5  # a 3-way tie at the top
    want = L.sampling_distribution(row[None], temperature=temperature, top_k=top_k, top_p=top_p)[0]
    logits = row[None].repeat(n, 1).to(gpu_device)
    idx = qa.sample_logits(logits, temperature=temperature, top_k=top_k, top_p=top_p, do_sample=True, seed=1234)[:, 0].cpu()
    emp = torch.bincount(idx, minlength=V).float() / n
    support = want > 0
    assert not (emp[~support] > 0).any(), "a filtered-out token was sampled"
    k_eff = int(support.sum())
    # total variation of an n-sample empirical distribution over k cells concentrates around sqrt(k / (2 pi n))
    assert _tv(emp, want) < 3.0 * (k_eff / (2 * 3.14159 * n)) ** 0.5 + 0.01, (_tv(emp, want), k_eff)
    # same seed -> same draws; another seed -> other draws; greedy -> the first maximum
    again = qa.sample_logits(logits, temperature=temperature, top_k=top_k, top_p=top_p, do_sample=True, seed=1234)[:, 0].cpu()
    other = qa.sample_logits(logits, temperature=temperature, top_k=top_k, top_p=top_p, do_sample=True, seed=99)[:, 0].cpu()
    assert torch.equal(idx, again) and (k_eff == 1 or not torch.equal(idx, other))
    greedy = qa.sample_logits(logits[:4], temperature=temperature, top_k=top_k, top_p=top_p, do_sample=False)[:, 0].cpu()
    assert (greedy == 10).all()


def test_sample_logits_wide_slice_and_all_equal_row(qa_lib, gpu_device):
    import unified_audio_amd as qa

    V = 8192  # the semantic slice width (sorted in LDS as 8192 64-bit keys)
    gen = torch.Generator().manual_seed(6)
    rows = torch.randn(16, V, generator=gen) * 3
    rows[1] = 0.25  # everything tied: top-k keeps all (strict '<', llm.py:263), nucleus keeps ceil-ish(top_p * V)
    want = L.sampling_distribution(rows, temperature=0.8, top_k=50, top_p=0.95)
    idx = qa.sample_logits(rows.to(gpu_device), seed=7)[:, 0].cpu()
    ok = want[torch.arange(16), idx] > 0
    ok[1] = True  # which of the tied entries survive the nucleus cut is the sort's choice (torch.sort is not stable): count checked below
    assert ok.all()
    n_keep_ref = int((want[1] > 0).sum())
    draws = qa.sample_logits(rows[1:2].repeat(4096, 1).to(gpu_device), seed=8)[:, 0].cpu()
    assert abs(int(draws.max()) + 1 - n_keep_ref) <= max(2, n_keep_ref // 1000)  # lowest indices first among ties; fp32 cumsum edge
    assert draws.unique().numel() > 0.3 * n_keep_ref


def test_generate_do_sample_default_runs_and_is_seeded(qa_lib, gpu_device):
    """do_sample=True is the signature default (llm_sft.py:106): ranges, determinism under torch.manual_seed, variation
    across seeds, and equality with greedy when the filters leave one candidate (top_k = 1)."""
    sd, lm = _model(SMALL, 21, gpu_device)
    B, Nm, S, G = 4, 9, 30, 8
    mix = L.synth_feats(1, B, Nm, SMALL.feats_dim).to(gpu_device)
    mel = torch.zeros(B, S, 80)
    torch.manual_seed(3)
    g1, s1 = lm.generate("se", None, None, mel, mix, global_length=G)
    torch.manual_seed(3)
    g2, s2 = lm.generate("se", None, None, mel, mix, global_length=G)
    g3, s3 = lm.generate("se", None, None, mel, mix, global_length=G)
    assert torch.equal(g1, g2) and torch.equal(s1, s2)
    assert not (torch.equal(g1, g3) and torch.equal(s1, s3))
    assert int(g1.min()) >= 0 and int(g1.max()) < SMALL.global_size and int(s1.min()) >= 0 and int(s1.max()) < SMALL.semantic_size
    gg, sg = lm.generate("se", None, None, mel, mix, global_length=G, do_sample=False)
    gk, sk = lm.generate("se", None, None, mel, mix, global_length=G, top_k=1)
    assert torch.equal(gg, gk) and torch.equal(sg, sk)


# ------------------------------------------------------------------------------------------- B > 32: concurrent chains

def test_batches_above_32_run_as_concurrent_chains(qa_lib, gpu_device, knob):
    """B > 32 (BASELINE configs[3] on ONE GPU is 64 segments): ceil(B / 32) chains on internal streams, each replaying its own
    captured step.  Every sequence must produce what it produces alone, the whole batch must pass the oracle audit, and forcing
    another chain count (QA_LM_CHAINS) must not change a token."""
    spec = SMALL
    sd, lm = _model(spec, 21, gpu_device)
    B, Nm, Ne, S, G = 40, 6, 5, 9, 3
    mix = L.synth_feats(5, B, Nm, spec.feats_dim)
    enr = L.synth_feats(6, B, Ne, spec.feats_dim)
    mel = torch.zeros(B, S, 80)
    g, s = lm.generate("tse", mel, enr.to(gpu_device), mel, mix.to(gpu_device), global_length=G, do_sample=False)
    torch.cuda.synchronize()
    assert g.shape == (B, G) and s.shape == (B, S)
    _audit(sd, spec, "tse", enr, mix, S, G, g.cpu(), s.cpu())
    for i in (0, 19, 20, 39):
        g1, s1 = lm.generate("tse", mel[i:i + 1], enr[i:i + 1].to(gpu_device), mel[i:i + 1], mix[i:i + 1].to(gpu_device), global_length=G,
                             do_sample=False)
        assert torch.equal(g1[0], g[i]) and torch.equal(s1[0], s[i]), i
    for chains in (1, 2, 4, 40):  # 1 (= the default since r05): ONE chain, two row groups per launch; 2: two chains of 20; 40: one sequence per chain is capped at 16 chains
        knob("QA_LM_CHAINS", chains)
        g2, s2 = lm.generate("tse", mel, enr.to(gpu_device), mel, mix.to(gpu_device), global_length=G, do_sample=False)
        assert torch.equal(g2, g) and torch.equal(s2, s), chains
    knob("QA_LM_CHAINS", 0)
    g3, s3 = lm.generate("tse", mel, enr.to(gpu_device), mel, mix.to(gpu_device), global_length=G, do_sample=False)  # cached graphs re-used
    assert torch.equal(g3, g) and torch.equal(s3, s)


@pytest.mark.parametrize("B", [64, 80, 33])
def test_chains_full_width_b64_matches_b16_chunks(qa_lib, gpu_device, B):
    """Full UniSE width.  64 segments = ONE chain whose launches carry two row groups of 32 (r05; two chains of 32 before); 80 = two
    chains of 40 (two row groups each, the second group 8 rows); 33 = one chain with a one-row second group: every 16-sequence chunk of
    the call equals the same chunk generated alone."""
    spec = L.SPEC_UNISE
    _, lm = _model(spec, 33, gpu_device)
    Nm, S = 40, 24
    mix = L.synth_feats(9, B, Nm).to(gpu_device)
    mel = torch.zeros(B, S, 80)
    g, s = lm.generate("se", None, None, mel, mix, do_sample=False)
    for b0 in (0, 16, 48) if B >= 64 else (0, 17):
        g1, s1 = lm.generate("se", None, None, mel[b0:b0 + 16], mix[b0:b0 + 16], do_sample=False)
        assert torch.equal(g1, g[b0:b0 + 16]) and torch.equal(s1, s[b0:b0 + 16]), b0
    if B > 64:  # the tail of the second chain's second row group
        g1, s1 = lm.generate("se", None, None, mel[B - 5:], mix[B - 5:], do_sample=False)
        assert torch.equal(g1, g[B - 5:]) and torch.equal(s1, s[B - 5:])


@pytest.mark.parametrize("spec_name", ["small", "unise"])
def test_row_group_boundaries_do_not_change_a_token(qa_lib, gpu_device, spec_name):
    """r05: which row groups a decode launch carries depends on the batch - 8-row groups for o_proj always and for MLP / qkv at 9 .. 16
    sequences, 16-row groups at 17 .. 32, 32-row groups above, the o_proj tile width 4 / 8 / 16, chains above 64 - but a row's arithmetic
    never knows its group: the first B sequences of one fixed batch must produce the same tokens for EVERY B across those boundaries."""
    spec = SMALL if spec_name == "small" else L.SPEC_UNISE
    _, lm = _model(spec, 29, gpu_device)
    Bmax, Nm, Ne, S, G = 66, 6, 5, (9 if spec_name == "small" else 6), 3
    mix = L.synth_feats(15, Bmax, Nm, spec.feats_dim).to(gpu_device)
    enr = L.synth_feats(16, Bmax, Ne, spec.feats_dim).to(gpu_device)
    mel = torch.zeros(Bmax, S, 80)
    g, s = lm.generate("tse", mel, enr, mel, mix, global_length=G, do_sample=False)
    sizes = (1, 7, 8, 9, 15, 16, 17, 31, 32, 33, 47, 48, 63, 64, 65) if spec_name == "small" else (8, 9, 16, 17, 32, 33, 64, 65)
    for B in sizes:
        gb, sb = lm.generate("tse", mel[:B], enr[:B], mel[:B], mix[:B], global_length=G, do_sample=False)
        assert torch.equal(gb, g[:B]) and torch.equal(sb, s[:B]), B


def test_sampled_chains_key_the_rng_by_the_global_sequence_index(qa_lib, gpu_device, knob):
    """The device sampler's Philox stream is keyed by (seed, sequence index IN THE CALL, step): splitting a call into chains must not
    change a draw."""
    sd, lm = _model(SMALL, 21, gpu_device)
    B, Nm, S, G = 36, 5, 10, 4
    mix = L.synth_feats(3, B, Nm, SMALL.feats_dim).to(gpu_device)
    mel = torch.zeros(B, S, 80)
    outs = []
    for chains in (0, 3, 6):
        knob("QA_LM_CHAINS", chains)
        torch.manual_seed(11)
        outs.append(lm.generate("se", None, None, mel, mix, global_length=G))
    for g, s in outs[1:]:
        assert torch.equal(g, outs[0][0]) and torch.equal(s, outs[0][1])
    assert outs[0][1][0].tolist() != outs[0][1][33].tolist() or outs[0][0][0].tolist() != outs[0][0][33].tolist()


def test_config4_on_one_gpu_64_tse_segments_match_the_reference_golden(qa_lib, gpu_device):
    """BASELINE configs[3] placed on ONE GPU: 64 TSE segments (prompt 503, KV 786) = one chain, two row groups of 32 per launch (r05; two concurrent chains of 32 before).  The batch is the
    reference-golden case `lm_config4_tse_b8` eight times over, so every block of 8 sequences must reproduce the token stream the
    reference's own LLM_SFT.generate produced (smallest top-2 gap of the golden 3.2e-4: no near-tie, the streams must be identical)."""
    import os

    import numpy as np

    from oracle import gen_golden_lm as GG

    name = "lm_config4_tse_b8"
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    spec, sd, task, mix, enr, S, G = GG.case_tensors(name)
    _, lm = _model(spec, GG.CASES[name][1], gpu_device)
    rep = 8
    mix64, enr64 = mix.repeat(rep, 1, 1).to(gpu_device), enr.repeat(rep, 1, 1).to(gpu_device)
    mel = torch.zeros(mix64.shape[0], S, 80)
    gids, sids = lm.generate(task, mel, enr64, mel, mix64, global_length=G, do_sample=False)
    want_g, want_s = torch.from_numpy(g["global_ids"].astype(np.int64)), torch.from_numpy(g["semantic_ids"].astype(np.int64))
    for r in range(rep):
        assert torch.equal(gids[8 * r:8 * r + 8].cpu(), want_g) and torch.equal(sids[8 * r:8 * r + 8].cpu(), want_s), r
