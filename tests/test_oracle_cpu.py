"""Pins the CPU oracle: against the reference's own modules (where /root/reference exists), against the committed golden
vectors (everywhere), and the C / torch RVQ restatements against each other."""
import dataclasses
import glob
import os

import numpy as np
import pytest
import torch

from oracle import hcodec_ref as R
from oracle import ref_shim, rvq_c, synth

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hcodec10_*.npz")))


def test_golden_fixtures_present():
    assert len(GOLDEN) >= 3


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_reference_golden(path):
    """tests/golden/*.npz were produced by the reference's vq.Codec (oracle/gen_golden.py); the restatement must
    reproduce the integer codes exactly and the waveform to fp32 round-off."""
    g = np.load(path)
    seed = int(g["seed"])
    sd = synth.hcodec10_state_dict(seed, head_logmag_bias=float(g["head_bias"]))
    if "stress" in g.files and int(g["stress"]):  # hcodec10_b1_stress: saturated LSTM gates, gamma ~ 1, half of the ISTFT bins in the clip
        sd = synth.stress_state_dict(sd)
    wav = R.pad_wav(synth.synth_wav(seed + 1, int(g["batch"]), int(g["samples"])))
    feat = synth.synth_feat(seed + 2, int(g["batch"]), wav.shape[-1] // 320)
    taps = {}
    spec = dataclasses.replace(R.SPEC_10, causal=bool(int(g["causal"])) if "causal" in g.files else False)
    with torch.no_grad():
        ac, sc = R.encode(sd, wav.unsqueeze(1), feat, spec, taps=taps)
        rec = R.decode(sd, ac, sc, spec)
    assert np.array_equal(ac.numpy(), g["acoustic_codes"].astype(np.int64))
    assert np.array_equal(sc.numpy(), g["semantic_codes"].astype(np.int64))
    np.testing.assert_allclose(taps["enc.emb"][:, ::37, ::3].numpy(), g["emb_sample"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(taps["enc.sem"][:, ::37, ::3].numpy(), g["sem_sample"], rtol=0, atol=2e-5)
    err = float(np.sqrt(np.mean((rec.numpy() - g["wav_rec"]) ** 2)) / np.sqrt(np.mean(g["wav_rec"] ** 2)))
    assert err < 1e-5, err
    assert rec.shape[-1] == wav.shape[-1]  # length rule len_out = ceil(len_in / hop) * hop


@pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference is only mounted in the build container")
def test_causal_restatement_matches_reference_modules():
    """spec.causal against the reference's own encoder / decoder classes constructed with causal=True (the flag every block
    takes); a causal graph must also differ from the non-causal one and must not look ahead."""
    sd = synth.hcodec10_state_dict(4322)
    model = ref_shim.load_state(ref_shim.make_causal_10(ref_shim.load_reference_codec("1.0")), sd)
    spec = dataclasses.replace(R.SPEC_10, causal=True)
    wav = R.pad_wav(synth.synth_wav(15, 2, 640 * 6 + 200))
    feat = synth.synth_feat(16, 2, wav.shape[-1] // 320)
    with torch.no_grad():
        emb_r = model.encoder(wav.unsqueeze(1))
        emb = R.seanet_encoder(sd, wav.unsqueeze(1), spec)
        assert float((emb - emb_r).abs().max()) < 2e-5 * float(emb_r.abs().max())
        ac_r, sc_r = model.encode(wav.unsqueeze(1), feat)
        ac, sc = R.encode(sd, wav.unsqueeze(1), feat, spec)
        assert torch.equal(ac, ac_r) and torch.equal(sc, sc_r)
        w_r, w = model.decode(ac_r, sc_r), R.decode(sd, ac_r, sc_r, spec)
        assert float((w - w_r).abs().max()) < 1e-5 * float(w_r.abs().max())
        assert not torch.equal(ac, R.encode(sd, wav.unsqueeze(1), feat)[0])
        # no look-ahead in the SEANet encoder + causal transformer: changing the last second leaves earlier frames unchanged
        wav2 = wav.clone()
        wav2[:, -1600:] = 0.3 * torch.randn(2, 1600, generator=torch.Generator().manual_seed(3))
        emb2 = R.seanet_encoder(sd, wav2.unsqueeze(1), spec)
        keep = (wav.shape[-1] - 1600) // 640
        # (the reflect padding of the LAST window is the only right-side dependence and lies inside the changed region)
        assert torch.equal(emb[..., :keep - 1], emb2[..., :keep - 1])
        assert not torch.equal(emb[..., keep + 1:], emb2[..., keep + 1:])


@pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference is only mounted in the build container")
def test_restatement_matches_reference_modules():
    sd = synth.hcodec10_state_dict(4321)
    model = ref_shim.load_state(ref_shim.load_reference_codec("1.0"), sd)
    ref_sd = model.state_dict()
    for k, v in ref_sd.items():  # same key names and shapes as the reference's own state_dict
        if not k.startswith("semantic_decoder."):
            assert k in sd and tuple(sd[k].shape) == tuple(v.shape), k
    assert not [k for k in sd if k not in ref_sd]
    wav = R.pad_wav(synth.synth_wav(5, 2, 640 * 7 + 11))
    feat = synth.synth_feat(6, 2, wav.shape[-1] // 320)
    with torch.no_grad():
        ac_r, sc_r = model.encode(wav.unsqueeze(1), feat)
        ac, sc = R.encode(sd, wav.unsqueeze(1), feat)
        assert torch.equal(ac, ac_r) and torch.equal(sc, sc_r)
        w_r, w = model.decode(ac_r, sc_r), R.decode(sd, ac_r, sc_r)
    assert float((w - w_r).abs().max()) < 1e-5 * float(w_r.abs().max())


def test_rvq_c_and_torch_restatements_agree():
    rng = np.random.default_rng(0)
    cb = np.stack([rng.standard_normal((128, 64)).astype(np.float32) * 0.5 ** q for q in range(3)])
    x = rng.standard_normal((500, 64)).astype(np.float32)
    idx_c = rvq_c.search_f32(x, cb)
    idx_t, quant_t = R.rvq_search(torch.from_numpy(x), torch.from_numpy(cb))
    assert (idx_c == idx_t.numpy()).all(axis=1).mean() > 0.995
    excess, best, gap = rvq_c.check_f64(x, cb, idx_c)
    assert excess.max() < 1e-4 and (idx_c[gap > 1e-4] == best[gap > 1e-4]).all()
    np.testing.assert_allclose(rvq_c.lookup_f32(idx_c, cb), R.rvq_lookup(torch.from_numpy(idx_c), torch.from_numpy(cb)).numpy(),
                               rtol=0, atol=1e-6)
    # core_vq.py:394-412 invariant: the quantised output is the sum of the looked-up codes
    np.testing.assert_allclose(quant_t.numpy(), R.rvq_lookup(idx_t, torch.from_numpy(cb)).numpy(), rtol=0, atol=1e-6)


def test_rvq_first_maximum_wins_on_exact_ties():
    cb = np.zeros((1, 8, 4), np.float32)
    cb[0, 3] = cb[0, 6] = [1, 0, 0, 0]
    x = np.array([[1, 0, 0, 0]], np.float32)
    assert rvq_c.search_f32(x, cb)[0, 0] == 3
    assert int(R.rvq_search(torch.from_numpy(x), torch.from_numpy(cb))[0][0, 0]) == 3


def test_pad_wav_rule():
    assert R.pad_wav(torch.zeros(1, 78480)).shape[-1] == 78720  # H15/sample.flac -> wav_rec.wav
    assert R.pad_wav(torch.zeros(1, 64000)).shape[-1] == 64000


# ------------------------------------------------------------------------------------------- H-Codec 1.5
GOLDEN_15 = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hcodec15_*.npz")))


def _spec15(threshold):
    import dataclasses

    return dataclasses.replace(R.SPEC_15, agg_layers=2, bt_layers=2, threshold=threshold)


@pytest.mark.parametrize("path", GOLDEN_15, ids=[os.path.basename(p)[:-4] for p in GOLDEN_15])
def test_oracle15_reproduces_reference_golden(path):
    from oracle import hcodec15_ref as R15

    g = np.load(path)
    seed, spec = int(g["seed"]), _spec15(float(g["threshold"]))
    flags = {k: (bool(g[k]) if k.endswith("causal") else int(g[k])) for k in ("agg_causal", "agg_context", "bt_causal", "bt_context") if k in g.files}
    spec = dataclasses.replace(spec, **flags)  # hcodec15_b2_causal_stacks: the reference built from the YAML with causal: true
    if "layers" in g.files:  # hcodec15_b2_full_depth: the published 32-layer stacks (654 M parameters), produced by the reference itself
        spec = dataclasses.replace(spec, agg_layers=int(g["layers"]), bt_layers=int(g["layers"]))
    sd = synth.hcodec10_state_dict(seed, spec)
    if "stress" in g.files and int(g["stress"]):
        sd = synth.stress_state_dict(sd)
    wav = R.pad_wav(synth.synth_wav(seed + 1, int(g["batch"]), int(g["samples"])))
    feat = synth.synth_feat(seed + 2, int(g["batch"]), wav.shape[-1] // 320, spec.sem_in)
    with torch.no_grad():
        codes = R15.encode(sd, wav.unsqueeze(1), feat, spec)
        rec = R15.decode(sd, codes["acoustic_codes"], codes["semantic_codes"], spec)
    assert np.array_equal(codes["acoustic_codes"].numpy(), g["acoustic_codes"].astype(np.int64))
    assert np.array_equal(codes["semantic_codes"].numpy(), g["semantic_codes"].astype(np.int64))
    err = float(np.sqrt(np.mean((rec.numpy() - g["wav_rec"]) ** 2)) / np.sqrt(np.mean(g["wav_rec"] ** 2)))
    assert err < 1e-5, err
    # code range of the length-injected format: < max_tokens * codebook_size
    assert int(codes["acoustic_codes"].max()) < spec.max_tokens_per_group * spec.codebook_size


def test_golden15_present():
    assert len(GOLDEN_15) >= 2


@pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference is only mounted in the build container")
def test_restatement15_matches_reference_modules():
    from oracle import hcodec15_ref as R15

    spec = _spec15(0.7)
    sd = synth.hcodec10_state_dict(777, spec)
    model = ref_shim.load_state(ref_shim.load_reference_codec("1.5", spec), sd)
    ref_sd = model.state_dict()
    assert not [k for k in ref_sd if k not in sd and not k.startswith("semantic_decoder.")]
    assert not [k for k in sd if k not in ref_sd]
    wav = R.pad_wav(synth.synth_wav(8, 2, 640 * 17 + 5))
    feat = synth.synth_feat(9, 2, wav.shape[-1] // 320, spec.sem_in)
    with torch.no_grad():
        ref = model.encode(wav.unsqueeze(1), feat)
        mine = R15.encode(sd, wav.unsqueeze(1), feat, spec)
        assert torch.equal(ref["acoustic_codes"], mine["acoustic_codes"]) and torch.equal(ref["semantic_codes"], mine["semantic_codes"])
        w_r = model.decode(**ref)
        w = R15.decode(sd, ref["acoustic_codes"], ref["semantic_codes"], spec)
    assert float((w - w_r).abs().max()) < 1e-5 * float(w_r.abs().max())


@pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference is only mounted in the build container")
def test_restatement15_causal_stacks_match_reference_modules():
    """The YAML's `causal: true` + `context_frames` / `context` for the aggregators and the bottleneck (config_adaptive_v3.yaml:84-105;
    shipped false): the reference builds causal, windowed mimi stacks from it and the restatement follows."""
    import dataclasses

    from oracle import hcodec15_ref as R15

    spec = dataclasses.replace(_spec15(0.7), agg_causal=True, agg_context=5, bt_causal=True, bt_context=7)
    sd = synth.hcodec10_state_dict(778, spec)
    model = ref_shim.load_state(ref_shim.load_reference_codec("1.5", spec), sd)
    wav = R.pad_wav(synth.synth_wav(18, 2, 640 * 15))
    feat = synth.synth_feat(19, 2, wav.shape[-1] // 320, spec.sem_in)
    with torch.no_grad():
        ref = model.encode(wav.unsqueeze(1), feat)
        mine = R15.encode(sd, wav.unsqueeze(1), feat, spec)
        assert torch.equal(ref["acoustic_codes"], mine["acoustic_codes"]) and torch.equal(ref["semantic_codes"], mine["semantic_codes"])
        w_r = model.decode(**ref)
        w = R15.decode(sd, ref["acoustic_codes"], ref["semantic_codes"], spec)
        assert float((w - w_r).abs().max()) < 1e-5 * float(w_r.abs().max())
        w_plain = R15.decode(sd, ref["acoustic_codes"], ref["semantic_codes"], _spec15(0.7))
    assert float((w - w_plain).abs().max()) > 1e-3 * float(w_r.abs().max())  # the flags do change the graph


def test_alignment_scan_rules():
    """Grouping rules of modeling_flexicodec_new.py:862-895 on a hand-made similarity pattern."""
    from oracle import hcodec15_ref as R15

    t = 20
    h = torch.zeros(1, t, 4)
    h[0, :, 0] = 1.0  # all frames identical -> one similarity run, split every 8 frames by max_tokens
    seg, nseg, _ = R15.similarity_alignment(h, 0.6, 8)
    assert seg[0].tolist() == [0] * 8 + [1] * 8 + [2] * 4 and int(nseg) == 3
    h[0, 5:, 0], h[0, 5:, 1] = 0.0, 1.0  # orthogonal from frame 5 on -> boundary at 5
    seg, nseg, _ = R15.similarity_alignment(h, 0.6, 8)
    assert seg[0].tolist() == [0] * 5 + [1] * 8 + [2] * 7 and int(nseg) == 3


# ------------------------------------------------------------------------------------------- H-Codec 2.0
def _spec20_small():
    from oracle import hcodec20_ref as R20
    from oracle.gen_golden import SPEC20_SMALL

    return R20.HCodec20Spec(**SPEC20_SMALL)


@pytest.mark.parametrize("name", ["hcodec20_small_b2", "hcodec20_small_b2_causal", "hcodec20_b1_full", "hcodec20_small_b2_stress"])
def test_oracle20_reproduces_reference_golden(name):
    from oracle import hcodec20_ref as R20

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    seed, spec = int(g["seed"]), _spec20_small()
    if "full" in g.files and int(g["full"]):  # hcodec20_b1_full: the shipped large_12.5hz_config.yaml shapes (1.17 G parameters)
        def _available_bytes():  # /proc/meminfo rather than psutil: a host without psutil must skip, not fail
            try:
                for line in open("/proc/meminfo"):
                    if line.startswith("MemAvailable:"):
                        return int(line.split()[1]) * 1024
            except OSError:
                pass
            return os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_AVPHYS_PAGES")

        if _available_bytes() < 20 * 2 ** 30:
            pytest.skip("the 1.17 G-parameter oracle needs ~12 GB of host memory")
        spec = R20.HCodec20Spec()
    if "causal" in g.files and int(g["causal"]):
        spec = dataclasses.replace(spec, causal=True)
    sd = synth.hcodec20_state_dict(seed, spec)
    if "stress" in g.files and int(g["stress"]):
        sd = synth.stress_state_dict(sd)
    wav = R.pad_wav(synth.synth_wav_fullband(seed + 1, int(g["batch"]), int(g["samples"])), spec.frame_hop)
    feat = synth.synth_feat(seed + 2, int(g["batch"]), wav.shape[-1] // spec.hop, spec.sem_in)
    with torch.no_grad():
        ac, sc = R20.encode(sd, wav, feat, spec)
        rec = R20.decode(sd, ac, sc, spec)
    assert np.array_equal(ac.numpy(), g["acoustic_codes"].astype(np.int64)) and np.array_equal(sc.numpy(), g["semantic_codes"].astype(np.int64))
    assert float(np.sqrt(np.mean((rec.numpy() - g["wav_rec"]) ** 2)) / np.sqrt(np.mean(g["wav_rec"] ** 2))) < 1e-5
    assert rec.shape[-1] == wav.shape[-1] and wav.shape[-1] % 3840 == 0  # H20/test.wav -> wav_rec.wav length rule


@pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference is only mounted in the build container")
def test_restatement20_matches_reference_modules():
    from oracle import hcodec20_ref as R20

    spec = _spec20_small()
    sd = synth.hcodec20_state_dict(31, spec)
    model = ref_shim.load_state(ref_shim.load_reference_codec("2.0", spec), sd)
    ref_sd = model.state_dict()
    assert not [k for k in ref_sd if k not in sd and not k.startswith("semantic_decoder.")] and not [k for k in sd if k not in ref_sd]
    wav = synth.synth_wav_fullband(32, 2, 3840 * 4)
    feat = synth.synth_feat(33, 2, wav.shape[-1] // spec.hop, spec.sem_in)
    with torch.no_grad():
        ac_r, sc_r = model.encode(wav, feat)
        ac, sc = R20.encode(sd, wav, feat, spec)
        assert torch.equal(ac, ac_r) and torch.equal(sc, sc_r)
        assert float((R20.decode(sd, ac_r, sc_r, spec) - model.decode(ac_r, sc_r)).abs().max()) < 1e-5


@pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference is only mounted in the build container")
def test_restatement20_causal_matches_reference_modules():
    """`causal: true` in encoder_config / decoder_config of the H-Codec 2.0 YAML (codec_encoder.py:23, codec_decoder.py:25; shipped
    false): causal Conv1d paddings of the embed / out / decoder-embed convolutions, the ConvNeXt depthwise and ResnetBlock k3, tril masks."""
    import dataclasses

    from oracle import hcodec20_ref as R20

    spec = dataclasses.replace(_spec20_small(), causal=True)
    sd = synth.hcodec20_state_dict(41, spec)
    model = ref_shim.load_state(ref_shim.load_reference_codec("2.0", spec), sd)
    wav = synth.synth_wav_fullband(42, 2, 3840 * 5)
    feat = synth.synth_feat(43, 2, wav.shape[-1] // spec.hop, spec.sem_in)
    with torch.no_grad():
        emb_r = model.encoder(wav)
        emb = R20.codec_encoder(sd, wav, spec)
        assert float((emb - emb_r).abs().max()) < 2e-5 * float(emb_r.abs().max())
        ac_r, sc_r = model.encode(wav, feat)
        ac, sc = R20.encode(sd, wav, feat, spec)
        assert torch.equal(ac, ac_r) and torch.equal(sc, sc_r)
        w_r, w = model.decode(ac_r, sc_r), R20.decode(sd, ac_r, sc_r, spec)
        assert float((w - w_r).abs().max()) < 1e-5 * max(1.0, float(w_r.abs().max()))
        w_plain = R20.decode(sd, ac_r, sc_r, _spec20_small())
    assert float((w - w_plain).abs().max()) > 1e-3 * float(w_r.abs().max())
