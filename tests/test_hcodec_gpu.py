"""Whole-graph parity of the HIP H-Codec path (through Codec.encode / Codec.decode -> C-ABI) against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import hcodec_ref as R
from oracle import rvq_c, synth
from tests.util import MINI, audit_codes_bnq, mini_oracle_spec, rel_err

pytestmark = pytest.mark.gpu

STAGE_TOL = 5e-5  # relative RMS per stage, fp32 everywhere; only summation order / libm differ


def _cl(t):  # oracle [B,C,T] -> library layout [B,T,C], flattened
    return t.transpose(1, 2).contiguous().flatten()


def _make(spec_kwargs, seed, device):
    import unified_audio_amd as qa

    ospec = R.HCodecSpec(**spec_kwargs)
    sd = synth.hcodec10_state_dict(seed, ospec)
    codec = qa.Codec(None, None, None, spec=qa.HCodecSpec(**spec_kwargs), device=device).load_state_dict(sd)
    codec.enable_taps()
    return ospec, sd, codec


def _run_parity(spec_kwargs, B, T, device, seed=11):
    ospec, sd, codec = _make(spec_kwargs, seed, device)
    wav = synth.synth_wav(seed + 1, B, T)
    feat = synth.synth_feat(seed + 2, B, T // (ospec.enc_hop // 2), ospec.sem_in)
    taps = {}
    ac_o, sc_o = R.encode(sd, wav.unsqueeze(1), feat, ospec, taps)
    ac, sc = codec.encode(wav.to(device).unsqueeze(1), feat.to(device))
    torch.cuda.synchronize()
    report = {}
    for name in ["enc.conv0"] + [f"enc.stage{i}" for i in range(len(ospec.ratios))] + ["enc.transformer", "enc.emb", "enc.sem"]:
        report[name] = rel_err(codec.tap(name), _cl(taps[name]))
    tr = f"encoder.model.{3 * len(ospec.ratios) + 2}.layers.0.self_attn.rnn"
    report[tr] = rel_err(codec.tap(tr), taps[tr].flatten())
    # RVQ at the kernel boundary: feed the LIBRARY's own embeddings to the double-precision checker
    emb = codec.tap("enc.emb").view(-1, ospec.code_dim).cpu().numpy()
    sem = codec.tap("enc.sem").view(-1, ospec.code_dim).cpu().numpy()
    cb_a = R.rvq_codebooks(sd, "quantizer", ospec.num_quantizers).numpy()
    cb_s = R.rvq_codebooks(sd, "semantic_quantizer", ospec.num_quantizers).numpy()
    for e, cb, codes in ((emb, cb_a, ac), (sem, cb_s, sc)):
        got = codes.transpose(1, 2).reshape(-1, ospec.num_quantizers).cpu().numpy()
        excess, best, gap = rvq_c.check_f64(e, cb, got)
        tol = 2e-5 * float((e.astype(np.float64) ** 2).sum(1).mean())
        assert excess.max() <= tol
        assert (got[gap > tol] == best[gap > tol]).all()
    # whole-graph integer output: equal to the oracle's codes except at audited near-ties (tests/util.audit_codes)
    agree = (1.0 - audit_codes_bnq(taps["enc.emb"], cb_a, ac, ac_o), 1.0 - audit_codes_bnq(taps["enc.sem"], cb_s, sc, sc_o))
    # decode from the ORACLE's codes so that both sides start from identical integers
    dtaps = {}
    wav_o = R.decode(sd, ac_o, sc_o, ospec, dtaps)
    wav_g = codec.decode(ac_o.to(device), sc_o.to(device))
    torch.cuda.synchronize()
    for name in ["dec.embed", "dec.prior_res1", "dec.transformer", "dec.prior"]:
        report[name] = rel_err(codec.tap(name), _cl(dtaps[name]))
    report["dec.backbone"] = rel_err(codec.tap("dec.backbone"), dtaps["dec.backbone"].flatten())
    report["wav"] = rel_err(wav_g, wav_o)
    return report, agree, (wav_g.cpu(), wav_o)


def test_mini_codec_parity(qa_lib, gpu_device):
    report, agree, (wav_g, wav_o) = _run_parity(MINI, B=3, T=16 * 40, device=gpu_device)
    print(report, agree)
    bad = {k: v for k, v in report.items() if not v < STAGE_TOL}
    assert not bad, bad
    assert min(agree) > 0.998
    assert wav_g.shape == wav_o.shape


def test_causal_variant_parity_and_no_lookahead(qa_lib, gpu_device):
    """spec.causal (the causal= flags of conv.py:39-47 / transformer.py:432-475): stage-by-stage against the oracle,
    whose causal graph is pinned to the reference's own blocks built with causal=True (tests/test_oracle_cpu.py), and the property
    that makes it streamable: frames before a changed tail do not move."""
    kw = dict(MINI, causal=True)
    report, agree, (wav_g, wav_o) = _run_parity(kw, B=3, T=16 * 40, device=gpu_device)
    print(report, agree)
    bad = {k: v for k, v in report.items() if not v < STAGE_TOL}
    assert not bad, bad
    assert min(agree) > 0.998
    ospec, sd, codec = _make(kw, 11, gpu_device)
    hop = ospec.enc_hop
    wav = synth.synth_wav(21, 2, hop * 24)
    wav2 = wav.clone()
    wav2[:, -hop * 4:] = synth.synth_wav(22, 2, hop * 4)
    feat = synth.synth_feat(23, 2, wav.shape[-1] // (hop // 2), ospec.sem_in)
    codec.encode(wav.to(gpu_device).unsqueeze(1), feat.to(gpu_device))
    e1 = codec.tap("enc.emb").view(2, -1, ospec.code_dim).clone()
    codec.encode(wav2.to(gpu_device).unsqueeze(1), feat.to(gpu_device))
    e2 = codec.tap("enc.emb").view(2, -1, ospec.code_dim)
    keep = 24 - 4 - 1  # the last window before the change still reflects into it (extra right padding only at the end)
    assert torch.equal(e1[:, :keep], e2[:, :keep])
    assert not torch.equal(e1[:, keep + 2:], e2[:, keep + 2:])
    # and the non-causal graph does look ahead
    _, _, nc = _make(dict(MINI), 11, gpu_device)
    nc.encode(wav.to(gpu_device).unsqueeze(1), feat.to(gpu_device))
    n1 = nc.tap("enc.emb").view(2, -1, ospec.code_dim).clone()
    nc.encode(wav2.to(gpu_device).unsqueeze(1), feat.to(gpu_device))
    assert not torch.equal(n1[:, :keep], nc.tap("enc.emb").view(2, -1, ospec.code_dim)[:, :keep])


def test_hcodec10_full_size_parity(qa_lib, gpu_device):
    """The real H-Codec 1.0 architecture (141 M parameters) on 2 clips x 2 s (+ragged pad), seeded weights."""
    spec_kwargs = {f: getattr(R.SPEC_10, f) for f in R.SPEC_10.__dataclass_fields__}
    report, agree, (wav_g, wav_o) = _run_parity(spec_kwargs, B=2, T=640 * 51, device=gpu_device, seed=1234)
    print(report, agree)
    bad = {k: v for k, v in report.items() if not v < STAGE_TOL}
    assert not bad, bad
    assert min(agree) > 0.998
    # north_star tolerance: <= 1e-3 RMS on the reconstructed waveform (we hold it relative to the signal RMS too)
    assert float((wav_g - wav_o).pow(2).mean().sqrt()) < 1e-3
    assert report["wav"] < 1e-3


def test_baseline_config0_one_4s_clip(qa_lib, gpu_device):
    """BASELINE.json configs[0] as a parity case: H-Codec 1.0, ONE 4 s 16 kHz clip, T = 64 000,
    feat [1, 768, 200] -> codes [1, 4, 100] x 2 -> wav [1, 64 000]."""
    spec_kwargs = {f: getattr(R.SPEC_10, f) for f in R.SPEC_10.__dataclass_fields__}
    report, agree, (wav_g, wav_o) = _run_parity(spec_kwargs, B=1, T=64000, device=gpu_device, seed=4000)
    print(report, agree)
    bad = {k: v for k, v in report.items() if not v < STAGE_TOL}
    assert not bad, bad
    assert wav_g.shape == wav_o.shape == (1, 64000)
    assert float((wav_g - wav_o).pow(2).mean().sqrt()) < 1e-3 and report["wav"] < 1e-4


def test_encode_rejects_unpadded_wav(qa_lib, gpu_device):
    import unified_audio_amd as qa

    _, _, codec = _make(MINI, 3, gpu_device)
    with pytest.raises(qa.QuarkAudioError):
        codec.encode(torch.zeros(1, 1, 16 * 4 + 3, device=gpu_device), torch.zeros(1, 64, 8, device=gpu_device))


def test_decode_rejects_out_of_range_codes(qa_lib, gpu_device):
    _, _, codec = _make(MINI, 3, gpu_device)
    bad = torch.full((1, 3, 4), 64, dtype=torch.int64)
    with pytest.raises(IndexError):
        codec.decode(bad, bad)


# ------------------------------------------------------------------------------------------- H-Codec 1.5

def _spec15(**kw):
    import dataclasses

    return dataclasses.replace(R.SPEC_15, **kw)


def _run_parity_15(ospec, B, T, device, seed=31):
    import unified_audio_amd as qa
    from oracle import hcodec15_ref as R15

    sd = synth.hcodec10_state_dict(seed, ospec)
    kw = {f: getattr(ospec, f) for f in ospec.__dataclass_fields__}
    codec = qa.Codec(None, None, None, spec=qa.HCodecSpec(**kw), device=device).load_state_dict(sd)
    codec.enable_taps()
    wav = synth.synth_wav(seed + 1, B, T)
    feat = synth.synth_feat(seed + 2, B, T // 320, ospec.sem_in)
    taps = {}
    ref = R15.encode(sd, wav.unsqueeze(1), feat, ospec, taps)
    got = codec.encode(wav.to(device).unsqueeze(1), feat.to(device))
    torch.cuda.synchronize()
    assert set(got) == {"acoustic_codes", "semantic_codes"}
    assert got["acoustic_codes"].shape == ref["acoustic_codes"].shape, (got["acoustic_codes"].shape, ref["acoustic_codes"].shape)
    K = ospec.codebook_size
    # grouping (token lengths) is integer output of a threshold test on fp32 similarities: must match exactly here
    len_g = torch.div(got["semantic_codes"][:, 0].cpu(), K, rounding_mode="floor") + 1
    len_o = torch.div(ref["semantic_codes"][:, 0], K, rounding_mode="floor") + 1
    assert torch.equal(len_g, len_o)
    report = {
        "enc.emb_agg": rel_err(codec.tap("enc.emb_agg"), taps["enc.emb_agg"].transpose(1, 2).contiguous().flatten()),
        "enc.sem_agg": rel_err(codec.tap("enc.sem_agg"), taps["enc.sem_agg"].transpose(1, 2).contiguous().flatten()),
    }
    agree = 1.0 - max(
        audit_codes_bnq(taps["enc.emb_agg"], R.rvq_codebooks(sd, "quantizer", ospec.num_quantizers), got["acoustic_codes"] % K,
                        ref["acoustic_codes"] % K),
        audit_codes_bnq(taps["enc.sem_agg"], R.rvq_codebooks(sd, "semantic_quantizer", ospec.num_quantizers),
                        got["semantic_codes"] % K, ref["semantic_codes"] % K))
    wav_o = R15.decode(sd, ref["acoustic_codes"], ref["semantic_codes"], ospec)
    wav_g = codec.decode(ref["acoustic_codes"].to(device), ref["semantic_codes"].to(device))
    torch.cuda.synchronize()
    report["wav"] = rel_err(wav_g, wav_o)
    # decode(plain codes, token_lengths) is the same computation (codec_adaptive.py:184-186)
    plain_a, plain_s = ref["acoustic_codes"] % K, ref["semantic_codes"] % K
    wav_g2 = codec.decode(plain_a.to(device), plain_s.to(device), token_lengths=len_o.to(device))
    assert torch.equal(wav_g2, wav_g)
    return report, agree, int(ref["acoustic_codes"].shape[-1]), (wav_g.cpu(), wav_o)


@pytest.mark.parametrize("threshold", [0.6, 0.72])
def test_hcodec15_reduced_depth_parity(qa_lib, gpu_device, threshold):
    """H-Codec 1.5 at full width with 2-layer adaptive stacks (the layer count is a YAML knob of the reference)."""
    ospec = _spec15(agg_layers=2, bt_layers=2, threshold=threshold)
    report, agree, G, (wav_g, wav_o) = _run_parity_15(ospec, B=3, T=640 * 30, device=gpu_device)
    print(report, agree, "groups", G)
    assert all(v < STAGE_TOL for v in report.values()), report
    assert agree > 0.998
    assert wav_g.shape == wav_o.shape == (3, 640 * 30)


def test_hcodec15_full_depth_parity(qa_lib, gpu_device):
    """The shipped H-Codec 1.5 configuration: 32-layer aggregators and bottleneck (687 M parameters), 2 clips x 2.6 s."""
    report, agree, G, (wav_g, wav_o) = _run_parity_15(R.SPEC_15, B=2, T=640 * 65, device=gpu_device, seed=41)
    print(report, agree, "groups", G)
    assert all(v < 4 * STAGE_TOL for v in report.values()), report
    assert float((wav_g - wav_o).pow(2).mean().sqrt()) < 1e-3  # north_star waveform tolerance
    assert agree > 0.998


# ------------------------------------------------------------------------------------------- H-Codec 2.0

def _run_parity_20(ospec, B, T, device, seed=51, tol=STAGE_TOL):
    import unified_audio_amd as qa
    from oracle import hcodec20_ref as R20

    sd = synth.hcodec20_state_dict(seed, ospec)
    pspec = qa.HCodecSpec(version=20, enc_dim=ospec.enc_dim, enc_inter=ospec.enc_inter, enc_convnext_layers=ospec.enc_convnext_layers,
                          enc_layers=ospec.enc_transformer_layers, frame_stride=ospec.stride, tr_inter_cap=ospec.tr_inter_cap,
                          dimension=ospec.dimension, code_dim=ospec.dimension, sem_in=ospec.sem_in, sem_ch=ospec.sem_ch,
                          sem_strides=ospec.sem_strides, codebook_size=ospec.codebook_size, num_quantizers=ospec.num_quantizers,
                          dec_dim=ospec.dec_dim, dec_inter=ospec.dec_inter, dec_heads=ospec.dec_dim // 64,
                          dec_layers=ospec.dec_transformer_layers, convnext_layers=ospec.dec_convnext_layers, n_fft=ospec.n_fft,
                          hop=ospec.hop, gn_groups=ospec.gn_groups, causal=ospec.causal)
    codec = qa.Codec(None, None, None, spec=pspec, device=device).load_state_dict(sd)
    codec.enable_taps()
    wav = synth.synth_wav_fullband(seed + 1, B, T)
    feat = synth.synth_feat(seed + 2, B, T // ospec.hop, ospec.sem_in)
    taps = {}
    ac_o, sc_o = R20.encode(sd, wav, feat, ospec, taps)
    ac, sc = codec.encode(wav.to(device), feat.to(device))  # [B, T] without channel dim, like the reference
    torch.cuda.synchronize()
    nb2 = ospec.n_fft + 2
    stft = codec.tap("enc.stft").view(B, T // ospec.hop, -1)[..., :nb2].cpu()
    report = {"enc.stft": rel_err(stft, taps["enc.stft"].transpose(1, 2)),
              "enc.emb": rel_err(codec.tap("enc.emb"), _cl(taps["enc.emb"])), "enc.sem": rel_err(codec.tap("enc.sem"), _cl(taps["enc.sem"]))}
    agree = (1.0 - audit_codes_bnq(taps["enc.emb"], R.rvq_codebooks(sd, "quantizer", ospec.num_quantizers), ac, ac_o),
             1.0 - audit_codes_bnq(taps["enc.sem"], R.rvq_codebooks(sd, "semantic_quantizer", ospec.num_quantizers), sc, sc_o))
    wav_o = R20.decode(sd, ac_o, sc_o, ospec)
    wav_g = codec.decode(ac_o.to(device), sc_o.to(device))
    torch.cuda.synchronize()
    report["wav"] = rel_err(wav_g, wav_o)
    assert wav_g.shape == wav_o.shape == (B, T)
    return report, agree


def test_hcodec20_reduced_parity(qa_lib, gpu_device):
    from oracle import hcodec20_ref as R20

    ospec = R20.HCodec20Spec(enc_dim=256, enc_inter=512, enc_convnext_layers=2, enc_transformer_layers=1, dimension=128, sem_in=64,
                             sem_ch=128, codebook_size=64, num_quantizers=5, dec_dim=256, dec_inter=512, dec_convnext_layers=2,
                             dec_transformer_layers=1)
    report, agree = _run_parity_20(ospec, B=2, T=3840 * 6, device=gpu_device)
    print(report, agree)
    assert all(v < STAGE_TOL for v in report.values()), report
    assert min(agree) > 0.998


def test_hcodec20_full_width_parity(qa_lib, gpu_device):
    """Full H-Codec 2.0 widths (1536 / 4608, 24 heads, 16 codebooks, n_fft 1920) with 2 ConvNeXt blocks per stack."""
    from oracle import hcodec20_ref as R20
    import dataclasses

    ospec = dataclasses.replace(R20.SPEC_20, enc_convnext_layers=2, dec_convnext_layers=2)
    report, agree = _run_parity_20(ospec, B=2, T=3840 * 8, device=gpu_device, seed=61)
    print(report, agree)
    assert all(v < 2 * STAGE_TOL for v in report.values()), report
    assert min(agree) > 0.998


def test_hcodec20_causal_parity(qa_lib, gpu_device):
    """`causal: true` of the H-Codec 2.0 YAML (pinned to the reference built from it in tests/test_oracle_cpu.py)."""
    import dataclasses

    from oracle import hcodec20_ref as R20

    ospec = dataclasses.replace(R20.HCodec20Spec(enc_dim=256, enc_inter=512, enc_convnext_layers=2, enc_transformer_layers=1, dimension=128,
                                                 sem_in=64, sem_ch=128, codebook_size=64, num_quantizers=5, dec_dim=256, dec_inter=512,
                                                 dec_convnext_layers=2, dec_transformer_layers=1), causal=True)
    report, agree = _run_parity_20(ospec, B=2, T=3840 * 6, device=gpu_device, seed=53)
    print(report, agree)
    assert all(v < STAGE_TOL for v in report.values()), report


def test_hcodec20_full_depth_parity(qa_lib, gpu_device):
    """The published H-Codec 2.0 architecture at FULL depth (24 + 32 ConvNeXt blocks, 1.17 G parameters, large_12.5hz_config.yaml)
    on one 0.64 s clip: every stage and the waveform against the oracle, codes by the near-tie audit."""
    from oracle import hcodec20_ref as R20

    report, agree = _run_parity_20(R20.SPEC_20, B=1, T=3840 * 8, device=gpu_device, seed=71)
    print(report, agree)
    assert all(v < 4 * STAGE_TOL for v in report.values()), report  # 56 residual blocks deep: round-off accumulates
    assert report["wav"] < 1e-4


# ---- product mode overlaps the two H-Codec 1.5 aggregator stacks on internal streams; with taps enabled or qa_set_serial(1)
# everything stays on the caller's stream.  Same integers, same floats (and for 1.0: repeated calls re-use the arena).
def _streams_vs_serial(codec, wav, feat, adaptive):
    outs = []
    for taps in (True, False):
        codec.enable_taps(taps)
        enc = codec.encode(wav, feat)
        torch.cuda.synchronize()
        ac, sc = (enc["acoustic_codes"], enc["semantic_codes"]) if adaptive else enc
        w = codec.decode(ac, sc)
        torch.cuda.synchronize()
        outs.append((ac.clone(), sc.clone(), w.clone()))
    (ac0, sc0, w0), (ac1, sc1, w1) = outs
    assert torch.equal(ac0, ac1) and torch.equal(sc0, sc1) and torch.equal(w0, w1)


def test_internal_streams_match_serial_mini(qa_lib, gpu_device):
    ospec, sd, codec = _make(MINI, 71, gpu_device)
    B, T = 5, ospec.enc_hop * 40
    wav = synth.synth_wav(72, B, T).to(gpu_device)
    feat = synth.synth_feat(73, B, T // (ospec.enc_hop // 2), ospec.sem_in).to(gpu_device)
    for _ in range(3):  # repeated calls re-use the arena: a stale-buffer race would show up as a changed result
        _streams_vs_serial(codec, wav.unsqueeze(1), feat, False)


def test_internal_streams_match_serial_hcodec15(qa_lib, gpu_device):
    import unified_audio_amd as qa

    ospec = _spec15(agg_layers=2, bt_layers=2, threshold=0.7)
    sd = synth.hcodec10_state_dict(81, ospec)
    kw = {f: getattr(ospec, f) for f in ospec.__dataclass_fields__}
    codec = qa.Codec(None, None, None, spec=qa.HCodecSpec(**kw), device=gpu_device).load_state_dict(sd)
    B, T = 4, 640 * 25
    wav = synth.synth_wav(82, B, T).to(gpu_device)
    feat = synth.synth_feat(83, B, T // 320, ospec.sem_in).to(gpu_device)
    for _ in range(2):
        _streams_vs_serial(codec, wav.unsqueeze(1), feat, True)


