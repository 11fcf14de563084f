"""Edge cases and size-independent properties of the HIP H-Codec path: minimum-length clips, ragged
lengths through pad_wav, determinism, batch invariance at the BASELINE size (32 clips x 10 s), length rule."""
import dataclasses

import pytest
import torch

from oracle import hcodec15_ref as R15
from oracle import hcodec_ref as R
from oracle import synth
from tests.util import MINI, audit_codes_bnq, rel_err

pytestmark = pytest.mark.gpu


def _codec(ospec, seed, device):
    import unified_audio_amd as qa

    sd = synth.hcodec10_state_dict(seed, ospec)
    kw = {f: getattr(ospec, f) for f in ospec.__dataclass_fields__}
    return sd, qa.Codec(None, None, None, spec=qa.HCodecSpec(**kw), device=device).load_state_dict(sd)


@pytest.mark.parametrize("frames", [1, 2, 3])
def test_minimum_length_clips_match_oracle(qa_lib, gpu_device, frames):
    """One to three code frames (640 samples each): LSTM / attention over 2 positions, reflect padding of 2-frame inputs."""
    ospec = R.SPEC_10
    sd, codec = _codec(ospec, 1234, gpu_device)
    T = 640 * frames
    wav, feat = synth.synth_wav(3, 1, T), synth.synth_feat(4, 1, T // 320)
    taps = {}
    ac_o, sc_o = R.encode(sd, wav.unsqueeze(1), feat, ospec, taps)
    ac, sc = codec.encode(wav.to(gpu_device).unsqueeze(1), feat.to(gpu_device))
    assert ac.shape == (1, 4, frames)
    # 1-3 vectors: any flip must be an audited near-tie (max_flip_frac is a population bound, not applicable here)
    audit_codes_bnq(taps["enc.emb"], R.rvq_codebooks(sd, "quantizer", 4), ac, ac_o, max_flip_frac=1.0)
    audit_codes_bnq(taps["enc.sem"], R.rvq_codebooks(sd, "semantic_quantizer", 4), sc, sc_o, max_flip_frac=1.0)
    w_o = R.decode(sd, ac_o, sc_o, ospec)
    w = codec.decode(ac_o.to(gpu_device), sc_o.to(gpu_device))
    assert w.shape == (1, T) and rel_err(w, w_o) < 1e-4


def test_minimum_length_hcodec15(qa_lib, gpu_device):
    ospec = dataclasses.replace(R.SPEC_15, agg_layers=1, bt_layers=1)
    sd, codec = _codec(ospec, 5, gpu_device)
    wav, feat = synth.synth_wav(3, 2, 640), synth.synth_feat(4, 2, 2, 1024)
    ref = R15.encode(sd, wav.unsqueeze(1), feat, ospec)
    got = codec.encode(wav.to(gpu_device).unsqueeze(1), feat.to(gpu_device))
    assert got["acoustic_codes"].shape == ref["acoustic_codes"].shape == (2, 4, 1)  # T <= 1 frame: a single group
    w = codec.decode(ref["acoustic_codes"].to(gpu_device), ref["semantic_codes"].to(gpu_device))
    assert rel_err(w, R15.decode(sd, ref["acoustic_codes"], ref["semantic_codes"], ospec)) < 1e-4


def test_tokenizer_pads_ragged_length_like_reference(qa_lib, gpu_device):
    import unified_audio_amd as qa

    sd = synth.hcodec10_state_dict(9, R.HCodecSpec(**MINI))
    tok = qa.HCodecTokenizer(state_dict=sd, device=gpu_device, spec=qa.HCodecSpec(**MINI))
    wav = synth.synth_wav(1, 2, 16 * 10 + 7)  # MINI hop = 16 -> padded to 16 * 11
    feats = synth.synth_feat(2, 2, 22, 64).transpose(1, 2)
    ac, sc = tok.tokenize(wav, feats=feats)
    assert ac.shape == (2, 3, 11)
    assert tok.detokenize(ac, sc).shape == (2, 16 * 11)  # len_out = ceil(len_in / hop) * hop


def test_baseline_size_determinism_and_batch_invariance(qa_lib, gpu_device):
    """32 clips x 10 s @ 16 kHz (the BASELINE batch) through H-Codec 1.0: (a) two runs are bit-identical (no atomics, fixed
    reduction orders); (b) clips 0 and 31 processed alone give the same integer codes and the same waveform bits as inside
    the batch (every output element's summation order is independent of its tile position); (c) length rule."""
    ospec = R.SPEC_10
    sd, codec = _codec(ospec, 1234, gpu_device)
    B, T = 32, 160000
    wav = synth.synth_wav(7, B, T).to(gpu_device)
    feat = synth.synth_feat(9, B, T // 320).to(gpu_device)
    ac, sc = codec.encode(wav.unsqueeze(1), feat)
    rec = codec.decode(ac, sc)
    assert ac.shape == (B, 4, 250) and rec.shape == (B, T) and torch.isfinite(rec).all()
    assert int(ac.min()) >= 0 and int(ac.max()) < 1024 and int(sc.min()) >= 0 and int(sc.max()) < 1024
    ac2, sc2 = codec.encode(wav.unsqueeze(1), feat)
    assert torch.equal(ac, ac2) and torch.equal(sc, sc2) and torch.equal(rec, codec.decode(ac2, sc2))
    for i in (0, 31):
        a1, s1 = codec.encode(wav[i:i + 1].unsqueeze(1), feat[i:i + 1])
        assert torch.equal(a1[0], ac[i]) and torch.equal(s1[0], sc[i])
        assert torch.equal(codec.decode(a1, s1)[0], rec[i])
    # later stages see a spread of residuals: a collapsed search would show up as a single code
    assert ac[:, 3].unique().numel() > 4


def baseline_config_parity(device, B=32, seconds=10.0, seed=1234, verbose=print):
    """BASELINE configs[1] ITSELF - H-Codec 1.5 at full depth (32-layer aggregators + bottleneck, 654 M parameters on the
    path), B clips x 10 s @ 16 kHz - through the HIP path and through oracle/hcodec15_ref.py on the host cores:
      * grouping (token lengths) equal for every clip, except clips where the oracle's own cosine similarity sits within
        1e-4 of the threshold at a boundary (a near-tie of the `sim <= threshold` test; such clips are reported and skipped
        by the code audit, they may not exceed 2 of 32);
      * RVQ codes equal except audited near-ties (tests/util.audit_codes);
      * decode of the oracle's codes: waveform within 1e-3 RMS (north_star) and 1e-4 relative.
    Also called by __graft_entry__.smoke()."""
    import time

    import unified_audio_amd as qa

    ospec = R.SPEC_15
    K, nq = ospec.codebook_size, ospec.num_quantizers
    t0 = time.perf_counter()
    sd = synth.hcodec10_state_dict(seed, ospec)
    kw = {f: getattr(ospec, f) for f in ospec.__dataclass_fields__}
    codec = qa.Codec(None, None, None, spec=qa.HCodecSpec(**kw), device=device).load_state_dict(sd)
    T = int(round(seconds * 16000 / 640)) * 640
    wav = synth.synth_wav(seed + 1, B, T)
    feat = synth.synth_feat(seed + 2, B, T // 320, ospec.sem_in)
    t1 = time.perf_counter()
    got = codec.encode(wav.to(device).unsqueeze(1), feat.to(device))
    torch.cuda.synchronize()
    taps = {}
    with torch.no_grad():
        ref = R15.encode(sd, wav.unsqueeze(1), feat, ospec, taps)
    t2 = time.perf_counter()
    len_o = torch.div(ref["semantic_codes"][:, 0], K, rounding_mode="floor") + 1  # [B, G]; padded groups carry length 1... see below
    G_o, G_g = ref["acoustic_codes"].shape[-1], got["acoustic_codes"].shape[-1]
    nseg_o = taps["enc.seg"][:, -1] + 1
    margin = (taps["enc.sim"] - ospec.threshold).abs().min(dim=1).values  # closest call of each clip's threshold tests
    ok = torch.ones(B, dtype=torch.bool)
    if G_o == G_g:
        len_g = torch.div(got["semantic_codes"][:, 0].cpu(), K, rounding_mode="floor") + 1
        ok = (len_g == len_o).all(dim=1)
    else:
        ok[:] = False
    for b in torch.nonzero(~ok).flatten().tolist():
        assert float(margin[b]) < 1e-4, f"clip {b}: grouping differs from the oracle without a near-tie (margin {float(margin[b]):.2e})"
    assert int((~ok).sum()) <= max(2, B // 16), f"{int((~ok).sum())} clips with near-tie grouping flips"
    if G_o != G_g:  # (only possible when the clip that sets max-G flipped) nothing further is comparable position by position
        verbose(f"group count differs at a near-tie: oracle {G_o}, HIP {G_g}")
        flips = None
    else:
        keep = torch.nonzero(ok).flatten()
        sel = lambda t: t[keep]  # noqa: E731
        flips = max(
            audit_codes_bnq(sel(taps["enc.emb_agg"]), R.rvq_codebooks(sd, "quantizer", nq), sel(got["acoustic_codes"].cpu()) % K,
                            sel(ref["acoustic_codes"]) % K),
            audit_codes_bnq(sel(taps["enc.sem_agg"]), R.rvq_codebooks(sd, "semantic_quantizer", nq),
                            sel(got["semantic_codes"].cpu()) % K, sel(ref["semantic_codes"]) % K))
    with torch.no_grad():
        wav_o = R15.decode(sd, ref["acoustic_codes"], ref["semantic_codes"], ospec)
    wav_g = codec.decode(ref["acoustic_codes"].to(device), ref["semantic_codes"].to(device)).cpu()
    t3 = time.perf_counter()
    assert wav_g.shape == wav_o.shape == (B, T)
    rms = float((wav_g - wav_o).pow(2).mean().sqrt())
    rel = rel_err(wav_g, wav_o)
    assert rms < 1e-3 and rel < 1e-4, (rms, rel)
    verbose(f"BASELINE config parity: {B} x {T / 16000:.1f} s, groups {int(nseg_o.min())}..{int(nseg_o.max())}, clips with near-tie "
            f"grouping {int((~ok).sum())}, near-tie code flips {flips}, waveform RMS err {rms:.2e} (rel {rel:.2e}); "
            f"setup {t1 - t0:.0f} s, encode both {t2 - t1:.0f} s, decode both {t3 - t2:.0f} s")
    return dict(groups=(int(nseg_o.min()), int(nseg_o.max())), grouping_ties=int((~ok).sum()), code_flips=flips, rms=rms, rel=rel)


def test_baseline_config_at_size_matches_oracle(qa_lib, gpu_device):
    baseline_config_parity(gpu_device)
