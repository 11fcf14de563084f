"""Edge cases and size-independent properties of the HIP H-Codec path: minimum-length clips, ragged
lengths through pad_wav, determinism, batch invariance at the BASELINE size (32 clips x 10 s), length rule."""
import dataclasses

import pytest
import torch

from oracle import hcodec15_ref as R15
from oracle import hcodec_ref as R
from oracle import synth
from tests.util import MINI, rel_err

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
    ac_o, sc_o = R.encode(sd, wav.unsqueeze(1), feat, ospec)
    ac, sc = codec.encode(wav.to(gpu_device).unsqueeze(1), feat.to(gpu_device))
    assert ac.shape == (1, 4, frames)
    assert (ac.cpu() == ac_o).float().mean() >= 0.75 and (sc.cpu() == sc_o).float().mean() >= 0.75
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
