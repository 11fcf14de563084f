"""Host logic of the UniSE segmenting / batching driver against the reference's own lines (QuarkAudio-UniSE/model/model.py)."""
import math

import numpy as np
import pytest
import torch

from unified_audio_amd import unise as U


@pytest.mark.parametrize("T", [1, 7, 79999, 80000, 80001, 200000, 30001])
def test_wrap_pad_is_numpy_wrap(T):
    src = torch.arange(T, dtype=torch.float32)[None] * 0.5 - 3.0
    seg_len = 5 * 16000
    pad_len = math.ceil(src.size(-1) / seg_len) * seg_len - src.size(-1)          # model.py:176-177
    want = torch.from_numpy(np.pad(src.numpy(), [(0, 0), (0, pad_len)], "wrap"))   # model.py:178
    assert torch.equal(U.wrap_pad(src), want)


def test_segment_se_normalises_by_utterance_peak_and_tse_does_not():
    g = torch.Generator().manual_seed(3)
    src = torch.randn(1, 130001, generator=g) * 0.2
    seg_len = 80000
    pad_len = math.ceil(src.size(-1) / seg_len) * seg_len - src.size(-1)
    want = torch.from_numpy(np.pad(src.numpy(), [(0, 0), (0, pad_len)], "wrap")).reshape(-1, seg_len)
    assert torch.equal(U.segment(src, normalise=False), want)                      # model.py:199-203
    assert torch.equal(U.segment(src, normalise=True), want / src.abs().max(dim=-1, keepdim=True)[0])  # model.py:179-182
    with pytest.raises(ValueError):
        U.segment(torch.zeros(2, 100), True)


@pytest.mark.parametrize("T", [80000, 80001, 79999, 16000, 12345, 639, 641])
def test_mel_frames_matches_torch_stft(T):
    """stft_logmel of model.py:53-79 (hop 320, win = n_fft = 640, center=False) - frame count only."""
    x = torch.zeros(1, T)
    hop, win = 320, 640
    pad_length = math.ceil(x.size(-1) / hop) * hop - x.size(-1)
    xp = torch.nn.functional.pad(x, ((win - hop) // 2, pad_length + (win - hop) // 2))
    spec = torch.stft(xp, win, hop, win_length=win, window=torch.hann_window(win), onesided=True, center=False, return_complex=True)
    assert U.mel_frames(T) == spec.transpose(1, 2).shape[1]
    assert U.mel_frames(80000) == 250


def test_driver_batches_segments_of_several_utterances():
    """Composition logic with stand-in models: segments of all utterances go through the front-end and the LM as ONE batch,
    enrollments are tiled per utterance (model.py:207-210), tokens come back per utterance."""
    calls = {}

    class FakeSSL:
        def __call__(self, wavs):
            calls.setdefault("ssl", []).append(tuple(wavs.shape))
            return wavs[:, :6:2].unsqueeze(-1).repeat(1, 1, 4)  # [B, 3, 4], carries the first samples of each segment

    class FakeLM:
        def generate(self, task_name, enroll_mel, enroll_feats, mix_mel, mix_feats, do_sample):
            calls["lm"] = dict(task=task_name, B=mix_feats.shape[0], S=mix_mel.size(1),
                               enroll=None if enroll_feats is None else enroll_feats.clone(), Ne=None if enroll_mel is None else enroll_mel.size(1))
            B = mix_feats.shape[0]
            return torch.arange(B * 32).view(B, 32), torch.arange(B * mix_mel.size(1)).view(B, -1)

    drv = U.UniSE(FakeLM(), FakeSSL())
    a, b = torch.randn(1, 90000), torch.randn(1, 170000)
    out = drv.enhance_tokens("se", [a, b])
    assert calls["ssl"] == [(5, 80000)] and calls["lm"]["B"] == 5 and calls["lm"]["S"] == 250 and calls["lm"]["enroll"] is None
    assert [o[0].shape[0] for o in out] == [2, 3] and out[1][1][0, 0].item() == 2 * 250
    ea, eb = torch.full((1, 48000), 1.0), torch.full((1, 48000), 2.0)
    calls.clear()
    out = drv.enhance_tokens("tse", [a, b], [ea, eb])
    assert calls["ssl"] == [(2, 48000), (5, 80000)] and calls["lm"]["Ne"] == 150  # enrollment first, like model.py:204-212
    e = calls["lm"]["enroll"]
    assert e.shape[0] == 5 and (e[:2] == 1.0).all() and (e[2:] == 2.0).all()
    with pytest.raises(ValueError):
        drv.enhance_tokens("tse", [a, b], [ea])
    with pytest.raises(KeyError):
        drv.enhance_tokens("ss", [a])
    with pytest.raises(RuntimeError):
        drv.enhance("se", [a])


def test_micro_batches_of_max_segments_keep_results_and_enrollment_tiling():
    """UniSE(max_segments=m): the segments of a call go through the three stages m at a time (default 64); with batch-invariant stages the
    tokens and the waveforms do not depend on m, a micro-batch that starts in the middle of an utterance still gets THAT utterance's
    enrollment, and detokenize is chunked the same way."""
    log = []

    class FakeSSL:
        def __call__(self, wavs):
            return wavs[:, :6:2].unsqueeze(-1).repeat(1, 1, 4)

    class FakeLM:
        def generate(self, task_name, enroll_mel, enroll_feats, mix_mel, mix_feats, do_sample):
            B = mix_feats.shape[0]
            log.append(B)
            tag = mix_feats[:, 0, 0] * 1000 + (0 if enroll_feats is None else enroll_feats[:, 0, 0])  # depends on the segment AND its enrollment
            return tag.view(B, 1).repeat(1, 32).round().long(), tag.view(B, 1).repeat(1, mix_mel.size(1)).round().long()

    def detok(g, s):
        log.append(("detok", g.shape[0]))
        return (g[:, 0, :1].float() + s[:, :1].float()).view(-1, 1, 1).repeat(1, 1, 80000)

    utts = [torch.randn(1, n) for n in (90000, 250000, 80000, 170000)]  # 2 + 4 + 1 + 3 = 10 segments
    enr = [torch.full((1, 48000), float(i + 1)) for i in range(4)]
    ref_drv = U.UniSE(FakeLM(), FakeSSL(), detokenize=detok, max_segments=64)
    ref_tok = ref_drv.enhance_tokens("tse", utts, enr)
    ref_wav = ref_drv.enhance("tse", utts, enr)
    assert 10 in log
    for m in (1, 3, 4, 7):
        log.clear()
        drv = U.UniSE(FakeLM(), FakeSSL(), detokenize=detok, max_segments=m)
        tok = drv.enhance_tokens("tse", utts, enr)
        assert [b for b in log if isinstance(b, int)] == [min(m, 10 - a) for a in range(0, 10, m)]
        for (g0, s0), (g1, s1) in zip(ref_tok, tok):
            assert torch.equal(g0, g1) and torch.equal(s0, s1)
        wav = drv.enhance("tse", utts, enr)
        assert max(b[1] for b in log if isinstance(b, tuple)) <= m
        assert all(torch.equal(a, b) for a, b in zip(ref_wav, wav))


def test_ss_mode_runs_the_three_passes_of_the_reference():
    """model.py:223-290: SE on the first 5 s -> detokenize -> peak-normalised enrollment (x 0.99) -> TSE and rTSE over all segments
    with that enrollment tiled; two waveforms per mixture, each cut to the mixture's length."""
    log = []

    class FakeSSL:
        def __call__(self, wavs):
            log.append(("ssl", tuple(wavs.shape), float(wavs.abs().max())))
            return wavs[:, :4].unsqueeze(-1).repeat(1, 1, 2)

    class FakeLM:
        def generate(self, task_name, enroll_mel, enroll_feats, mix_mel, mix_feats, do_sample):
            B = mix_feats.shape[0]
            log.append(("lm", task_name, B, None if enroll_feats is None else tuple(enroll_feats.shape), None if enroll_mel is None else enroll_mel.size(1)))
            base = {"se": 0, "tse": 1000, "rtse": 2000}[task_name]
            return torch.full((B, 32), base), torch.full((B, 250), base) + torch.arange(B)[:, None]

    class FakeTok:
        def detokenize(self, global_tokens, semantic_tokens):
            assert global_tokens.shape[1:] == (1, 32) and semantic_tokens.shape[1] == 250
            B = semantic_tokens.shape[0]
            # waveform value encodes (task base + segment index); longer than a segment like a real codec's padding would be
            return (semantic_tokens[:, :1].float() * 0.001).view(B, 1, 1).expand(B, 1, 80000 + 320).contiguous()

    drv = U.UniSE(FakeLM(), FakeSSL(), tokenizer=FakeTok())
    a, b = torch.randn(1, 30000), torch.randn(1, 170000)   # shorter than one segment / three segments
    out = drv.enhance("ss", [a, b])
    lm = [x for x in log if x[0] == "lm"]
    assert [x[1:3] for x in lm] == [("se", 2), ("tse", 4), ("rtse", 4)]
    assert lm[1][3] == (4, 4, 2) and lm[1][4] == 250 and lm[2][3] == (4, 4, 2)   # enrollment features tiled 1 + 3, 5 s of mel frames
    ssl = [x for x in log if x[0] == "ssl"]
    assert ssl[0][1] == (2, 80000)                       # first 5 s of both mixtures (the short one wrap-padded), un-normalised
    assert ssl[1][1] == (2, 80000) and abs(ssl[1][2] - 0.99 * 0.001 / (0.001 + 1e-5)) < 1e-5   # enroll / (max|enroll| + 1e-5) * 0.99 (model.py:243)
    (s1a, s2a), (s1b, s2b) = out
    assert s1a.shape == s2a.shape == (30000,) and s1b.shape == s2b.shape == (170000,)
    assert abs(float(s1b[0]) - 1.001) < 1e-5 and abs(float(s2b[0]) - 2.001) < 1e-5 and abs(float(s1a[0]) - 1.000) < 1e-5
    assert abs(float(s1b[80320]) - 1.002) < 1e-5         # est.reshape(-1): segment 1 starts after segment 0's FULL decoded length (model.py:193)


def test_stft_logmel_matches_the_reference_formula():
    """model.py:53-79 with torchaudio.functional.melscale_fbanks(321, 0, 8000, 80, 16000) (HTK, norm=None) restated."""
    x = torch.randn(2, 16000 + 123, generator=torch.Generator().manual_seed(0))
    mel = U.stft_logmel(x)
    assert mel.shape == (2, U.mel_frames(x.shape[-1]), 80) and torch.isfinite(mel).all()
    # filter bank properties: triangles cover [0, 8000] Hz, peak 1 at their centres, every interior bin belongs to <= 2 filters
    import math
    tone = torch.sin(2 * math.pi * 1000.0 * torch.arange(16000) / 16000.0)[None]
    m = U.stft_logmel(tone)[0].mean(0)
    centre = 2595.0 * math.log10(1 + 1000.0 / 700.0) / (2595.0 * math.log10(1 + 8000.0 / 700.0)) * 81 - 1
    assert abs(int(m.argmax()) - centre) <= 1.5
