"""SURVEY 8f-4 on the GPU: causal / context windows and the RingKVCache streaming step of the mimi StreamingTransformer through the
C-ABI (qa_mimi_*), against the oracle (pinned to the reference's own module in tests/test_mimi_stream_cpu.py) and the golden vectors
the reference produced."""
import os

import numpy as np
import pytest
import torch

from oracle import hcodec15_ref as R15
from oracle import synth
from tests.test_mimi_stream_cpu import CHUNKS, CTX, D, FF, GOLDEN, H, L, stream_chunks
from tests.util import audit_codes_bnq, rel_err

pytestmark = pytest.mark.gpu
TOL = 5e-5


def _model(sd, causal, context, device, **kw):
    import unified_audio_amd as qa

    m = qa.StreamingTransformer(kw.get("d", D), kw.get("h", H), kw.get("layers", L), kw.get("ff", FF), causal=causal, context=context,
                                device=device, prefix="transformer")
    return m.load_state_dict(sd)


@pytest.mark.parametrize("causal,context", [(False, CTX), (True, CTX), (True, 0), (True, 40)])
def test_offline_windows_match_oracle(qa_lib, gpu_device, causal, context):
    sd = synth.mimi_state_dict(31, D, L, FF)
    x = torch.randn(3, 70, D, generator=torch.Generator().manual_seed(5))  # 70 frames: several 32-key tiles, the window skips some
    with torch.no_grad():
        ref = R15.mimi_transformer(sd, "transformer", x, L, H, causal, context)
    y = _model(sd, causal, context or None, gpu_device)(x.to(gpu_device))
    assert rel_err(y, ref) < TOL


def test_streaming_matches_oracle_and_reference_golden(qa_lib, gpu_device):
    g = np.load(GOLDEN)
    seed = int(g["seed"])
    sd = synth.mimi_state_dict(seed, D, L, FF)
    x = torch.randn(2, sum(CHUNKS), D, generator=torch.Generator().manual_seed(seed + 1))
    m = _model(sd, True, CTX, gpu_device)
    xg = x.to(gpu_device)
    off = m(xg)
    assert rel_err(off, torch.from_numpy(g["offline"])) < TOL
    with m.streaming(2):
        y = stream_chunks(m, xg, CHUNKS)
        assert m.streaming_offset == sum(CHUNKS)
        m.reset_streaming()  # stale ring contents must be invisible after a reset
        y1 = stream_chunks(m, xg[:, :9], (1,) * 9)
    assert not m.is_streaming
    assert rel_err(y, torch.from_numpy(g["streamed"])) < TOL
    # frame-by-frame == the offline stack with a window of context - 1 (the reference's RingKVCache quirk, see the CPU test)
    with torch.no_grad():
        off_m1 = R15.mimi_transformer(sd, "transformer", x, L, H, True, CTX - 1)
    assert rel_err(y1, off_m1[:, :9]) < TOL
    assert rel_err(y, off_m1) > 1e-3


def test_streaming_long_run_wide_heads(qa_lib, gpu_device):
    """head_dim 64, ring of 16 (the YAML's context), 100 single-frame steps then mixed chunks: the ring wraps many times."""
    d, h, layers, ff, ctx = 256, 4, 2, 512, 16
    sd = synth.mimi_state_dict(77, d, layers, ff)
    x = torch.randn(2, 130, d, generator=torch.Generator().manual_seed(78))
    chunks = (1,) * 100 + (7, 16, 3, 4)
    st = R15.MimiStreamState(2, layers, h, d // h, ctx)
    with torch.no_grad():
        ref = stream_chunks(lambda c: R15.mimi_transformer(sd, "transformer", c, layers, h, True, ctx, st), x, chunks)
    m = _model(sd, True, ctx, gpu_device, d=d, h=h, layers=layers, ff=ff)
    with m.streaming(2):
        y = stream_chunks(m, x.to(gpu_device), chunks)
    assert rel_err(y, ref) < TOL


def test_streaming_forever_rolls_the_rope_window(qa_lib, gpu_device, knob):
    """StreamingTransformer.streaming_forever(): the reference computes RoPE from the running offset and never stops
    (mimi/module/rope.py:38-56).  Here positions beyond the static table come from a rolling window of it; QA_MIMI_ROPE_WINDOW
    shrinks the table to 32 positions so that a 130-frame stream rolls the window several times (chunks straddle its end)."""
    d, h, layers, ff, ctx = 256, 4, 2, 512, 16
    sd = synth.mimi_state_dict(77, d, layers, ff)
    x = torch.randn(2, 130, d, generator=torch.Generator().manual_seed(78))
    chunks = (1,) * 100 + (7, 16, 3, 4)
    st = R15.MimiStreamState(2, layers, h, d // h, ctx)
    with torch.no_grad():
        ref = stream_chunks(lambda c: R15.mimi_transformer(sd, "transformer", c, layers, h, True, ctx, st), x, chunks)
    knob("QA_MIMI_ROPE_WINDOW", 32)
    m = _model(sd, True, ctx, gpu_device, d=d, h=h, layers=layers, ff=ff)
    with m.streaming(2):
        y = stream_chunks(m, x.to(gpu_device), chunks)
        assert m.streaming_offset == 130
        m.reset_streaming()  # back inside the static table
        y1 = stream_chunks(m, x[:, :5].to(gpu_device), (5,))
    assert rel_err(y, ref) < TOL
    st2 = R15.MimiStreamState(2, layers, h, d // h, ctx)
    with torch.no_grad():
        ref1 = R15.mimi_transformer(sd, "transformer", x[:, :5], layers, h, True, ctx, st2)
    assert rel_err(y1, ref1) < TOL


def test_streaming_errors(qa_lib, gpu_device):
    import unified_audio_amd as qa

    sd = synth.mimi_state_dict(31, D, L, FF)
    with pytest.raises(qa.QuarkAudioError, match="causal"):
        with _model(sd, False, CTX, gpu_device).streaming(1):
            pass
    with pytest.raises(qa.QuarkAudioError, match="context"):
        with _model(sd, True, None, gpu_device).streaming(1):
            pass
    m = _model(sd, True, CTX, gpu_device)
    with pytest.raises(qa.QuarkAudioError, match="wasn't streaming"):
        m.reset_streaming()
    with m.streaming(2):
        with pytest.raises(qa.QuarkAudioError, match="does not fit the ring"):
            m(torch.zeros(2, CTX + 1, D, device=gpu_device))
        with pytest.raises(qa.QuarkAudioError, match="batch"):
            m(torch.zeros(3, 1, D, device=gpu_device))


def test_hcodec15_causal_stacks_match_oracle(qa_lib, gpu_device):
    """H-Codec 1.5 with the YAML's `causal: true` for the aggregators and the bottleneck (pinned to the reference's modules built
    from that YAML in tests/test_oracle_cpu.py): decode from the oracle's codes, waveform within the bar."""
    import dataclasses

    import unified_audio_amd as qa
    from oracle import hcodec_ref as R

    ospec = dataclasses.replace(R.SPEC_15, agg_layers=2, bt_layers=2, threshold=0.7, agg_causal=True, agg_context=5, bt_causal=True,
                                bt_context=7)
    sd = synth.hcodec10_state_dict(778, ospec)
    kw = {f: getattr(ospec, f) for f in ospec.__dataclass_fields__}
    tok = qa.HCodecTokenizer(state_dict=sd, device=gpu_device, spec=qa.HCodecSpec(**kw))
    wav = synth.synth_wav(18, 2, 640 * 15)
    feat = synth.synth_feat(19, 2, wav.shape[-1] // 320, ospec.sem_in)
    taps = {}
    with torch.no_grad():
        codes = R15.encode(sd, R.pad_wav(wav).unsqueeze(1), feat, ospec, taps)
        rec_o = R15.decode(sd, codes["acoustic_codes"], codes["semantic_codes"], ospec)
        rec_plain = R15.decode(sd, codes["acoustic_codes"], codes["semantic_codes"], dataclasses.replace(ospec, bt_causal=False))
    got = tok.tokenize(wav.to(gpu_device), feats=feat.transpose(1, 2).contiguous().to(gpu_device))
    assert torch.equal(got["acoustic_codes"].cpu() // 1024, codes["acoustic_codes"] // 1024)  # same grouping
    audit_codes_bnq(taps["enc.emb_agg"], R.rvq_codebooks(sd, "quantizer", 4), got["acoustic_codes"] % 1024, codes["acoustic_codes"] % 1024)
    audit_codes_bnq(taps["enc.sem_agg"], R.rvq_codebooks(sd, "semantic_quantizer", 4), got["semantic_codes"] % 1024,
                    codes["semantic_codes"] % 1024)
    rec = tok.detokenize(codes["acoustic_codes"].to(gpu_device), codes["semantic_codes"].to(gpu_device))
    assert rel_err(rec, rec_o) < 1e-4
    assert rel_err(rec, rec_plain) > 1e-3
