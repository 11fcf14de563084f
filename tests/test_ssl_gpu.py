"""SSL front-end (qa_ssl_* through SSLFeatureExtractor) against the CPU oracle on seeded HF-layout weights."""
import dataclasses

import pytest
import torch

from oracle import ssl_ref as S
from tests.util import rel_err

pytestmark = pytest.mark.gpu

TOL = 5e-5  # relative RMS on the (uncompressed) hidden-state average: fp32 everywhere, only summation order / libm differ


def _run(ospec, kind, B, T, device, seed=3):
    import unified_audio_amd as qa

    sd = S.synth_state_dict(seed, ospec, kind)
    kw = {f: getattr(ospec, f) for f in ospec.__dataclass_fields__}
    g = torch.Generator().manual_seed(seed + 10)
    wav = torch.randn(B, T, generator=g) * 0.2 + 0.05 * torch.sin(torch.arange(T) * 0.03)[None]
    out = {}
    for expo in (0.0, ospec.compress_exponent):
        spec = dataclasses.replace(ospec, compress_exponent=expo)
        with torch.no_grad():
            ref = S.extract_features(sd, wav, spec)
        fx = qa.SSLFeatureExtractor(qa.SSLSpec(**{**kw, "compress_exponent": expo}), device=device).load_state_dict(sd)
        got = fx(wav.to(device))
        torch.cuda.synchronize()
        assert got.shape == ref.shape == (B, fx.frames(T), ospec.hidden_size)
        assert torch.isfinite(got).all()
        out[expo] = (got.cpu(), ref)
    plain, comp = out[0.0], out[ospec.compress_exponent]
    err = rel_err(*plain)
    # |x|^0.3 has unbounded slope at 0: compare the compressed features in absolute terms, away from sign flips of tiny means
    far = plain[1].abs() > 1e-3
    cerr = float((comp[0] - comp[1])[far].abs().max())
    return err, cerr


def test_ssl_small_post_ln_group_norm(qa_lib, gpu_device):
    spec = S.SSLSpec(conv_dim=(64,) * 7, hidden_size=96, num_hidden_layers=3, num_attention_heads=3, intermediate_size=192,
                     num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=2)
    err, cerr = _run(spec, "hubert", B=3, T=4000, device=gpu_device)
    print(err, cerr)
    assert err < TOL and cerr < 1e-3


def test_ssl_small_stable_ln_layer_norm(qa_lib, gpu_device):
    spec = S.SSLSpec(conv_dim=(64,) * 7, hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=256,
                     num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=2, conv_bias=True, feat_extract_norm="layer",
                     do_stable_layer_norm=True, select=(1, 3))
    err, cerr = _run(spec, "wav2vec2", B=2, T=5000, device=gpu_device)
    print(err, cerr)
    assert err < TOL and cerr < 1e-3


def test_ssl_hubert_base_width(qa_lib, gpu_device):
    """HuBERT-base widths (512-channel extractor, d = 768, 12 heads, FFN 3072, 16-group k128 positional conv), 2 layers, 2 x 1 s."""
    spec = dataclasses.replace(S.SPEC_HUBERT_BASE, num_hidden_layers=2)
    err, cerr = _run(spec, "hubert", B=2, T=16000, device=gpu_device)
    print(err, cerr)
    assert err < TOL and cerr < 1e-3


def test_ssl_xlsr_width(qa_lib, gpu_device):
    """wav2vec2-large / XLSR widths (conv bias + per-layer LayerNorm, d = 1024, 16 heads, FFN 4096, stable LN), 2 layers."""
    spec = dataclasses.replace(S.SPEC_XLSR53, num_hidden_layers=2, select=(1, 2))
    err, cerr = _run(spec, "wav2vec2", B=1, T=16000, device=gpu_device)
    print(err, cerr)
    assert err < TOL and cerr < 1e-3


def test_ssl_rejects_short_input(qa_lib, gpu_device):
    import unified_audio_amd as qa

    spec = S.SSLSpec(conv_dim=(64,) * 7, hidden_size=96, num_hidden_layers=1, num_attention_heads=3, intermediate_size=192,
                     num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=2)
    sd = S.synth_state_dict(1, spec)
    kw = {f: getattr(spec, f) for f in spec.__dataclass_fields__}
    fx = qa.SSLFeatureExtractor(qa.SSLSpec(**kw), device=gpu_device).load_state_dict(sd)
    with pytest.raises(qa.QuarkAudioError):
        fx(torch.zeros(1, 50, device=gpu_device))


def test_tokenizer_with_hip_front_end_equals_precomputed_features(qa_lib, gpu_device):
    """HCodecTokenizer.tokenize(wav) with the SSLFeatureExtractor in front == tokenize(wav, feats=extractor(wav))
    (audio_tokenizer.py:56-62), and the frame count is the N50 the codec expects."""
    import unified_audio_amd as qa
    from oracle import hcodec_ref as R
    from oracle import synth
    from tests.util import MINI

    sspec = S.SSLSpec(conv_dim=(64,) * 7, hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128,
                      num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=1)
    ssd = S.synth_state_dict(9, sspec)
    fx = qa.SSLFeatureExtractor(qa.SSLSpec(**{f: getattr(sspec, f) for f in sspec.__dataclass_fields__}), device=gpu_device).load_state_dict(ssd)
    # a codec whose frame rates match the extractor: hop 640 = 2 * prod(ratios) with ratios (8, 5, 4, 2)
    kw = dict(MINI, ratios=(8, 5, 4, 2), dimension=512, code_dim=512, enc_heads=8, hop=320, n_fft=1280)
    ospec = R.HCodecSpec(**kw)
    sd = synth.hcodec10_state_dict(13, ospec)
    tok = qa.HCodecTokenizer(state_dict=sd, feature_extractor=fx, device=gpu_device, spec=qa.HCodecSpec(**kw))
    wav = synth.synth_wav(14, 2, 640 * 12 - 100).to(gpu_device)  # not a multiple of the hop: pad_wav applies
    ac, sc = tok.tokenize(wav)
    feats = fx(tok.pad_wav(wav))
    assert feats.shape[1] == 2 * ac.shape[-1]
    ac2, sc2 = tok.tokenize(wav, feats=feats)
    assert torch.equal(ac, ac2) and torch.equal(sc, sc2)
    assert tok.detokenize(ac, sc).shape == (2, 640 * 12)


def test_ssl_small_wavlm_gated_relative_bias(qa_lib, gpu_device):
    """WavLM: few buckets / short saturation distance, so 26 frames cover the exact, logarithmic and saturated buckets."""
    spec = S.SSLSpec(conv_dim=(64,) * 7, hidden_size=96, num_hidden_layers=3, num_attention_heads=3, intermediate_size=192,
                     num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=2, num_buckets=16, max_bucket_distance=10)
    err, cerr = _run(spec, "wavlm", B=2, T=8200, device=gpu_device)
    print(err, cerr)
    assert err < TOL and cerr < 1e-3


def test_ssl_wavlm_base_plus_width(qa_lib, gpu_device):
    """wavlm-base-plus widths and its 320 buckets / 800 max distance, 2 layers, 3 s (149 frames: exact + logarithmic buckets);
    UniSE's recipe: plain mean of the hidden states (QuarkAudio-UniSE/model/model.py:38-51)."""
    spec = dataclasses.replace(S.SPEC_WAVLM_BASE_PLUS, num_hidden_layers=2, compress_exponent=0.3)
    err, cerr = _run(spec, "wavlm", B=2, T=48000, device=gpu_device)
    print(err, cerr)
    assert err < TOL and cerr < 1e-3
