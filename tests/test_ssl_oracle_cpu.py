"""The SSL-front-end oracle (oracle/ssl_ref.py) pinned against the container's transformers HubertModel / Wav2Vec2Model."""
import dataclasses

import pytest
import torch

from oracle import ssl_ref as S

SMALL = dict(conv_dim=(64,) * 7, hidden_size=96, num_hidden_layers=3, num_attention_heads=3, intermediate_size=192,
             num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=2)


@pytest.mark.parametrize("kind,flavour", [("hubert", "base"), ("wav2vec2", "large"), ("hubert", "large"), ("wavlm", "base")])
def test_restatement_matches_transformers(kind, flavour):
    from transformers import HubertModel, Wav2Vec2Model, WavLMModel

    spec = S.SSLSpec(**SMALL) if flavour == "base" else S.SSLSpec(**SMALL, conv_bias=True, feat_extract_norm="layer",
                                                                  do_stable_layer_norm=True, select=(1, 3))
    if kind == "wavlm":  # few buckets and a short saturation distance so that 13 frames exercise exact, log and saturated buckets
        spec = dataclasses.replace(spec, num_buckets=16, max_bucket_distance=10)
    sd = S.synth_state_dict(5, spec, kind)
    model = {"hubert": HubertModel, "wav2vec2": Wav2Vec2Model, "wavlm": WavLMModel}[kind](S.hf_config(spec, kind)).eval()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all("masked_spec_embed" in m for m in missing), (missing, unexpected)
    torch.manual_seed(0)
    wav = torch.randn(2, 4000) * 0.3
    with torch.no_grad():
        ref = model(torch.nn.functional.pad(wav, (spec.pad, spec.pad)), output_hidden_states=True).hidden_states
        got = S.hidden_states(sd, torch.nn.functional.pad(wav, (spec.pad, spec.pad)), spec)
    assert len(ref) == len(got) == spec.num_hidden_layers + 1
    for a, b in zip(got, ref):
        assert a.shape == b.shape
        assert torch.allclose(a, b, rtol=1e-4, atol=2e-5), float((a - b).abs().max())


def test_extract_features_is_the_reference_recipe():
    """audio_tokenizer.py:35-48: mean over stacked hidden states, then sign * |x| ** 0.3 with sign(0) = -1."""
    spec = S.SSLSpec(**SMALL)
    sd = S.synth_state_dict(7, spec)
    wav = torch.randn(1, 3200) * 0.2
    with torch.no_grad():
        hs = S.hidden_states(sd, torch.nn.functional.pad(wav, (160, 160)), spec)
        feats_mix = torch.stack(hs, dim=1).mean(1)
        want = ((feats_mix > 0).float() * 2 - 1) * feats_mix.abs() ** 0.3
        got = S.extract_features(sd, wav, spec)
        plain = S.extract_features(sd, wav, dataclasses.replace(spec, compress_exponent=0.0))
    assert torch.equal(got, want) and torch.equal(plain, feats_mix)
    assert got.shape == (1, (3200 + 320 - 400) // 320 + 1, spec.hidden_size)


def test_frame_count_rule():
    """10 s @16 kHz padded by 160+160 gives 500 frames = the N50 H-Codec expects."""
    L = 160000 + 320
    for k, s in zip(S.SPEC_HUBERT_BASE.conv_kernel, S.SPEC_HUBERT_BASE.conv_stride):
        L = (L - k) // s + 1
    assert L == 500
