"""Pins the facade `unified_audio_amd.HCodecTokenizer` (pad rule, SSL feature recipe per version, call conventions of tokenize /
detokenize) to the reference's OWN three HCodecTokenizer classes (oracle/ref_tokenizer_shim.py), both wrapped around the SAME
reference `vq.Codec` and the same transformers SSL model on CPU - so every difference would be facade glue."""
import dataclasses

import pytest
import torch

from oracle import hcodec_ref as R
from oracle import ref_shim
from oracle import ref_tokenizer_shim as RT
from oracle import ssl_ref as S
from unified_audio_amd import synth

pytestmark = pytest.mark.skipif(not RT.reference_available(), reason="/root/reference is only mounted in the build container")


class _CodecOnCPU:
    """What the facade needs of a Codec (spec / encode / decode / device), served by the reference's vq.Codec."""

    def __init__(self, ref_codec, spec):
        self.ref, self.spec, self.device = ref_codec, spec, torch.device("cpu")

    def encode(self, x, feat, threshold=0.0):
        if self.spec.adaptive:
            return self.ref.encode(x, feat, threshold=threshold) if threshold else self.ref.encode(x, feat)
        if self.spec.version == 20:
            return self.ref.encode(x.squeeze(1) if x.dim() == 3 else x, feat)
        return self.ref.encode(x, feat)

    def decode(self, ac, sc, token_lengths=None):
        return self.ref.decode(ac, sc, token_lengths) if self.spec.adaptive else self.ref.decode(ac, sc)


def _ssl(kind, hidden, layers, stable, seed):
    from transformers import HubertModel, Wav2Vec2Model

    spec = S.SSLSpec(conv_dim=(32,) * 7, hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=4, intermediate_size=64,
                     num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=2, conv_bias=stable,
                     feat_extract_norm="layer" if stable else "group", do_stable_layer_norm=stable)
    model = {"hubert": HubertModel, "wav2vec2": Wav2Vec2Model}[kind](S.hf_config(spec, kind)).eval()
    model.load_state_dict(S.synth_state_dict(seed, spec, kind), strict=False)
    return model


def test_tokenizer_10_matches_reference_class():
    import unified_audio_amd as qa

    codec = ref_shim.load_state(ref_shim.load_reference_codec("1.0"), synth.hcodec10_state_dict(3))
    fx = _ssl("hubert", 768, 2, False, 31)
    ref = RT.load_reference_tokenizer("1.0", codec, fx)
    ours = qa.HCodecTokenizer(model=_CodecOnCPU(codec, qa.SPEC_10), feature_extractor=fx, device="cpu")
    wav = synth.synth_wav(5, 2, 640 * 7 + 123)
    assert torch.equal(ours.pad_wav(wav), ref.pad_wav(wav)) and ours.hop_length == ref.hop_length
    assert torch.equal(ours.extract_wav2vec2_features(ref.pad_wav(wav)), ref.extract_wav2vec2_features(ref.pad_wav(wav)))
    ac, sc = ours.tokenize(wav)
    rac, rsc = ref.tokenize(wav)
    assert torch.equal(ac, rac) and torch.equal(sc, rsc)
    assert torch.equal(ours.detokenize(ac, sc), ref.detokenize(rac, rsc))


def test_tokenizer_15_matches_reference_class():
    import unified_audio_amd as qa

    spec = dataclasses.replace(R.SPEC_15, agg_layers=1, bt_layers=1, threshold=0.7)
    codec = ref_shim.load_state(ref_shim.load_reference_codec("1.5", spec), synth.hcodec10_state_dict(4, spec))
    fx = _ssl("wav2vec2", 1024, 17, True, 32)  # hidden_states[16] must exist (HCodec-1.5/audio_tokenizer.py:58-61)
    cfg = ref_shim._config_15(spec)
    ref = RT.load_reference_tokenizer("1.5", codec, fx, cfg)
    pspec = qa.HCodecSpec(**{f: getattr(spec, f) for f in spec.__dataclass_fields__})
    ours = qa.HCodecTokenizer(model=_CodecOnCPU(codec, pspec), feature_extractor=fx, config=cfg, spec=pspec, device="cpu")
    wav = synth.synth_wav(6, 2, 640 * 9 + 77)
    assert torch.equal(ours.pad_wav(wav), ref.pad_wav(wav))
    assert torch.equal(ours.extract_wav2vec2_features(ref.pad_wav(wav)), ref.extract_wav2vec2_features(ref.pad_wav(wav)))
    codes, rcodes = ours.tokenize(wav), ref.tokenize(wav)
    assert set(codes) == set(rcodes) == {"acoustic_codes", "semantic_codes"}
    assert all(torch.equal(codes[k], rcodes[k]) for k in codes)
    assert torch.equal(ours.detokenize(**codes), ref.detokenize(**rcodes))


def test_tokenizer_20_matches_reference_class():
    """2.0: hop 3840 from the YAML, wav handed to encode WITHOUT the channel dimension (audio_tokenizer.py:73), tuple result.  The
    facade's own Resample runs on the device (tests/test_boundary_gpu.py); here the features of the reference's
    extract_ssl_features are passed in."""
    import unified_audio_amd as qa
    from oracle import hcodec20_ref as R20
    from oracle.gen_golden import SPEC20_SMALL

    o = R20.HCodec20Spec(**SPEC20_SMALL)
    codec = ref_shim.load_state(ref_shim.load_reference_codec("2.0", o), synth.hcodec20_state_dict(7, o))
    fx = _ssl("hubert", o.sem_in, 2, False, 33)
    cfg = ref_shim._config_20(o)
    ref = RT.load_reference_tokenizer("2.0", codec, fx, cfg)
    pspec = qa.HCodecSpec(version=20, enc_dim=o.enc_dim, enc_inter=o.enc_inter, enc_convnext_layers=o.enc_convnext_layers,
                          enc_layers=o.enc_transformer_layers, frame_stride=o.stride, tr_inter_cap=o.tr_inter_cap, dimension=o.dimension,
                          code_dim=o.dimension, sem_in=o.sem_in, sem_ch=o.sem_ch, sem_strides=o.sem_strides, codebook_size=o.codebook_size,
                          num_quantizers=o.num_quantizers, dec_dim=o.dec_dim, dec_inter=o.dec_inter, dec_heads=o.dec_dim // 64,
                          dec_layers=o.dec_transformer_layers, convnext_layers=o.dec_convnext_layers, n_fft=o.n_fft, hop=o.hop)
    ours = qa.HCodecTokenizer(model=_CodecOnCPU(codec, pspec), feature_extractor=fx, config=cfg, spec=pspec, device="cpu")
    assert ours.hop_length == ref.hop_length == 3840 and ours.sampling_rate == 48000
    wav = synth.synth_wav_fullband(8, 2, 3840 * 3 + 1000)
    assert torch.equal(ours.pad_wav(wav), ref.pad_wav(wav))
    feats = ref.extract_ssl_features(ref.pad_wav(wav))
    ac, sc = ours.tokenize(wav, feats=feats)
    rac, rsc = ref.tokenize(wav)
    assert torch.equal(ac, rac) and torch.equal(sc, rsc)
    assert torch.equal(ours.detokenize(ac, sc), ref.detokenize(rac, rsc))
