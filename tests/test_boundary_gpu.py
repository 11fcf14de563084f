"""Boundary behaviour on the device: per-call threshold of H-Codec 1.5, 48 -> 16 kHz Resample of H-Codec 2.0, the range check of
decode, both kinds of SSL front-end behind HCodecTokenizer, nn.Module manners of the facades."""
import dataclasses
import types

import pytest
import torch

from oracle import hcodec15_ref as R15
from oracle import hcodec_ref as R
from oracle import resample_ref as RR
from oracle import ssl_ref as S
from oracle import synth
from tests.util import MINI, rel_err

pytestmark = pytest.mark.gpu


def _codec15(ospec, seed, device):
    import unified_audio_amd as qa

    sd = synth.hcodec10_state_dict(seed, ospec)
    kw = {f: getattr(ospec, f) for f in ospec.__dataclass_fields__}
    return sd, qa.Codec(None, None, None, spec=qa.HCodecSpec(**kw), device=device).load_state_dict(sd)


def test_per_call_threshold_overrides_the_models(qa_lib, gpu_device):
    """Codec.encode(..., threshold=t) (codec_adaptive.py:150-158): t <= 0 keeps manual_threshold, t in (0, 1] replaces it for
    this call only - the grouping must equal the oracle's at that threshold."""
    ospec = dataclasses.replace(R.SPEC_15, agg_layers=1, bt_layers=1, threshold=0.6)
    sd, codec = _codec15(ospec, 17, gpu_device)
    wav, feat = synth.synth_wav(3, 2, 640 * 40), synth.synth_feat(4, 2, 80, 1024)
    K = ospec.codebook_size
    lens = {}
    for thr in (0.0, 0.72, 0.9, 0.0):
        got = codec.encode(wav.to(gpu_device).unsqueeze(1), feat.to(gpu_device), threshold=thr)
        ref = R15.encode(sd, wav.unsqueeze(1), feat, dataclasses.replace(ospec, threshold=thr or 0.6))
        assert got["semantic_codes"].shape == ref["semantic_codes"].shape
        assert torch.equal(got["semantic_codes"].cpu() // K, ref["semantic_codes"] // K)
        lens[thr] = got["semantic_codes"].shape[-1]
    assert lens[0.9] > lens[0.72] >= lens[0.0]  # a higher threshold splits more
    with pytest.raises(AssertionError):
        codec.encode(wav.to(gpu_device).unsqueeze(1), feat.to(gpu_device), threshold=1.5)


@pytest.mark.parametrize("orig,T", [(48000, 3840 * 5), (48000, 3841), (44100, 4410)])
def test_resample_matches_the_torchaudio_restatement(qa_lib, gpu_device, orig, T):
    import ctypes as C

    from unified_audio_amd import _lib

    g = torch.Generator().manual_seed(orig + T)
    wav = torch.randn(3, T, generator=g)
    want = RR.resample(wav, orig, 16000)
    n = qa_lib.qa_resample_length(T, orig, 16000)
    assert n == want.shape[-1]
    x = wav.to(gpu_device)
    out = torch.full((3, n), float("nan"), device=gpu_device)
    _lib.check(qa_lib.qa_resample(x.data_ptr(), 3, T, orig, 16000, out.data_ptr(), None))
    torch.cuda.synchronize()
    assert float((out.cpu() - want).abs().max()) < 2e-6 * float(want.abs().max()) + 1e-6


def test_decode_range_check_is_one_sync_and_optional(qa_lib, gpu_device):
    import unified_audio_amd as qa

    sd = synth.hcodec10_state_dict(3, R.HCodecSpec(**MINI))
    codec = qa.Codec(None, None, None, spec=qa.HCodecSpec(**MINI), device=gpu_device).load_state_dict(sd)
    ok = torch.randint(0, 64, (2, 3, 5), dtype=torch.int64)
    bad = ok.clone()
    bad[1, 2, 4] = 64
    neg = ok.clone()
    neg[0, 0, 0] = -2
    assert torch.isfinite(codec.decode(ok, ok)).all()
    for a, s in ((bad, ok), (ok, bad), (neg, ok)):
        with pytest.raises(IndexError):
            codec.decode(a, s)
    codec.check_codes = False  # unchecked: indices are clamped by the kernels, nothing faults
    assert torch.isfinite(codec.decode(bad, neg)).all()


def test_decode_treats_minus_one_as_a_dropped_code(qa_lib, gpu_device):
    """the third-party get_output_from_indices behind Codec.decode (vq/codec.py:183-184) masks -1 - a code dropped by
    quantize dropout - to a zero vector; so does the HIP decode, against the oracle through the stand-in that restates upstream's
    masking: trailing stages dropped on some frames of one stream, a whole frame dropped in the other."""
    import unified_audio_amd as qa

    ospec = R.HCodecSpec(**MINI)
    sd = synth.hcodec10_state_dict(3, ospec)
    codec = qa.Codec(None, None, None, spec=qa.HCodecSpec(**MINI), device=gpu_device).load_state_dict(sd)
    gen = torch.Generator().manual_seed(11)
    ac = torch.randint(0, 64, (2, 3, 7), dtype=torch.int64, generator=gen)
    sc = torch.randint(0, 64, (2, 3, 7), dtype=torch.int64, generator=gen)
    ac[0, 1:, 2] = -1
    ac[1, 2, :] = -1
    sc[1, :, 5] = -1
    got = codec.decode(ac, sc).cpu()
    with torch.no_grad():
        want = R.decode(sd, ac, sc, ospec)
    assert float((got - want).pow(2).mean().sqrt()) < 1e-5 * float(want.pow(2).mean().sqrt()) + 1e-7
    full = codec.decode(ac.clamp(min=0), sc.clamp(min=0)).cpu()
    assert not torch.equal(full, got)  # the dropped stages really are missing from the sum


class _RefStyleExtractor(torch.nn.Module):
    """Stand-in with the reference's call convention (transformers HubertModel / Wav2Vec2Model): called on the padded wave,
    returns an object with `.hidden_states`; computed by the CPU oracle, which is pinned to the transformers classes."""

    def __init__(self, sd, spec):
        super().__init__()
        self.sd, self.spec = sd, dataclasses.replace(spec, pad=0)  # the tokenizer pads (160, 160) itself for this kind

    def forward(self, wavs, output_hidden_states=True):
        with torch.no_grad():
            hs = S.hidden_states(self.sd, wavs.cpu(), self.spec)
        return types.SimpleNamespace(hidden_states=tuple(h.to(wavs.device) for h in hs))


@pytest.mark.parametrize("version", ["1.0", "1.5"])
def test_tokenizer_feature_path_is_the_same_for_both_extractor_kinds(qa_lib, gpu_device, version):
    """with the reference's own PyTorch extractor the 1.5 tokenizer must average hidden states 11 / 14 / 16
    (HCodec-1.5/audio_tokenizer.py:58-61), not all of them - and give what the HIP front-end (select baked in) gives."""
    import unified_audio_amd as qa

    if version == "1.5":
        ospec = S.SSLSpec(conv_dim=(32,) * 7, hidden_size=96, num_hidden_layers=17, num_attention_heads=3, intermediate_size=128,
                          num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=2, conv_bias=True, feat_extract_norm="layer",
                          do_stable_layer_norm=True, select=(11, 14, 16))
        kind, cspec = "wav2vec2", dataclasses.replace(R.HCodecSpec(**{**MINI, "sem_in": 64}), adaptive=True, agg_layers=1, bt_layers=1,
                                                     agg_heads=2, bt_heads=2, agg_ff=128, bt_ff=128)
    else:
        ospec = S.SSLSpec(conv_dim=(32,) * 7, hidden_size=96, num_hidden_layers=3, num_attention_heads=3, intermediate_size=128,
                          num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=2)
        kind, cspec = "hubert", R.HCodecSpec(**MINI)
    ssl_sd = S.synth_state_dict(5, ospec, kind)
    kw = {f: getattr(ospec, f) for f in ospec.__dataclass_fields__}
    hip_fx = qa.SSLFeatureExtractor(qa.SSLSpec(**kw), device=gpu_device).load_state_dict(ssl_sd)
    ckw = {f: getattr(cspec, f) for f in cspec.__dataclass_fields__}
    sd = synth.hcodec10_state_dict(9, cspec)
    tok_hip = qa.HCodecTokenizer(state_dict=sd, spec=qa.HCodecSpec(**ckw), device=gpu_device, feature_extractor=hip_fx)
    tok_ref = qa.HCodecTokenizer(state_dict=sd, spec=qa.HCodecSpec(**ckw), device=gpu_device,
                                 feature_extractor=_RefStyleExtractor(ssl_sd, ospec))
    assert isinstance(tok_ref, torch.nn.Module) and tok_ref.select_layers == ((11, 14, 16) if version == "1.5" else None)
    wav = (torch.randn(2, 4000, generator=torch.Generator().manual_seed(1)) * 0.2).to(gpu_device)
    f_hip, f_ref = tok_hip.extract_wav2vec2_features(wav).cpu(), tok_ref.extract_wav2vec2_features(wav).cpu()
    assert f_hip.shape == f_ref.shape
    far = f_ref.abs() > 0.05  # |x|^0.3 has unbounded slope at 0
    assert float((f_hip - f_ref)[far].abs().max()) < 2e-3 and rel_err(f_hip, f_ref) < 5e-3
    # a mismatched HIP front-end (all layers for a 1.5 tokenizer) is refused instead of silently averaging the wrong states
    if version == "1.5":
        wrong = qa.SSLFeatureExtractor(qa.SSLSpec(**{**kw, "select": ()}), device=gpu_device).load_state_dict(ssl_sd)
        tok_hip.feature_extractor = wrong
        with pytest.raises(qa.QuarkAudioError):
            tok_hip.extract_wav2vec2_features(wav)


def test_hcodec20_tokenizer_resamples_before_the_ssl_model(qa_lib, gpu_device):
    """tokenize(wav48k) end to end for the 2.0 flavour: 3840-sample hop, Resample(48000, 16000), HuBERT-style front-end: the
    semantic frame count must line up with the acoustic one (r01: 3x too many frames, encode failed)."""
    import unified_audio_amd as qa
    from oracle import hcodec20_ref as R20
    from oracle.gen_golden import SPEC20_SMALL

    o = R20.HCodec20Spec(**SPEC20_SMALL)
    sd = synth.hcodec20_state_dict(7, o)
    pspec = qa.HCodecSpec(version=20, enc_dim=o.enc_dim, enc_inter=o.enc_inter, enc_convnext_layers=o.enc_convnext_layers,
                          enc_layers=o.enc_transformer_layers, frame_stride=o.stride, tr_inter_cap=o.tr_inter_cap, dimension=o.dimension,
                          code_dim=o.dimension, sem_in=o.sem_in, sem_ch=o.sem_ch, sem_strides=o.sem_strides, codebook_size=o.codebook_size,
                          num_quantizers=o.num_quantizers, dec_dim=o.dec_dim, dec_inter=o.dec_inter, dec_heads=o.dec_dim // 64,
                          dec_layers=o.dec_transformer_layers, convnext_layers=o.dec_convnext_layers, n_fft=o.n_fft, hop=o.hop)
    assert o.sem_in == 64  # one positional-conv group of 64 channels (the library supports 48- and 64-channel groups)
    ospec = S.SSLSpec(conv_dim=(32,) * 7, hidden_size=o.sem_in, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128,
                      num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=1)
    ssl_sd = S.synth_state_dict(5, ospec, "hubert")
    kw = {f: getattr(ospec, f) for f in ospec.__dataclass_fields__}
    fx = qa.SSLFeatureExtractor(qa.SSLSpec(**kw), device=gpu_device).load_state_dict(ssl_sd)
    tok = qa.HCodecTokenizer(state_dict=sd, spec=pspec, device=gpu_device, feature_extractor=fx)
    assert tok.sampling_rate == 48000 and tok.hop_length == 3840
    wav = synth.synth_wav_fullband(8, 2, 3840 * 6 + 100)
    padded = tok.pad_wav(wav)
    feats = tok.extract_ssl_features(padded.to(gpu_device))
    want = S.extract_features(ssl_sd, RR.resample(padded, 48000, 16000), ospec)
    assert feats.shape == want.shape and feats.shape[1] == padded.shape[-1] // o.hop  # 50 Hz features for a 960-sample STFT hop
    far = want.abs() > 0.05
    assert float((feats.cpu() - want)[far].abs().max()) < 2e-3
    ac, sc = tok.tokenize(wav)
    assert ac.shape == sc.shape == (2, o.num_quantizers, padded.shape[-1] // 3840)
    assert tok.detokenize(ac, sc).shape == (2, padded.shape[-1])


def test_facades_move_between_device_spellings(qa_lib, gpu_device):
    import unified_audio_amd as qa

    sd = synth.hcodec10_state_dict(3, R.HCodecSpec(**MINI))
    codec = qa.Codec(None, None, None, spec=qa.HCodecSpec(**MINI), device=gpu_device).load_state_dict(sd)
    assert codec.to("cuda") is codec and codec.to(gpu_device) is codec and codec.cuda() is codec and codec.eval() is codec
    assert set(codec.state_dict()) == set(sd)
    tok = qa.HCodecTokenizer(state_dict=sd, spec=qa.HCodecSpec(**MINI), device="cpu")  # the 2.0 signature's default device
    assert tok.device.type == "cuda" and tok.to(gpu_device) is tok
