"""Host-side boundary behaviour that needs no device: the reference's YAML configurations map to the built-in specs, and the
facades keep nn.Module manners (QuarkAudio-HCodec/HCodec-1.0/audio_tokenizer.py:18-33, HCodec-1.5/...:38-51, HCodec-2.0/...:19-46)."""
import os

import pytest
import torch

from oracle import ref_shim

REF = os.path.join(ref_shim.REFERENCE_ROOT, "QuarkAudio-HCodec")
live = pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference is only mounted in the build container")


@live
def test_reference_yaml_configs_give_the_builtin_specs(qa_lib):
    import yaml

    import unified_audio_amd as qa
    from unified_audio_amd.hcodec import _spec_from_config

    c15 = yaml.safe_load(open(os.path.join(REF, "HCodec-1.5", "conf", "config_adaptive_v3.yaml")))
    assert _spec_from_config(c15) == qa.SPEC_15
    c20 = yaml.safe_load(open(os.path.join(REF, "HCodec-2.0", "conf", "large_12.5hz_config.yaml")))
    assert _spec_from_config(c20) == qa.SPEC_20
    # the reference's positional constructor calls (HCodec-1.5/audio_tokenizer.py:43, HCodec-2.0/audio_tokenizer.py:27-33)
    m15 = qa.Codec(c15["encoder_config"], c15["decoder_config"], c15["quantizer_config"], c15["adaptive_config"])
    m20 = qa.Codec(c20["encoder_config"], c20["decoder_config"], c20["quantizer_config"], c20["semantic_encoder_config"],
                   c20["semantic_decoder_config"])
    assert m15.spec == qa.SPEC_15 and m20.spec == qa.SPEC_20 and qa.Codec(None, None, None).spec == qa.SPEC_10


def test_codec_is_an_inference_only_module(qa_lib):
    import unified_audio_amd as qa

    m = qa.Codec(None, None, None)
    assert isinstance(m, torch.nn.Module) and m.eval() is m and m.requires_grad_(False) is m and m.float() is m
    assert list(m.parameters()) == [] and m.state_dict() == {}
    for bad in (lambda: m.train(), lambda: m.half(), lambda: m.to("cpu"), lambda: m.to(torch.bfloat16), lambda: m.cpu(),
                lambda: m(torch.zeros(1))):
        with pytest.raises(qa.QuarkAudioError):
            bad()
    with pytest.raises(qa.QuarkAudioError):  # no weights yet
        m.encode(torch.zeros(1, 1, 640), torch.zeros(1, 768, 2))
    with pytest.raises(ValueError):
        qa.HCodecTokenizer()


def test_eval_reaches_a_pytorch_feature_extractor_and_training_flags_follow(qa_lib):
    """the facades are nn.Modules - eval() must set `training` and recurse, so a HuBERT / XLSR module handed over in
    train mode loses its dropout / layerdrop like the reference's (audio_tokenizer.py:28-29 calls .eval() on it)."""
    import unified_audio_amd as qa

    fx = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Dropout(0.5))
    fx.train()
    codec = qa.Codec(None, None, None)
    assert codec.eval().training is False
    tok = qa.HCodecTokenizer(model=codec, feature_extractor=fx)
    assert fx.training is False and fx[1].training is False  # put in eval mode at construction
    fx.train()
    assert tok.eval() is tok and tok.training is False and fx[1].training is False and codec.training is False
    with pytest.raises(qa.QuarkAudioError):
        tok.train()
    assert qa.BiCodec().eval().training is False


def test_tensor_table_accepts_reduced_precision_checkpoints_and_skips_integer_buffers(qa_lib):
    from unified_audio_amd import _lib

    sd = {"a.weight": torch.randn(3, 4, dtype=torch.float16), "b.weight": torch.randn(5, dtype=torch.bfloat16),
          "c.weight": torch.randn(2, 2, dtype=torch.float64), "d.initted": torch.tensor([True]),
          "e.num_batches_tracked": torch.tensor(3), "f.weight": torch.randn(7)}
    arr, n, (keep, names) = _lib.tensor_table(sd)
    assert n == 4 and sorted(x.decode() for x in names) == ["a.weight", "b.weight", "c.weight", "f.weight"]
    assert all(t.dtype == torch.float32 and t.is_contiguous() for t in keep)
    assert torch.equal(keep[0], sd["a.weight"].float())


def test_streaming_transformer_rejects_configurations_it_does_not_implement(qa_lib):
    """Only the H-Codec 1.5 configuration of the mimi StreamingTransformer (transformer.py:722-736) is offered; anything else
    fails at construction, not silently at run time."""
    import unified_audio_amd as qa

    for kw in (dict(positional_embedding="sin"), dict(gating="silu"), dict(norm="rms_norm"), dict(layer_scale=None),
               dict(weights_per_step=4), dict(max_period=100.0)):
        with pytest.raises(qa.QuarkAudioError):
            qa.StreamingTransformer(64, 2, 1, 128, **kw)
    m = qa.StreamingTransformer(64, 2, 1, 128, causal=True, context=4)
    assert not m.is_streaming and m.streaming_offset == -1
    with pytest.raises(qa.QuarkAudioError, match="no weights"):
        m(torch.zeros(1, 2, 64))


def test_causal_flags_are_read_from_the_reference_yaml(qa_lib):
    """`causal:` / `context_frames:` / `context:` of config_adaptive_v3.yaml:84-105 reach the spec (shipped: causal false)."""
    import copy

    import yaml

    from oracle import ref_shim
    from unified_audio_amd.hcodec import _spec_from_config

    if not ref_shim.reference_available():
        pytest.skip("/root/reference is only mounted in the build container")
    cfg = yaml.safe_load(open(os.path.join(ref_shim.REFERENCE_ROOT, "QuarkAudio-HCodec/HCodec-1.5/conf/config_adaptive_v3.yaml")))
    spec = _spec_from_config(cfg)
    assert (spec.agg_causal, spec.agg_context, spec.bt_causal, spec.bt_context, spec.causal) == (False, 16, False, 16, False)
    cfg2 = copy.deepcopy(cfg)
    cfg2["adaptive_config"]["aggregators"]["semantic_aggregator"].update(causal=True, context_frames=9)
    cfg2["adaptive_config"]["transformer_kwargs"].update(causal=True, context=11)
    spec2 = _spec_from_config(cfg2)
    assert (spec2.agg_causal, spec2.agg_context, spec2.bt_causal, spec2.bt_context) == (True, 9, True, 11)
