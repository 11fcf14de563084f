"""TEST INFRASTRUCTURE ONLY - import the reference's OWN modules from /root/reference (this container
only; the GPU box has no /root/reference) so the restatement in hcodec_ref.py can be validated against
them and golden vectors generated (oracle/gen_golden.py).

Recipe: put oracle/stubs (vector_quantize_pytorch, torchaudio) and the version's
directory on sys.path; the three H-Codec trees all expose a top-level package called `vq`, so
sys.modules is purged between versions.
"""
from __future__ import annotations

import os
import sys
import warnings

REFERENCE_ROOT = os.environ.get("QA_REFERENCE_ROOT", "/root/reference")
_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stubs")
_VERSIONS = {
    "1.0": "QuarkAudio-HCodec/HCodec-1.0",
    "1.5": "QuarkAudio-HCodec/HCodec-1.5",
    "2.0": "QuarkAudio-HCodec/HCodec-2.0",
}


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, _VERSIONS["1.0"], "vq"))


def _config_15(spec):
    import yaml

    cfg = yaml.safe_load(open(os.path.join(REFERENCE_ROOT, _VERSIONS["1.5"], "conf", "config_adaptive_v3.yaml")))
    ad = cfg["adaptive_config"]
    if spec is not None:  # the reference builds its stacks from this YAML, so smaller test variants are the reference's own code
        for k in ("semantic_aggregator", "acoustic_aggregator"):
            ad["aggregators"][k].update(num_layers=spec.agg_layers, num_heads=spec.agg_heads, dim_feedforward=spec.agg_ff,
                                        causal=spec.agg_causal, context_frames=spec.agg_context)
        ad["transformer_kwargs"].update(num_layers=spec.bt_layers, num_heads=spec.bt_heads, dim_feedforward=spec.bt_ff,
                                        causal=spec.bt_causal, context=spec.bt_context)
        ad["manual_threshold"] = spec.threshold
        ad["max_tokens_per_group"] = spec.max_tokens_per_group
    return cfg


def _config_20(spec):
    import yaml

    cfg = yaml.safe_load(open(os.path.join(REFERENCE_ROOT, _VERSIONS["2.0"], "conf", "large_12.5hz_config.yaml")))
    if spec is not None:  # the reference builds H-Codec 2.0 from this YAML: reduced sizes are its own code path
        rate = 50.0 / spec.stride
        cfg["encoder_config"].update(dim=spec.enc_dim, intermediate_dim=spec.enc_inter, dimension=spec.dimension, n_fft=spec.n_fft,
                                     hop_length=spec.hop, convnext_layers=spec.enc_convnext_layers,
                                     transformer_layers=spec.enc_transformer_layers, target_frame_rate=rate, causal=spec.causal)
        cfg["decoder_config"].update(input_channels=2 * spec.dimension, dim=spec.dec_dim, intermediate_dim=spec.dec_inter,
                                     convnext_layers=spec.dec_convnext_layers, transformer_layers=spec.dec_transformer_layers,
                                     n_fft=spec.n_fft, hop_length=spec.hop, target_frame_rate=rate, causal=spec.causal)
        cfg["quantizer_config"].update(dim=spec.dimension, codebook_size=spec.codebook_size, num_quantizers=spec.num_quantizers)
        for k, a, b in (("semantic_encoder_config", "input_channels", "encode_channels"),):
            cfg[k].update({a: spec.sem_in, b: spec.sem_ch, "out_channels": spec.dimension, "strides": list(spec.sem_strides),
                           "channel_ratios": [1] * len(spec.sem_strides)})
        cfg["semantic_decoder_config"].update(code_dim=spec.dimension, output_channels=spec.sem_in, decode_channels=spec.sem_ch,
                                              strides=list(spec.sem_strides), channel_ratios=[1] * len(spec.sem_strides))
    return cfg


def load_reference_codec(version: str = "1.0", spec=None):
    """Construct the reference `vq.Codec` (eval mode, random init) for the given H-Codec version."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found under {REFERENCE_ROOT}")
    for name in [m for m in sys.modules if m == "vq" or m.startswith("vq.") or m == "adaptive"
                 or m.startswith("adaptive.")]:
        del sys.modules[name]
    root = os.path.join(REFERENCE_ROOT, _VERSIONS[version])
    sys.path[:0] = [_STUBS, root]
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            from vq import Codec  # type: ignore
            if version == "1.0":
                model = Codec(None, None, None)
            elif version == "1.5":
                import contextlib
                import io

                cfg = _config_15(spec)
                with contextlib.redirect_stdout(io.StringIO()):
                    model = Codec(cfg["encoder_config"], cfg["decoder_config"], cfg["quantizer_config"], cfg["adaptive_config"])
            else:
                cfg = _config_20(spec)
                model = Codec(cfg["encoder_config"], cfg["decoder_config"], cfg["quantizer_config"],
                              cfg["semantic_encoder_config"], cfg["semantic_decoder_config"])
    finally:
        sys.path.remove(_STUBS)
        sys.path.remove(root)
    return model.eval()


def load_reference_mimi(d_model: int, num_heads: int, num_layers: int, dim_feedforward: int, causal: bool, context: int):
    """The reference's own StreamingTransformer (HCodec-1.5/adaptive/model_blocks/mimi/transformer.py:605-698) with the keyword
    arguments QueryTokenAggregator / the bottleneck pass (:722-736): rope, layer_norm, gating none, layer_scale 0.01."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found under {REFERENCE_ROOT}")
    for name in [m for m in sys.modules if m == "vq" or m.startswith("vq.") or m == "adaptive" or m.startswith("adaptive.")]:
        del sys.modules[name]
    root = os.path.join(REFERENCE_ROOT, _VERSIONS["1.5"])
    sys.path[:0] = [_STUBS, root]
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            from adaptive.model_blocks.mimi.transformer import StreamingTransformer  # type: ignore
            model = StreamingTransformer(d_model=d_model, num_heads=num_heads, num_layers=num_layers, dim_feedforward=dim_feedforward,
                                         causal=causal, context=context, positional_embedding="rope", max_period=10000,
                                         layer_scale=0.01, gating="none", norm="layer_norm")
    finally:
        sys.path.remove(_STUBS)
        sys.path.remove(root)
    return model.eval()


def make_causal_10(model):
    """Swap the H-Codec 1.0 encoder / decoder for the reference's OWN classes built with causal=True.  vq/codec.py:30-47
    hard-codes causal=False, but every block takes the flag (seanet.py:107, codec_decoder.py:23); the constructor arguments
    below are codec.py's with that one flag flipped, so the parameters (and state-dict keys) are identical."""
    import sys

    vq_codec = sys.modules[type(model).__module__]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model.encoder = vq_codec.CodecEncoder(
            causal=True, n_residual_layers=1, norm='weight_norm', pad_mode='reflect', lstm=2,
            dimension=512, channels=1, n_filters=32, ratios=[8, 5, 4, 2], activation='ELU',
            kernel_size=7, residual_kernel_size=3, last_kernel_size=7, dilation_base=2,
            true_skip=False, compress=2, use_transformer=True)
        model.decoder = vq_codec.CodecDecoder(input_channels=512 * 2, dim=768, intermediate_dim=2304, causal=True)
    return model.eval()


def load_state(model, sd, strict_subset: bool = True):
    """Load a synth state_dict; keys the synth generator leaves out (training-only semantic_decoder.*)
    keep their random init."""
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    if strict_subset:
        bad = [k for k in missing if not k.startswith("semantic_decoder.")]
        assert not bad, bad
    return model
