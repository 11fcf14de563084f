"""TEST INFRASTRUCTURE ONLY - the reference's OWN BiCodec modules (QuarkAudio-UniSE/model/bicodec/modules/*) imported from
/root/reference (this container only) and assembled exactly as `BiCodec.detokenize` assembles them
(model/bicodec/bicodec.py:182-199), so that oracle/bicodec_ref.py can be pinned to them and golden vectors generated
(oracle/gen_golden_bicodec.py).  `bicodec.py` itself is not imported: it needs omegaconf and a Spark-TTS `config.yaml` /
`model.safetensors` that are not in the tree; the four sub-modules it builds are constructed here from an explicit spec
(the published Spark-TTS BiCodec shapes by default, see oracle/bicodec_ref.BiCodecSpec).  `einx` (one `get_at` call) is stubbed.
"""
from __future__ import annotations

import importlib
import os
import sys
import types
import warnings

import torch
from torch import nn

from oracle.ref_shim import REFERENCE_ROOT

_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stubs")
_BICODEC = os.path.join(REFERENCE_ROOT, "QuarkAudio-UniSE", "model", "bicodec")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(_BICODEC, "modules", "encoder_decoder", "wave_generator.py"))


def _pkg(name, path):
    if name not in sys.modules or not getattr(sys.modules[name], "_qa_shim", False):
        m = types.ModuleType(name)
        m.__path__ = [path]
        m._qa_shim = True
        sys.modules[name] = m


def _import(mod):
    _pkg("model", os.path.join(REFERENCE_ROOT, "QuarkAudio-UniSE", "model"))
    _pkg("model.bicodec", _BICODEC)
    sys.path.insert(0, _STUBS)
    try:
        return importlib.import_module("model.bicodec.modules." + mod)
    finally:
        sys.path.remove(_STUBS)


class ReferenceDetokenizer(nn.Module):
    """quantizer / speaker_encoder / prenet / decoder with the attribute names BiCodec gives them (bicodec.py:61-67)."""

    def __init__(self, spec):
        super().__init__()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fvq = _import("vq.factorized_vector_quantize")
            spk = _import("speaker.speaker_encoder")
            dec = _import("encoder_decoder.feat_decoder")
            wav = _import("encoder_decoder.wave_generator")
            self.quantizer = fvq.FactorizedVectorQuantize(input_dim=spec.latent_dim, codebook_size=spec.codebook_size,
                                                          codebook_dim=spec.codebook_dim, commitment=0.25)
            self.speaker_encoder = spk.SpeakerEncoder(input_dim=spec.mel_dim, out_dim=spec.latent_dim, latent_dim=spec.spk_latent_dim,
                                                      token_num=spec.token_num, fsq_levels=list(spec.fsq_levels), fsq_num_quantizers=1)
            self.prenet = dec.Decoder(input_channels=spec.latent_dim, vocos_dim=spec.vocos_dim,
                                      vocos_intermediate_dim=spec.vocos_inter, vocos_num_layers=spec.vocos_layers,
                                      out_channels=spec.latent_dim, condition_dim=spec.latent_dim, sample_ratios=[1, 1],
                                      use_tanh_at_final=False)
            self.decoder = wav.WaveGenerator(input_channel=spec.latent_dim, channels=spec.gen_channels, rates=list(spec.rates),
                                             kernel_sizes=list(spec.kernel_sizes))

    @torch.no_grad()
    def detokenize(self, semantic_tokens, global_tokens):  # bicodec.py:193-199, verbatim
        z_q = self.quantizer.detokenize(semantic_tokens)
        d_vector = self.speaker_encoder.detokenize(global_tokens)
        x = self.prenet(z_q, d_vector)
        x = x + d_vector.unsqueeze(-1)
        wav_recon = self.decoder(x)
        return wav_recon


DETOK_PREFIXES = ("quantizer.codebook.", "quantizer.out_project.", "speaker_encoder.quantizer.project_out.", "speaker_encoder.project.",
                  "prenet.", "decoder.")


def load_reference_detokenizer(spec, sd=None):
    if not reference_available():
        raise RuntimeError(f"reference tree not found under {REFERENCE_ROOT}")
    m = ReferenceDetokenizer(spec).eval()
    if sd is not None:
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        bad = [k for k in missing if k.startswith(DETOK_PREFIXES)]  # encoder-side parts (ECAPA, perceiver, in_project) stay random
        assert not bad, bad
    return m
