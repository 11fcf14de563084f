"""TEST INFRASTRUCTURE ONLY - run the reference's OWN `Model.test_step` (QuarkAudio-UniSE/model/model.py:170-290: the 'se', 'tse' and
three-pass 'ss' branches with their wrap padding, 5 s segmentation, normalisation, enrollment tiling and `est.reshape(-1)[:len]`)
in this container, so that `unified_audio_amd.unise.UniSE` (the driver around the HIP components) can be pinned to it.

`Model.__init__` downloads WavLM and loads a Spark-TTS BiCodec checkpoint; neither exists offline, so the object is built without it
(`Model.__new__` + `nn.Module.__init__`) and given the three members `test_step` uses, each of them the reference's / the third
party's own class with seeded random weights:
    semantic_model  transformers.WavLMModel (what AutoModel.from_pretrained("microsoft/wavlm-base-plus") returns, model.py:30)
    dnn             the reference's LLM_SFT (oracle/ref_llm_shim.py)
    tokenizer       BiCodecTokenizer.detokenize (bicodec/audio_tokenizer.py:107-120: `self.model.detokenize(semantic_tokens,
                    global_tokens)`) over the reference's own BiCodec modules (oracle/ref_bicodec_shim.py)
Import-time stubs (never on the arithmetic path): pytorch_lightning (LightningModule = nn.Module), soundfile (`write` captures the
estimate instead of writing a file), torchaudio.functional.melscale_fbanks (the mel only contributes its frame count, llm_sft.py:108).
The GPU box has no /root/reference: tests using this module skip there.
"""
from __future__ import annotations

import importlib
import os
import sys
import types
import warnings

import torch
from torch import nn

from oracle import ref_bicodec_shim, ref_llm_shim
from oracle.ref_shim import REFERENCE_ROOT

_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stubs")
_UNISE = os.path.join(REFERENCE_ROOT, "QuarkAudio-UniSE")

WRITTEN = []  # (path, samples, samplerate) of every soundfile.write the reference's test_step issued


def reference_available() -> bool:
    return os.path.isfile(os.path.join(_UNISE, "model", "model.py")) and ref_llm_shim.reference_available() and \
        ref_bicodec_shim.reference_available()


class _RefBiCodecTokenizer(nn.Module):
    """bicodec/audio_tokenizer.py:107-120 over the reference's BiCodec modules (its `tokenize` side is not on this path)."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    @torch.no_grad()
    def detokenize(self, global_tokens, semantic_tokens):
        return self.model.detokenize(semantic_tokens, global_tokens)


def _stub_modules():
    if "pytorch_lightning" not in sys.modules:
        pl = types.ModuleType("pytorch_lightning")

        class LightningModule(nn.Module):
            def save_hyperparameters(self, *a, **k):
                pass

        pl.LightningModule = LightningModule
        sys.modules["pytorch_lightning"] = pl
    sf = types.ModuleType("soundfile")
    sf.write = lambda path, data, samplerate=16000, **k: WRITTEN.append((str(path), data, int(samplerate)))
    sys.modules["soundfile"] = sf


def _import_model_module():
    ref_llm_shim._import_llm_sft()  # registers the shim package `model` and the transformers compatibility patch
    if "model.bicodec" not in sys.modules or not hasattr(sys.modules["model.bicodec"], "BiCodecTokenizer"):
        pkg = sys.modules.get("model.bicodec")
        if pkg is None or not getattr(pkg, "_qa_shim", False):
            pkg = types.ModuleType("model.bicodec")
            pkg.__path__ = [os.path.join(_UNISE, "model", "bicodec")]
            pkg._qa_shim = True
            sys.modules["model.bicodec"] = pkg
        pkg.BiCodecTokenizer = _RefBiCodecTokenizer  # `from .bicodec import BiCodecTokenizer` (model.py:13) without omegaconf / checkpoints
    _stub_modules()
    sys.path.insert(0, _STUBS)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return importlib.import_module("model.model")
    finally:
        sys.path.remove(_STUBS)


def load_reference_model(semantic_model, dnn, detokenizer, save_dir="/tmp/qa_unise_ref"):
    """The reference's `Model` with the three members its test_step uses.  `detokenizer`: ref_bicodec_shim.ReferenceDetokenizer."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found under {REFERENCE_ROOT}")
    mod = _import_model_module()
    m = mod.Model.__new__(mod.Model)
    nn.Module.__init__(m)
    m.config = {"save_enhanced": save_dir}
    m.stft_conf = dict(hop_length=320, win_length=640, n_fft=640, n_mels=80)  # conf/config.yaml:124-128
    m.semantic_model = semantic_model.eval()
    m.dnn = dnn
    m.tokenizer = _RefBiCodecTokenizer(detokenizer)
    return m.eval()


@torch.no_grad()
def run_test_step(model, mode, src, enroll=None, name="utt"):
    """One call of the reference's test_step on its own batch tuple (dataloader/data_module.py: batch_size 1).  Returns the list of
    estimates it wrote (one for 'se' / 'tse', two - s1, s2 - for 'ss')."""
    del WRITTEN[:]
    batch = (mode, enroll, src, None, [16000], None, [name])
    model.test_step(batch, 0)
    return [torch.from_numpy(w[1]).clone() for w in WRITTEN]
