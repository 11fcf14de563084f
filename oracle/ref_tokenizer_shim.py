"""TEST INFRASTRUCTURE ONLY - the reference's OWN three `HCodecTokenizer` classes (QuarkAudio-HCodec/HCodec-{1.0,1.5,2.0}/audio_tokenizer.py:
`pad_wav`, `extract_wav2vec2_features` / `extract_ssl_features`, `tokenize`, `detokenize`) imported from where they lie, so that the
facade `unified_audio_amd.HCodecTokenizer` can be pinned to them.  Their constructors download an SSL model from the hub and load a
checkpoint; the objects are therefore made with `__new__` and given the members the constructor would set (audio_tokenizer.py:21-31 /
:40-51 / :21-46): `model` = the reference's `vq.Codec` (oracle/ref_shim.py), `feature_extractor` = a transformers model the caller
builds, `hop_length` / `config` / `resample`.  Import-time stubs: librosa (only `load_wav` / `__main__` use it), torchaudio (stubs/;
its `Resample` is replaced by oracle/resample_ref.py's restatement - the one third-party piece that stays unpinned), the `.vq`
package of 1.0 resolved relative to a shim package.  The GPU box has no /root/reference: tests using this module skip there."""
from __future__ import annotations

import importlib.machinery
import importlib.util
import os
import sys
import types
import warnings

from torch import nn

from oracle import ref_shim
from oracle.ref_shim import REFERENCE_ROOT

_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stubs")


def reference_available() -> bool:
    return ref_shim.reference_available()


def _import_tokenizer_module(version: str):
    root = os.path.join(REFERENCE_ROOT, ref_shim._VERSIONS[version])
    for name in [m for m in sys.modules if m == "vq" or m.startswith("vq.") or m == "adaptive" or m.startswith("adaptive.")]:
        del sys.modules[name]
    pkg_name = "qa_ref_h" + version.replace(".", "")
    pkg = types.ModuleType(pkg_name)
    pkg.__path__ = [root]  # `from .vq import Codec` (1.0) resolves against the version's own directory
    sys.modules[pkg_name] = pkg
    had_librosa = "librosa" in sys.modules
    if not had_librosa:
        stub = types.ModuleType("librosa")
        stub.__spec__ = importlib.machinery.ModuleSpec("librosa", None)
        sys.modules["librosa"] = stub
    sys.path[:0] = [_STUBS, root]
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            spec = importlib.util.spec_from_file_location(pkg_name + ".audio_tokenizer", os.path.join(root, "audio_tokenizer.py"),
                                                          submodule_search_locations=None)
            mod = importlib.util.module_from_spec(spec)
            sys.modules[spec.name] = mod
            spec.loader.exec_module(mod)
    finally:
        sys.path.remove(_STUBS)
        sys.path.remove(root)
        if not had_librosa:
            del sys.modules["librosa"]
    return mod


class _OracleResample(nn.Module):
    """torchaudio.transforms.Resample(orig, new) stand-in (oracle/resample_ref.py)."""

    def __init__(self, orig_freq, new_freq):
        super().__init__()
        self.orig_freq, self.new_freq = orig_freq, new_freq

    def forward(self, x):
        from oracle import resample_ref

        return resample_ref.resample(x, self.orig_freq, self.new_freq)


def load_reference_tokenizer(version: str, codec, feature_extractor, config=None):
    """The reference's HCodecTokenizer of `version` around an already built reference `vq.Codec` and a transformers SSL model."""
    mod = _import_tokenizer_module(version)
    tok = mod.HCodecTokenizer.__new__(mod.HCodecTokenizer)
    nn.Module.__init__(tok)
    tok.model = codec.eval()
    tok.feature_extractor = feature_extractor.eval()
    if version == "1.0":
        tok.hop_length = 640  # audio_tokenizer.py:31
    elif version == "1.5":
        tok.config = config   # pad_wav reads config['encoder_config']['ratios'] (:71)
        tok.hop_length = 640  # :51
    else:
        tok.resample = _OracleResample(config["sampling_rate"], 16000)                                      # :44
        tok.hop_length = int(config["sampling_rate"] / config["encoder_config"]["target_frame_rate"])      # :46
    return tok.eval()
