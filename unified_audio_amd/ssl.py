"""Python mirror of the reference's SSL feature extraction, backed by libquarkaudio_hip.so.

    SSLFeatureExtractor(wavs)  <->  HCodecTokenizer.extract_wav2vec2_features
                                    (QuarkAudio-HCodec/HCodec-1.0/audio_tokenizer.py:35-48: hubert_base, mean of all hidden states;
                                     HCodec-1.5/audio_tokenizer.py:53-67: wav2vec2-large-xlsr-53, hidden states 11, 14, 16)
                               <->  Model.extract_semantic_features (QuarkAudio-UniSE/model/model.py:38-51: wavlm-base-plus,
                                     mean of all hidden states, no compression) with SPEC_WAVLM_BASE_PLUS

Weights come as the HF model's state_dict (`AutoModel.from_pretrained(...).state_dict()`), keys unchanged.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _lib


@dataclasses.dataclass(frozen=True)
class SSLSpec:
    """Architecture constants (transformers HubertConfig / Wav2Vec2Config field names in the comments)."""
    conv_dim: Tuple[int, ...] = (512,) * 7              # conv_dim
    conv_kernel: Tuple[int, ...] = (10, 3, 3, 3, 3, 2, 2)  # conv_kernel
    conv_stride: Tuple[int, ...] = (5, 2, 2, 2, 2, 2, 2)   # conv_stride
    conv_bias: bool = False                             # conv_bias
    feat_extract_norm: str = "group"                    # feat_extract_norm: "group" | "layer"
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    do_stable_layer_norm: bool = False
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16
    layer_norm_eps: float = 1e-5
    pad: int = 160                                      # F.pad(wavs, (160, 160))
    select: Tuple[int, ...] = ()                        # hidden_states averaged; () = all
    compress_exponent: float = 0.3                      # sign * |x| ** 0.3; 0 = off
    num_buckets: int = 0                                # WavLM gated relative position bias: 320 (0 = HuBERT / wav2vec2)
    max_bucket_distance: int = 800

    def to_c(self) -> _lib.qa_ssl_spec:
        s = _lib.qa_ssl_spec()
        s.n_conv = len(self.conv_dim)
        for i in range(s.n_conv):
            s.conv_dim[i], s.conv_kernel[i], s.conv_stride[i] = self.conv_dim[i], self.conv_kernel[i], self.conv_stride[i]
        s.conv_bias, s.feat_norm_layer = int(self.conv_bias), int(self.feat_extract_norm == "layer")
        s.hidden, s.n_layers, s.n_heads, s.intermediate = (self.hidden_size, self.num_hidden_layers, self.num_attention_heads,
                                                           self.intermediate_size)
        s.stable_layer_norm = int(self.do_stable_layer_norm)
        s.pos_kernel, s.pos_groups, s.pad = self.num_conv_pos_embeddings, self.num_conv_pos_embedding_groups, self.pad
        s.n_select = len(self.select)
        for i, v in enumerate(self.select):
            s.select[i] = v
        s.layer_norm_eps, s.compress_exponent = self.layer_norm_eps, self.compress_exponent
        s.rel_pos_buckets, s.rel_pos_max_distance = self.num_buckets, self.max_bucket_distance
        return s


SPEC_HUBERT_BASE = SSLSpec()  # bosonai/hubert_base as H-Codec 1.0 / 2.0 use it
SPEC_XLSR53 = SSLSpec(conv_bias=True, feat_extract_norm="layer", hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                      intermediate_size=4096, do_stable_layer_norm=True, select=(11, 14, 16))  # H-Codec 1.5


# microsoft/wavlm-base-plus as UniSE uses it (QuarkAudio-UniSE/model/model.py:30,38-51): mean of all hidden states, no compression
SPEC_WAVLM_BASE_PLUS = SSLSpec(num_buckets=320, max_bucket_distance=800, compress_exponent=0.0)


class SSLFeatureExtractor:
    def __init__(self, spec: SSLSpec = SPEC_HUBERT_BASE, *, device: str | torch.device = "cuda:0"):
        if spec.feat_extract_norm not in ("group", "layer"):
            raise ValueError(f"feat_extract_norm={spec.feat_extract_norm!r}")
        self.spec = spec
        self.device = torch.device(device)
        self._lib = _lib.load_library()
        self._handle = C.c_void_p()

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        _lib.require_device()
        self._free()
        table, n, keep = _lib.tensor_table(state_dict)
        handle = C.c_void_p()
        cspec = self.spec.to_c()
        _lib.check(self._lib.qa_ssl_create(C.byref(handle), C.byref(cspec), table, n, self.device.index or 0))
        del keep
        self._handle = handle
        return self

    @classmethod
    def from_pretrained(cls, path, spec: Optional[SSLSpec] = None, *, device: str | torch.device = "cuda:0",
                        default_spec: Optional[SSLSpec] = None) -> "SSLFeatureExtractor":
        """The offline stand-in for the reference's `AutoModel.from_pretrained("bosonai/hubert_base" | "facebook/wav2vec2-large-xlsr-53" |
        "microsoft/wavlm-base-plus")` (audio_tokenizer.py:28, HCodec-1.5/audio_tokenizer.py:47, model/model.py:30): `path` is a local
        Hugging Face snapshot directory (`model.safetensors` or `pytorch_model.bin`, `config.json` optional) or a single weight file
        (`.safetensors`, or a `torch.save`d state_dict).  `spec` fixes the architecture; without it the architecture is read from
        `config.json` (the HubertConfig / Wav2Vec2Config / WavLMConfig field names SSLSpec uses) and what the tokenizers add around the
        model - which hidden states are averaged, the |x|^0.3 compression - is taken from `default_spec` (SPEC_HUBERT_BASE when absent),
        which is also the architecture when there is no config.json."""
        import json
        import os

        def read(f):
            if f.endswith(".safetensors"):
                from safetensors.torch import load_file

                return load_file(f)
            sd = torch.load(f, map_location="cpu")
            return sd["state_dict"] if isinstance(sd, dict) and "state_dict" in sd and not torch.is_tensor(sd["state_dict"]) else sd

        path = os.fspath(path)
        if os.path.isdir(path):
            files = [os.path.join(path, n) for n in ("model.safetensors", "pytorch_model.bin") if os.path.isfile(os.path.join(path, n))]
            if not files:
                raise FileNotFoundError(f"{path}: neither model.safetensors nor pytorch_model.bin")
            sd = read(files[0])
            cfg_file = os.path.join(path, "config.json")
            if spec is None and os.path.isfile(cfg_file):
                with open(cfg_file) as f:
                    cfg = json.load(f)
                names = {f.name for f in dataclasses.fields(SSLSpec)} - {"pad", "select", "compress_exponent"}
                kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in cfg.items() if k in names}
                if cfg.get("model_type") == "wavlm":
                    kw.setdefault("num_buckets", 320)
                else:
                    kw["num_buckets"] = 0
                base = default_spec or SPEC_HUBERT_BASE
                spec = dataclasses.replace(base, **kw)
        else:
            sd = read(path)
        # a checkpoint saved from a task head (HubertForCTC, WavLMForXVector ...) prefixes the base model's entries
        for prefix in ("hubert.", "wav2vec2.", "wavlm."):
            if any(k.startswith(prefix) for k in sd):
                sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
                break
        return cls(spec or default_spec or SPEC_HUBERT_BASE, device=device).load_state_dict(sd)

    def eval(self):
        return self

    def frames(self, n_samples: int) -> int:
        self._require_loaded()
        n = self._lib.qa_ssl_frames(self._handle, int(n_samples))
        if n < 0:
            _lib.check(int(n))
        return int(n)

    @torch.no_grad()
    def __call__(self, wavs: torch.Tensor) -> torch.Tensor:
        """wavs float32 [B, T] on the device -> feats_mix float32 [B, frames, hidden] (what the reference transposes to (b, d, t))."""
        self._require_loaded()
        if wavs.dim() != 2:
            raise ValueError(f"wavs must be [B, T], got {tuple(wavs.shape)}")
        wavs = wavs.to(device=self.device, dtype=torch.float32).contiguous()
        B, T = wavs.shape
        out = torch.empty(B, self.frames(T), self.spec.hidden_size, device=self.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.qa_ssl_forward(self._handle, C.c_void_p(wavs.data_ptr()), B, T, C.c_void_p(out.data_ptr()),
                                            C.c_void_p(stream)))
        return out

    extract_wav2vec2_features = __call__

    def _require_loaded(self):
        if not self._handle.value:
            raise _lib.QuarkAudioError(-1, "SSLFeatureExtractor: call load_state_dict first")

    def _free(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.qa_ssl_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass
