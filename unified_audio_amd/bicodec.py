"""Python mirror of the reference's BiCodec detokenizer, backed by libquarkaudio_hip.so.

    BiCodec.detokenize           <->  QuarkAudio-UniSE/model/bicodec/bicodec.py:182-199
    BiCodecTokenizer.detokenize  <->  QuarkAudio-UniSE/model/bicodec/audio_tokenizer.py (called at model/model.py:193,223)

Only the decode side is on the UniSE inference path (the LM produces the tokens); `tokenize` (wav2vec2-BERT features, ECAPA speaker
encoder) belongs to training / data preparation and is not offered.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Tuple

import torch

from . import _lib


@dataclass(frozen=True)
class BiCodecSpec:
    """`audio_tokenizer` section of the Spark-TTS BiCodec `config.yaml` the reference loads from `codec_ckpt_dir`
    (bicodec.py:80-87; the file itself is not in the reference tree)."""

    latent_dim: int = 1024
    codebook_size: int = 8192
    codebook_dim: int = 8
    mel_dim: int = 128  # encoder side only
    spk_latent_dim: int = 128
    token_num: int = 32
    fsq_levels: Tuple[int, ...] = (4, 4, 4, 4, 4, 4)
    vocos_dim: int = 384
    vocos_inter: int = 2048
    vocos_layers: int = 12
    gen_channels: int = 1536
    rates: Tuple[int, ...] = (8, 5, 4, 2)
    kernel_sizes: Tuple[int, ...] = (16, 11, 8, 4)

    @property
    def hop(self) -> int:
        h = 1
        for r in self.rates:
            h *= r
        return h

    @property
    def global_size(self) -> int:
        n = 1
        for lv in self.fsq_levels:
            n *= lv
        return n

    def to_c(self) -> "_lib.qa_bicodec_spec":
        s = _lib.qa_bicodec_spec()
        s.latent_dim, s.codebook_size, s.codebook_dim = self.latent_dim, self.codebook_size, self.codebook_dim
        s.spk_latent_dim, s.token_num, s.n_levels = self.spk_latent_dim, self.token_num, len(self.fsq_levels)
        for i, v in enumerate(self.fsq_levels):
            s.levels[i] = v
        s.vocos_dim, s.vocos_inter, s.vocos_layers = self.vocos_dim, self.vocos_inter, self.vocos_layers
        s.gen_channels, s.n_rates = self.gen_channels, len(self.rates)
        for i, (r, k) in enumerate(zip(self.rates, self.kernel_sizes)):
            s.rates[i], s.kernel_sizes[i] = r, k
        return s


SPEC_BICODEC = BiCodecSpec()


class BiCodec(torch.nn.Module):
    """`detokenize(semantic_tokens, global_tokens)` of the reference's BiCodec.  Weights in the reference's key layout
    (`BiCodec.state_dict()` / the `model.safetensors` of the checkpoint; encoder-side entries are ignored)."""

    def __init__(self, spec: BiCodecSpec = SPEC_BICODEC, *, device: str | torch.device = "cuda:0", check_tokens: bool = True):
        super().__init__()
        self.spec = spec
        self.device = torch.device(device)
        self.check_tokens = check_tokens
        self._lib = _lib.load_library()
        self._handle = C.c_void_p()

    @classmethod
    def load_from_checkpoint(cls, model_dir, device="cuda:0", spec: BiCodecSpec = SPEC_BICODEC, **kwargs) -> "BiCodec":
        """bicodec.py:70-115: `{model_dir}/model.safetensors` (the architecture comes from `spec`; config.yaml is not parsed)."""
        from safetensors.torch import load_file

        return cls(spec, device=device).load_state_dict(load_file(f"{model_dir}/model.safetensors"))

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = False, assign: bool = False):
        _lib.require_device()
        self._free()
        table, n, keep = _lib.tensor_table(state_dict)
        handle = C.c_void_p()
        cspec = self.spec.to_c()
        _lib.check(self._lib.qa_bicodec_create(C.byref(handle), C.byref(cspec), table, n, self.device.index or 0))
        del keep
        self._handle = handle
        return self

    def train(self, mode: bool = True):
        if mode:
            raise _lib.QuarkAudioError(-4, "unified_audio_amd.BiCodec is the inference path")
        return super().train(False)

    def remove_weight_norm(self):  # bicodec.py:113: weight norm is folded when the weights are loaded
        return self

    @torch.no_grad()
    def detokenize(self, semantic_tokens: torch.Tensor, global_tokens: torch.Tensor) -> torch.Tensor:
        """semantic_tokens [B, T] int64, global_tokens [B, 1, token_num] (or [B, token_num]) int64 -> wav [B, 1, T * hop] float32."""
        if not self._handle.value:
            raise _lib.QuarkAudioError(-3, "BiCodec has no weights: call load_state_dict first")
        sem = semantic_tokens.to(device=self.device, dtype=torch.int64).contiguous()
        glob = global_tokens.to(device=self.device, dtype=torch.int64).contiguous()
        if sem.dim() != 2:
            raise _lib.QuarkAudioError(-1, f"semantic_tokens must be [B, T], got {tuple(sem.shape)}")
        B, T = sem.shape
        glob = glob.reshape(B, -1)
        if glob.shape[1] != self.spec.token_num:
            raise _lib.QuarkAudioError(-1, f"global_tokens must hold {self.spec.token_num} tokens per item, got {tuple(global_tokens.shape)}")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if self.check_tokens:  # F.embedding / the implicit FSQ codebook gather would refuse out-of-range ids: ONE check (one host
            # sync) for both tensors - the global ids are scaled onto the semantic range so that a single limit serves
            lim_s, lim_g = self.spec.codebook_size, self.spec.global_size
            both = torch.cat([sem.reshape(-1), torch.where((glob >= 0) & (glob < lim_g), 0, lim_s).reshape(-1)])
            bad = C.c_int64(0)
            _lib.check(self._lib.qa_codes_check(both.data_ptr(), both.numel(), lim_s, C.byref(bad), stream))
            if bad.value:
                n_g = int(((glob < 0) | (glob >= lim_g)).sum())
                raise IndexError(f"{bad.value - n_g} semantic_tokens out of range [0, {lim_s}), {n_g} global_tokens out of range [0, {lim_g})")
        wav = torch.empty((B, 1, T * self.spec.hop), dtype=torch.float32, device=self.device)
        _lib.check(self._lib.qa_bicodec_detokenize(self._handle, sem.data_ptr(), glob.data_ptr(), B, T, wav.data_ptr(), stream))
        return wav

    def enable_taps(self, on: bool = True):
        _lib.check(self._lib.qa_bicodec_enable_taps(self._handle, int(on)))
        return self

    def tap(self, name: str) -> torch.Tensor:
        n = self._lib.qa_bicodec_tap(self._handle, name.encode(), None, 0, None)
        if n < 0:
            _lib.check(int(n))
        out = torch.empty(int(n), dtype=torch.float32, device=self.device)
        n2 = self._lib.qa_bicodec_tap(self._handle, name.encode(), out.data_ptr(), n, torch.cuda.current_stream(self.device).cuda_stream)
        if n2 < 0:
            _lib.check(int(n2))
        return out

    def _free(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.qa_bicodec_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass


class BiCodecTokenizer(torch.nn.Module):
    """Decode side of the reference's BiCodecTokenizer (QuarkAudio-UniSE/model/bicodec/audio_tokenizer.py:107-120), the object
    `Model.test_step` calls: `detokenize(global_tokens [B, 1, 32], semantic_tokens [B, T]) -> wav [B, 1, T * 320]`."""

    def __init__(self, model_dir=None, device="cuda:0", *, model: BiCodec | None = None, spec: BiCodecSpec = SPEC_BICODEC, **kwargs):
        super().__init__()
        if model is None:
            if model_dir is None:
                raise ValueError("BiCodecTokenizer needs model_dir (with BiCodec/model.safetensors) or model=")
            model = BiCodec.load_from_checkpoint(f"{model_dir}/BiCodec", device=device, spec=spec)  # audio_tokenizer.py:50-53
        self.model = model
        self.device = model.device

    @torch.no_grad()
    def detokenize(self, global_tokens: torch.Tensor, semantic_tokens: torch.Tensor) -> torch.Tensor:
        return self.model.detokenize(semantic_tokens, global_tokens)

    def tokenize(self, *args, **kwargs):
        raise _lib.QuarkAudioError(-4, "BiCodec tokenize (wav2vec2-BERT features + ECAPA speaker encoder) is training-side and not part of "
                                       "the inference path: the LM produces the tokens")
