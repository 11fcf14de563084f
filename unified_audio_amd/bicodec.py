"""Python mirror of the reference's BiCodec detokenizer, backed by libquarkaudio_hip.so.

    BiCodec.detokenize           <->  QuarkAudio-UniSE/model/bicodec/bicodec.py:182-199
    BiCodecTokenizer.detokenize  <->  QuarkAudio-UniSE/model/bicodec/audio_tokenizer.py (called at model/model.py:193,223)

Only the decode side is on the UniSE inference path (the LM produces the tokens); `tokenize` (wav2vec2-BERT features, ECAPA speaker
encoder) belongs to training / data preparation and is not offered.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Any, Dict, Mapping, Tuple

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

    @classmethod
    def from_config(cls, audio_tokenizer: Mapping[str, Any]) -> "BiCodecSpec":
        """The `audio_tokenizer` section of `{model_dir}/config.yaml`, block by block as `BiCodec.load_from_checkpoint` hands it to its
        module constructors (bicodec.py:80-87): `quantizer` -> FactorizedVectorQuantize (factorized_vector_quantize.py:37-48),
        `speaker_encoder` -> SpeakerEncoder (speaker_encoder.py:48-56), `prenet` -> Decoder (feat_decoder.py:37-47), `decoder` ->
        WaveGenerator (wave_generator.py:60-67).  A missing required key is a TypeError and an unknown key in a block whose constructor
        takes no **kwargs is a TypeError, as in the reference; a value the detokenizer kernels have no path for is refused by name
        (QuarkAudioError -4) instead of being ignored.  `mel_params`, `encoder` and `postnet` configure the encoder / training side
        (tokenize, forward) and are not read."""
        def block(name, required, optional=(), open_kwargs=False):
            if name not in audio_tokenizer:
                raise KeyError(f"config.yaml: audio_tokenizer.{name} is missing (bicodec.py:80-87 reads it)")
            b = dict(audio_tokenizer[name])
            missing = [k for k in required if k not in b]
            if missing:
                raise TypeError(f"audio_tokenizer.{name}: missing required argument(s) {missing}")
            unknown = [k for k in b if k not in required and k not in optional]
            if unknown and not open_kwargs:
                raise TypeError(f"audio_tokenizer.{name}: unexpected keyword argument(s) {unknown}")
            return b

        def refuse(what):
            raise _lib.QuarkAudioError(-4, f"BiCodec config.yaml: {what} - the MI355X detokenizer has no path for it")

        q = block("quantizer", ("input_dim", "codebook_size", "codebook_dim", "commitment"), open_kwargs=True)  # **kwargs: training-side keys pass
        spk = block("speaker_encoder", (), ("input_dim", "out_dim", "latent_dim", "token_num", "fsq_levels", "fsq_num_quantizers"))
        pre = block("prenet", ("input_channels", "vocos_dim", "vocos_intermediate_dim", "vocos_num_layers", "out_channels"),
                    ("condition_dim", "sample_ratios", "use_tanh_at_final"))
        dec = block("decoder", ("input_channel", "channels", "rates", "kernel_sizes"), ("d_out",))
        latent = int(q["input_dim"])
        if int(q["codebook_dim"]) == latent:
            refuse("quantizer.input_dim == codebook_dim (Identity projections, factorized_vector_quantize.py:59-65)")
        spk_out = int(spk.get("out_dim", 512))
        if int(spk.get("fsq_num_quantizers", 1)) != 1:
            refuse(f"speaker_encoder.fsq_num_quantizers = {spk['fsq_num_quantizers']} (one FSQ stage is built)")
        levels = tuple(int(v) for v in spk.get("fsq_levels", (4, 4, 4, 4, 4, 4)))
        if not 1 <= len(levels) <= 8:
            refuse(f"speaker_encoder.fsq_levels with {len(levels)} entries (1 .. 8)")
        if [int(r) for r in pre.get("sample_ratios", (1, 1))] != [1, 1]:
            refuse(f"prenet.sample_ratios = {list(pre['sample_ratios'])} (the two ratio-1 SamplingBlocks of the published model are built)")
        if bool(pre.get("use_tanh_at_final", False)):
            refuse("prenet.use_tanh_at_final = true")
        if int(dec.get("d_out", 1)) != 1:
            refuse(f"decoder.d_out = {dec['d_out']} (mono)")
        rates, ks = tuple(int(v) for v in dec["rates"]), tuple(int(v) for v in dec["kernel_sizes"])
        if len(rates) != len(ks) or not 1 <= len(rates) <= 8:
            refuse(f"decoder.rates / kernel_sizes of lengths {len(rates)} / {len(ks)} (equal, 1 .. 8)")
        # the tensors that meet in detokenize (bicodec.py:193-199): z_q [latent] -> prenet -> + d_vector [spk_out] -> decoder
        widths = {"quantizer.input_dim": latent, "prenet.input_channels": int(pre["input_channels"]), "prenet.out_channels": int(pre["out_channels"]),
                  "prenet.condition_dim": int(pre["condition_dim"]) if pre.get("condition_dim") is not None else None,
                  "speaker_encoder.out_dim": spk_out, "decoder.input_channel": int(dec["input_channel"])}
        if widths["prenet.condition_dim"] is None:
            refuse("prenet.condition_dim = null (detokenize conditions the prenet on the d-vector)")
        if len(set(widths.values())) != 1:
            raise ValueError(f"BiCodec config.yaml: widths that must agree in detokenize differ: {widths}")
        mel = audio_tokenizer.get("mel_params") or {}
        return cls(latent_dim=latent, codebook_size=int(q["codebook_size"]), codebook_dim=int(q["codebook_dim"]),
                   mel_dim=int(spk.get("input_dim", mel.get("num_mels", 100))), spk_latent_dim=int(spk.get("latent_dim", 128)),
                   token_num=int(spk.get("token_num", 32)), fsq_levels=levels, vocos_dim=int(pre["vocos_dim"]),
                   vocos_inter=int(pre["vocos_intermediate_dim"]), vocos_layers=int(pre["vocos_num_layers"]),
                   gen_channels=int(dec["channels"]), rates=rates, kernel_sizes=ks)

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


def load_config(config_path) -> Dict[str, Any]:
    """utils/file.py:116-130 (`OmegaConf.load` + the optional `base_config` merge) with PyYAML: omegaconf is not a dependency here.  A
    value that uses OmegaConf interpolation (`${...}`) cannot be resolved by a plain YAML reader and is refused rather than passed on."""
    import yaml

    def read(path):
        with open(path, "r") as f:
            cfg = yaml.safe_load(f) or {}
        if not isinstance(cfg, dict):
            raise ValueError(f"{path}: a mapping is expected at the top level")
        return cfg

    def merge(base, over):  # OmegaConf.merge: mappings merge key by key, everything else is replaced
        out = dict(base)
        for k, v in over.items():
            out[k] = merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
        return out

    def check(node, where):
        if isinstance(node, dict):
            for k, v in node.items():
                check(v, f"{where}.{k}")
        elif isinstance(node, (list, tuple)):
            for i, v in enumerate(node):
                check(v, f"{where}[{i}]")
        elif isinstance(node, str) and "${" in node:
            raise _lib.QuarkAudioError(-4, f"{config_path}: {where} = {node!r} uses OmegaConf interpolation, which this loader does not resolve")

    cfg = read(config_path)
    if cfg.get("base_config") is not None:
        cfg = merge(read(cfg["base_config"]), cfg)
    check(cfg, "config")
    return cfg


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
    def load_from_checkpoint(cls, model_dir, device="cuda:0", spec: BiCodecSpec | None = None, **kwargs) -> "BiCodec":
        """bicodec.py:69-115: the architecture from `{model_dir}/config.yaml['audio_tokenizer']` (BiCodecSpec.from_config), the weights
        from `{model_dir}/model.safetensors`; missing detokenizer tensors are reported by name by qa_bicodec_create (the reference prints
        them and goes on with random weights, bicodec.py:103-108).  `spec=` overrides the file (a directory without config.yaml)."""
        from safetensors.torch import load_file

        if spec is None:
            spec = BiCodecSpec.from_config(load_config(f"{model_dir}/config.yaml")["audio_tokenizer"])
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

    def __init__(self, model_dir=None, device="cuda:0", *, model: BiCodec | None = None, spec: BiCodecSpec | None = None, **kwargs):
        super().__init__()
        self.model_dir = model_dir
        self.config = None
        if model is None:
            if model_dir is None:
                raise ValueError("BiCodecTokenizer needs model_dir (with BiCodec/config.yaml and BiCodec/model.safetensors) or model=")
            import os

            if os.path.isfile(f"{model_dir}/config.yaml"):  # audio_tokenizer.py:40 (sample_rate, ref_segment_duration, latent_hop_length: tokenize side)
                self.config = load_config(f"{model_dir}/config.yaml")
            model = BiCodec.load_from_checkpoint(f"{model_dir}/BiCodec", device=device, spec=spec)  # audio_tokenizer.py:45
        self.model = model
        self.device = model.device

    @torch.no_grad()
    def detokenize(self, global_tokens: torch.Tensor, semantic_tokens: torch.Tensor) -> torch.Tensor:
        return self.model.detokenize(semantic_tokens, global_tokens)

    def tokenize(self, *args, **kwargs):
        raise _lib.QuarkAudioError(-4, "BiCodec tokenize (wav2vec2-BERT features + ECAPA speaker encoder) is training-side and not part of "
                                       "the inference path: the LM produces the tokens")
