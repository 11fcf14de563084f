"""Python mirror of the reference's H-Codec interface, backed by libquarkaudio_hip.so.

    Codec            <-> QuarkAudio-HCodec/HCodec-1.0/vq/codec.py:21-187        (encode / decode / load_state_dict)
    HCodecTokenizer  <-> QuarkAudio-HCodec/HCodec-1.0/audio_tokenizer.py:18-66  (pad_wav / tokenize / detokenize)

Tensors stay PyTorch-ROCm CUDA tensors (allocation + streams only); all arithmetic happens in the HIP library.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Callable, Dict, Optional, Tuple

import torch

from . import _lib


@dataclass(frozen=True)
class HCodecSpec:
    """Architecture constants of H-Codec 1.0, hard-coded in the reference at vq/codec.py:30-136."""

    n_filters: int = 32
    ratios: Tuple[int, ...] = (2, 4, 5, 8)  # encoder order (codec.py:33 lists [8,5,4,2]; seanet.py:114 reverses)
    dimension: int = 512
    enc_heads: int = 8
    enc_layers: int = 2
    sem_in: int = 768
    sem_ch: int = 768
    sem_strides: Tuple[int, ...] = (2, 1)
    code_dim: int = 512
    codebook_size: int = 1024
    num_quantizers: int = 4
    dec_dim: int = 768
    dec_inter: int = 2304
    dec_heads: int = 8
    dec_layers: int = 2
    convnext_layers: int = 12
    n_fft: int = 1280
    hop: int = 320
    gn_groups: int = 32
    # H-Codec 1.5 (QuarkAudio-HCodec/HCodec-1.5/conf/config_adaptive_v3.yaml:65-111); adaptive=False is H-Codec 1.0
    adaptive: bool = False
    agg_layers: int = 32
    agg_heads: int = 8
    agg_ff: int = 2048
    bt_layers: int = 32
    bt_heads: int = 8
    bt_ff: int = 2048
    threshold: float = 0.6
    max_tokens_per_group: int = 8
    # H-Codec 2.0 (QuarkAudio-HCodec/HCodec-2.0/conf/large_12.5hz_config.yaml); version 10 = SEANet family (1.0 / 1.5)
    version: int = 10
    enc_dim: int = 1536
    enc_inter: int = 4608
    enc_convnext_layers: int = 24
    frame_stride: int = 4
    tr_inter_cap: int = 0  # transformer MLP width = min(4 d, cap); 0 = 4 d

    @property
    def enc_hop(self) -> int:
        """samples per code frame"""
        if self.version == 20:
            return self.hop * self.frame_stride
        return int(math.prod(self.ratios)) * 2

    @property
    def dec_upsample(self) -> int:
        return self.frame_stride if self.version == 20 else 2

    def to_c(self) -> "_lib.qa_hcodec_spec":
        s = _lib.qa_hcodec_spec()
        s.n_filters, s.n_ratios = self.n_filters, len(self.ratios)
        for i, r in enumerate(self.ratios):
            s.ratios[i] = r
        s.dimension, s.enc_heads, s.enc_layers = self.dimension, self.enc_heads, self.enc_layers
        s.sem_in, s.sem_ch, s.n_sem_strides = self.sem_in, self.sem_ch, len(self.sem_strides)
        for i, r in enumerate(self.sem_strides):
            s.sem_strides[i] = r
        s.code_dim, s.codebook_size, s.num_quantizers = self.code_dim, self.codebook_size, self.num_quantizers
        s.dec_dim, s.dec_inter, s.dec_heads, s.dec_layers = self.dec_dim, self.dec_inter, self.dec_heads, self.dec_layers
        s.convnext_layers, s.n_fft, s.hop, s.gn_groups = self.convnext_layers, self.n_fft, self.hop, self.gn_groups
        s.adaptive, s.agg_layers, s.agg_heads, s.agg_ff = int(self.adaptive), self.agg_layers, self.agg_heads, self.agg_ff
        s.bt_layers, s.bt_heads, s.bt_ff = self.bt_layers, self.bt_heads, self.bt_ff
        s.max_tokens_per_group, s.threshold = self.max_tokens_per_group, self.threshold
        s.version, s.enc_dim, s.enc_inter = self.version, self.enc_dim, self.enc_inter
        s.enc_convnext_layers, s.frame_stride, s.tr_inter_cap = self.enc_convnext_layers, self.frame_stride, self.tr_inter_cap
        return s


SPEC_10 = HCodecSpec()
# H-Codec 1.5: SEANet stride order 8,5,4,2 (config lists [2,4,5,8], seanet.py:114 reverses it), XLSR features, decoder 1024
SPEC_15 = HCodecSpec(ratios=(8, 5, 4, 2), sem_in=1024, sem_ch=1024, dec_dim=1024, dec_inter=2304, adaptive=True)
# H-Codec 2.0: 48 kHz, 12.5 Hz frames, STFT/ConvNeXt encoder, 16 + 16 codebooks
SPEC_20 = HCodecSpec(version=20, enc_dim=1536, enc_inter=4608, enc_convnext_layers=24, enc_layers=2, frame_stride=4,
                     tr_inter_cap=4096, dimension=512, sem_in=768, sem_ch=1536, sem_strides=(2, 1, 2), num_quantizers=16,
                     dec_dim=1536, dec_inter=4608, dec_heads=24, dec_layers=2, convnext_layers=32, n_fft=1920, hop=960)


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class Codec:
    """Drop-in for `vq.Codec` on the inference path: `encode(x, feat)` / `decode(acoustic_codes, semantic_codes)`.

    The constructor signature keeps the reference's three (ignored) kwargs dicts (codec.py:22-27); weights come
    through `load_state_dict` in the reference's own key layout (weight_g / weight_v, `layers.{q}._codebook.embed`).
    """

    def __init__(self, encoder_kwargs=None, decoder_kwargs=None, quantizer_kwargs=None, adaptive_kwargs=None,
                 semantic_decoder_kwargs=None, *, spec: HCodecSpec = SPEC_10,
                 device: str | torch.device = "cuda:0"):
        self.spec = spec
        self.device = torch.device(device)
        self._handle = C.c_void_p()
        self._lib = _lib.load_library()

    # -- weights -------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        _lib.require_device()
        if self.device.type != "cuda":
            raise _lib.QuarkAudioError(-1, f"Codec lives on a HIP device, got {self.device}")
        self._free()
        table, n, keep = _lib.tensor_table(state_dict)
        spec_c = self.spec.to_c()
        handle = C.c_void_p()
        _lib.check(self._lib.qa_hcodec_create(C.byref(handle), C.byref(spec_c), table, n, self.device.index or 0))
        del keep
        self._handle = handle
        return self

    def eval(self):
        return self

    def to(self, device):
        if torch.device(device) != self.device:
            raise _lib.QuarkAudioError(-4, "move the handle by constructing Codec(device=...) and reloading the weights")
        return self

    def _free(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.qa_hcodec_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    def _require_loaded(self):
        if not self._handle.value:
            raise _lib.QuarkAudioError(-3, "Codec has no weights: call load_state_dict first")

    # -- hot path ------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, x: torch.Tensor, feat: torch.Tensor, use_mask=False, domain_split=None, threshold: float = 0.0):
        """codec.py:166-175.  x [B,1,T] fp32, feat [B, sem_in, N50] fp32 (any strides) -> two int64 [B, nq, N25]."""
        self._require_loaded()
        if self.spec.version == 20 and x.dim() == 2:  # H-Codec 2.0 passes wav without the channel dim (audio_tokenizer.py:73)
            x = x.unsqueeze(1)
        if x.dim() != 3 or x.shape[1] != 1:
            raise _lib.QuarkAudioError(-1, f"encode expects x of shape [B,1,T], got {tuple(x.shape)}")
        if feat.dim() != 3 or feat.shape[0] != x.shape[0] or feat.shape[1] != self.spec.sem_in:
            raise _lib.QuarkAudioError(-1, f"encode expects feat of shape [B,{self.spec.sem_in},N], got {tuple(feat.shape)}")
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        feat = feat.to(device=self.device, dtype=torch.float32)
        B, _, T = x.shape
        n25 = T // self.spec.enc_hop
        q = self.spec.num_quantizers
        ac = torch.empty((B, q, n25), dtype=torch.int64, device=self.device)
        sc = torch.empty((B, q, n25), dtype=torch.int64, device=self.device)
        sb, sch, st = feat.stride()
        if self.spec.adaptive:
            # codec_adaptive.py:150-178: dict of length-injected codes [B, nq, G]; G is data dependent (host sync, as in the
            # reference: modeling_flexicodec_new.py:910)
            if threshold != 0.0:
                raise _lib.QuarkAudioError(-4, "per-call threshold is fixed at load time: set HCodecSpec.threshold")
            g = C.c_int64(0)
            _lib.check(self._lib.qa_hcodec_encode_adaptive(self._handle, x.data_ptr(), B, T, feat.data_ptr(), sb, sch, st,
                                                           feat.shape[2], ac.data_ptr(), sc.data_ptr(), C.byref(g),
                                                           _stream_ptr(self.device)))
            G = int(g.value)
            return {"acoustic_codes": ac.view(-1)[: B * q * G].view(B, q, G), "semantic_codes": sc.view(-1)[: B * q * G].view(B, q, G)}
        _lib.check(self._lib.qa_hcodec_encode(self._handle, x.data_ptr(), B, T, feat.data_ptr(), sb, sch, st,
                                              feat.shape[2], ac.data_ptr(), sc.data_ptr(), _stream_ptr(self.device)))
        return ac, sc

    @torch.no_grad()
    def decode(self, acoustic_codes: torch.Tensor, semantic_codes: torch.Tensor, token_lengths: Optional[torch.Tensor] = None):
        """codec.py:178-187.  int64 [B, nq, N25] x2 -> wav [B, N25 * 2 * hop].  H-Codec 1.5 (codec_adaptive.py:181-199):
        length-injected codes [B, nq, G] (or plain codes + token_lengths [B, G])."""
        self._require_loaded()
        if self.spec.adaptive:
            return self._decode_adaptive(acoustic_codes, semantic_codes, token_lengths)
        q = self.spec.num_quantizers
        if acoustic_codes.shape != semantic_codes.shape or acoustic_codes.dim() != 3 or acoustic_codes.shape[1] != q:
            raise _lib.QuarkAudioError(-1, f"decode expects two [B,{q},N] code tensors, got "
                                           f"{tuple(acoustic_codes.shape)} / {tuple(semantic_codes.shape)}")
        ac = acoustic_codes.to(device=self.device, dtype=torch.int64).contiguous()
        sc = semantic_codes.to(device=self.device, dtype=torch.int64).contiguous()
        for name, c in (("acoustic_codes", ac), ("semantic_codes", sc)):
            if c.numel() and (int(c.min()) < 0 or int(c.max()) >= self.spec.codebook_size):
                raise IndexError(f"{name} out of range [0, {self.spec.codebook_size})")  # reference: F.embedding raises
        B, _, N = ac.shape
        wav = torch.empty((B, N * self.spec.dec_upsample * self.spec.hop), dtype=torch.float32, device=self.device)
        _lib.check(self._lib.qa_hcodec_decode(self._handle, ac.data_ptr(), sc.data_ptr(), B, N, wav.data_ptr(),
                                              _stream_ptr(self.device)))
        return wav

    def enable_taps(self, on: bool = True):
        """Test hook: make encode/decode snapshot their named intermediates (see DESIGN.md "taps")."""
        self._require_loaded()
        _lib.check(self._lib.qa_hcodec_enable_taps(self._handle, int(on)))
        return self

    def _decode_adaptive(self, acoustic_codes, semantic_codes, token_lengths):
        q, K = self.spec.num_quantizers, self.spec.codebook_size
        if acoustic_codes.shape != semantic_codes.shape or acoustic_codes.dim() != 3 or acoustic_codes.shape[1] != q:
            raise _lib.QuarkAudioError(-1, f"decode expects two [B,{q},G] code tensors")
        ac = acoustic_codes.to(device=self.device, dtype=torch.int64).contiguous()
        sc = semantic_codes.to(device=self.device, dtype=torch.int64).contiguous()
        if token_lengths is not None:  # plain codes + explicit lengths: inject, exactly what encode emits (codec_adaptive.py:68-73)
            tl = token_lengths.to(device=self.device, dtype=torch.int64).unsqueeze(1)
            ac, sc = (tl - 1) * K + ac, (tl - 1) * K + sc
        B, _, G = ac.shape
        frames = C.c_int64(0)
        _lib.check(self._lib.qa_hcodec_adaptive_frames(self._handle, sc.data_ptr(), B, G, C.byref(frames), _stream_ptr(self.device)))
        n = int(frames.value)
        wav = torch.empty((B, n * 2 * self.spec.hop), dtype=torch.float32, device=self.device)
        _lib.check(self._lib.qa_hcodec_decode_adaptive(self._handle, ac.data_ptr(), sc.data_ptr(), B, G, n, wav.data_ptr(),
                                                       _stream_ptr(self.device)))
        return wav

    def tap(self, name: str) -> torch.Tensor:
        """Test hook: flat fp32 copy of a named intermediate of the last encode/decode (channel-last layout)."""
        n = self._lib.qa_hcodec_tap(self._handle, name.encode(), None, 0, None)
        if n < 0:
            _lib.check(int(n))
        out = torch.empty(int(n), dtype=torch.float32, device=self.device)
        n2 = self._lib.qa_hcodec_tap(self._handle, name.encode(), out.data_ptr(), n, _stream_ptr(self.device))
        if n2 < 0:
            _lib.check(int(n2))
        return out


class HCodecTokenizer:
    """Drop-in for the reference's HCodecTokenizer (audio_tokenizer.py:18-66).

    `feature_extractor`: either a `unified_audio_amd.SSLFeatureExtractor` (the HuBERT / XLSR front-end on the same HIP
    library: `tokenize(wav)` then never leaves the device path), or the reference's PyTorch module
    (`feature_extractor(wav[B,T+320], output_hidden_states=True).hidden_states`); or pass precomputed features to
    `tokenize(wav, feats=...)`.
    """

    def __init__(self, pt_path=None, *, state_dict=None, feature_extractor: Optional[Callable] = None,
                 device: str | torch.device = "cuda:0", spec: HCodecSpec = SPEC_10, **kwargs):
        if state_dict is None:
            if pt_path is None:
                raise ValueError("HCodecTokenizer needs pt_path or state_dict")
            state_dict = torch.load(pt_path, map_location="cpu")  # audio_tokenizer.py:24
        self.model = Codec(None, None, None, spec=spec, device=device).load_state_dict(state_dict)
        self.feature_extractor = feature_extractor
        self.hop_length = spec.enc_hop  # 640 = 25 Hz (audio_tokenizer.py:31)
        self.device = torch.device(device)

    @torch.no_grad()
    def extract_wav2vec2_features(self, wavs: torch.Tensor) -> torch.Tensor:
        """audio_tokenizer.py:35-48: pad (160,160), mean of all hidden states, sign*|x|^0.3 compression."""
        if self.feature_extractor is None:
            raise _lib.QuarkAudioError(-3, "no feature_extractor was given; pass feats= to tokenize()")
        from .ssl import SSLFeatureExtractor

        if isinstance(self.feature_extractor, SSLFeatureExtractor):  # padding, averaging and compression happen in the library
            return self.feature_extractor(wavs)
        wavs = torch.nn.functional.pad(wavs, (160, 160))
        feats = self.feature_extractor(wavs, output_hidden_states=True)
        feats_mix = torch.stack(feats.hidden_states, dim=1).mean(1)
        symbol = (feats_mix > 0).float() * 2 - 1
        return symbol * feats_mix.abs() ** 0.3

    def pad_wav(self, wav: torch.Tensor) -> torch.Tensor:
        pad = math.ceil(wav.size(-1) / self.hop_length) * self.hop_length - wav.size(-1)
        return torch.nn.functional.pad(wav, (0, pad))

    @torch.no_grad()
    def tokenize(self, wav: torch.Tensor, feats: Optional[torch.Tensor] = None):
        wav = self.pad_wav(wav.to(self.device))
        if feats is None:
            feats = self.extract_wav2vec2_features(wav)  # (b, t, d)
        feats = feats.to(self.device).transpose(-2, -1)  # (b, d, t) view; the library reads it through its strides
        return self.model.encode(wav.unsqueeze(1), feats)

    @torch.no_grad()
    def detokenize(self, acoustic_codes: torch.Tensor, semantic_codes: torch.Tensor, token_lengths=None):
        """1.0: audio_tokenizer.py:64-66; 1.5: HCodec-1.5/audio_tokenizer.py:83-86 (call as detokenize(**codes))."""
        if self.model.spec.adaptive:
            return self.model.decode(acoustic_codes, semantic_codes, token_lengths)
        return self.model.decode(acoustic_codes, semantic_codes)
