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
    # the causal variant every block of the SEANet family parameterises (encoder_modules/conv.py:203-206, vq/conv.py:44-47,76-79,
    # encoder_modules/transformer.py:470-475); vq/codec.py:31 ships False
    causal: bool = False
    # H-Codec 1.5: `causal` / `context_frames` of the aggregators, `causal` / `context` of the bottleneck transformer
    # (config_adaptive_v3.yaml:84-105; shipped causal: false, where mimi/transformer.py:403-413 ignores the context)
    agg_causal: bool = False
    agg_context: int = 16
    bt_causal: bool = False
    bt_context: int = 16

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
        s.causal = int(self.causal)
        s.agg_causal, s.agg_context, s.bt_causal, s.bt_context = int(self.agg_causal), self.agg_context, int(self.bt_causal), self.bt_context
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


def _spec_from_config(config: dict, device=None) -> HCodecSpec:
    """HCodecSpec from the reference's YAML dictionaries: HCodec-1.5/conf/config_adaptive_v3.yaml (keys encoder_config /
    decoder_config / quantizer_config / adaptive_config) or HCodec-2.0/conf/large_12.5hz_config.yaml (encoder_config with
    n_fft / target_frame_rate, semantic_encoder_config)."""
    enc, dec, qz = config["encoder_config"], config["decoder_config"], config["quantizer_config"]
    if "adaptive_config" in config:  # H-Codec 1.5
        e, se, d, q, ad = enc["encoder"], enc["semantic_encoder"], dec["decoder"], qz["quantizer"], config["adaptive_config"]
        agg, tk = ad["aggregators"]["semantic_aggregator"], ad["transformer_kwargs"]
        thr = ad.get("manual_threshold")
        return HCodecSpec(n_filters=e["n_filters"], ratios=tuple(reversed(e["ratios"])), dimension=e["dimension"],
                          sem_in=se["input_channels"], sem_ch=se["encode_channels"], sem_strides=tuple(se["strides"]),
                          code_dim=q["dim"], codebook_size=q["codebook_size"], num_quantizers=q["num_quantizers"],
                          dec_dim=d["dim"], dec_inter=d["intermediate_dim"], adaptive=True, agg_layers=agg["num_layers"],
                          agg_heads=agg["num_heads"], agg_ff=agg["dim_feedforward"], bt_layers=tk["num_layers"],
                          bt_heads=tk["num_heads"], bt_ff=tk["dim_feedforward"],
                          threshold=float(thr if thr is not None else ad["similarity_threshold"]),
                          max_tokens_per_group=ad["max_tokens_per_group"],
                          agg_causal=bool(agg.get("causal", False)), agg_context=int(agg.get("context_frames") or 0),
                          bt_causal=bool(tk.get("causal", False)), bt_context=int(tk.get("context") or 0))
    se = config["semantic_encoder_config"]  # H-Codec 2.0
    stride = int(50 / enc["target_frame_rate"])
    return HCodecSpec(version=20, enc_dim=enc["dim"], enc_inter=enc["intermediate_dim"], enc_convnext_layers=enc["convnext_layers"],
                      enc_layers=enc["transformer_layers"], frame_stride=stride, tr_inter_cap=4096, dimension=enc["dimension"],
                      sem_in=se["input_channels"], sem_ch=se["encode_channels"], sem_strides=tuple(se["strides"]),
                      code_dim=qz["dim"], codebook_size=qz["codebook_size"], num_quantizers=qz["num_quantizers"],
                      dec_dim=dec["dim"], dec_inter=dec["intermediate_dim"], dec_heads=dec["dim"] // 64,
                      dec_layers=dec["transformer_layers"], convnext_layers=dec["convnext_layers"], n_fft=enc["n_fft"],
                      hop=enc["hop_length"], causal=bool(enc.get("causal", False)))


class Codec(torch.nn.Module):
    """Drop-in for `vq.Codec` on the inference path: `encode(x, feat)` / `decode(acoustic_codes, semantic_codes)`.

    The constructor keeps the reference's positional kwargs dicts (1.0: codec.py:22-27, ignored - the architecture is hard-coded
    there; 1.5: encoder / decoder / quantizer / adaptive configs; 2.0: + semantic encoder / decoder configs): when the YAML
    dictionaries are given the architecture is read from them, otherwise from `spec`.  Weights come through `load_state_dict`
    in the reference's own key layout (weight_g / weight_v, `layers.{q}._codebook.embed`).  It is an `nn.Module` without
    parameters of its own (the folded weights live in the library's device blob): `.eval()`, `.requires_grad_()`,
    `.to(device)` / `.cuda()` work as for the reference module - moving to another GPU re-creates the handle there from the
    state_dict it was loaded with - and `.train()` / `.half()` raise: this is the fp32 inference path.
    """

    def __init__(self, encoder_kwargs=None, decoder_kwargs=None, quantizer_kwargs=None, adaptive_kwargs=None,
                 semantic_decoder_kwargs=None, *, spec: Optional[HCodecSpec] = None,
                 device: str | torch.device = "cuda:0", check_codes: bool = True):
        super().__init__()
        if spec is None:
            if isinstance(encoder_kwargs, dict) and isinstance(adaptive_kwargs, dict) and "aggregators" in adaptive_kwargs:
                spec = _spec_from_config({"encoder_config": encoder_kwargs, "decoder_config": decoder_kwargs,
                                          "quantizer_config": quantizer_kwargs, "adaptive_config": adaptive_kwargs})
            elif isinstance(encoder_kwargs, dict) and "n_fft" in encoder_kwargs:  # 2.0: Codec(enc, dec, quant, sem_enc, sem_dec)
                spec = _spec_from_config({"encoder_config": encoder_kwargs, "decoder_config": decoder_kwargs,
                                          "quantizer_config": quantizer_kwargs, "semantic_encoder_config": adaptive_kwargs})
            else:
                spec = SPEC_10
        self.spec = spec
        self.device = torch.device(device)
        self.check_codes = check_codes  # decode(): refuse out-of-range codes like F.embedding (one tiny kernel + one sync)
        self._handle = C.c_void_p()
        self._lib = _lib.load_library()
        self._state = None

    # -- weights -------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True, assign: bool = False):
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
        self._state = state_dict  # a reference, not a copy: lets .to(other_gpu) rebuild the handle there
        return self

    def state_dict(self, *args, destination=None, prefix: str = "", keep_vars: bool = False):
        """The state_dict this handle was loaded with (reference key layout)."""
        out = destination if destination is not None else {}
        for k, v in (self._state or {}).items():
            out[prefix + k] = v
        return out

    def train(self, mode: bool = True):
        if mode:
            raise _lib.QuarkAudioError(-4, "unified_audio_amd.Codec is the inference path (the reference calls .eval(), audio_tokenizer.py:26)")
        return super().train(False)  # nn.Module bookkeeping: self.training = False, recursion into (no) children

    def to(self, *args, **kwargs):
        device = kwargs.get("device")
        dtype = kwargs.get("dtype")
        for a in args:
            if isinstance(a, (str, torch.device, int)):
                device = a
            elif isinstance(a, torch.dtype):
                dtype = a
        if dtype is not None and dtype != torch.float32:
            raise _lib.QuarkAudioError(-4, f"the path computes in fp32 like the reference; {dtype} is not supported")
        if device is None:
            return self
        device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        if device.type != "cuda":
            raise _lib.QuarkAudioError(-1, f"there is no CPU path: Codec cannot move to {device}")
        if device != self.device:
            self.device = device
            if self._state is not None:
                self.load_state_dict(self._state)
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", device if device is not None else (self.device.index or 0)))

    def cpu(self):
        raise _lib.QuarkAudioError(-1, "there is no CPU path")

    def half(self):
        return self.to(torch.float16)

    def float(self):
        return self

    def forward(self, *args, **kwargs):
        raise _lib.QuarkAudioError(-4, "Codec.forward is the training forward of the reference; use encode() / decode()")

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

    def _check_range_begin(self, tensors, limit: int, lo: int = 0):
        """F.embedding's range check (reference: IndexError on the host, device-side assert on a GPU), first half: one tiny counting kernel
        per code tensor into a device counter - NO host synchronisation here.  `_check_range_end` reads the counter behind a
        synchronisation the decode performs anyway (the in-launch LSTM's ticket, or H-Codec 1.5's frame-count read-back), so a decode costs
        one host round trip less than with the synchronous check of round 3.  `check_codes=False` skips it (the kernels clamp indices for
        memory safety)."""
        if not self.check_codes:
            return None
        bad = torch.zeros(1, dtype=torch.int64, device=self.device)
        for t in tensors:
            _lib.check(self._lib.qa_codes_check_async(t.data_ptr(), t.numel(), lo, limit, bad.data_ptr(), _stream_ptr(self.device)))
        return bad, lo, limit

    @staticmethod
    def _check_range_end(pending):
        if pending is None:
            return
        bad, lo, limit = pending
        n = int(bad.item())
        if n:
            raise IndexError(f"{n} code indices out of range [{lo}, {limit})")

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
            assert 0 <= threshold <= 1.0  # codec_adaptive.py:151; 0 = the model's manual_threshold (:158)
            g = C.c_int64(0)
            _lib.check(self._lib.qa_hcodec_encode_adaptive(self._handle, x.data_ptr(), B, T, feat.data_ptr(), sb, sch, st,
                                                           feat.shape[2], ac.data_ptr(), sc.data_ptr(), C.byref(g),
                                                           float(threshold), _stream_ptr(self.device)))
            G = int(g.value)
            return {"acoustic_codes": ac.view(-1)[: B * q * G].view(B, q, G), "semantic_codes": sc.view(-1)[: B * q * G].view(B, q, G)}
        _lib.check(self._lib.qa_hcodec_encode(self._handle, x.data_ptr(), B, T, feat.data_ptr(), sb, sch, st,
                                              feat.shape[2], ac.data_ptr(), sc.data_ptr(), _stream_ptr(self.device)))
        return ac, sc

    @torch.no_grad()
    def decode(self, acoustic_codes: torch.Tensor, semantic_codes: torch.Tensor, token_lengths: Optional[torch.Tensor] = None):
        """codec.py:178-187.  int64 [B, nq, N25] x2 -> wav [B, N25 * 2 * hop].  H-Codec 1.5 (codec_adaptive.py:181-199):
        length-injected codes [B, nq, G] (or plain codes + token_lengths [B, G]).
        Codes: -1 = dropped (zero vector, as upstream's ResidualVQ masks it; 1.0 / 2.0 only - in the 1.5 wire format negative values carry
        the group lengths).  Any other value outside [0, codebook_size) raises IndexError like F.embedding - AFTER the decode has run on
        clamped indices (r04: the counter is read behind the decode's own host synchronisation, so the failing path costs a full decode;
        `check_codes=False` skips the check)."""
        self._require_loaded()
        if self.spec.adaptive:
            return self._decode_adaptive(acoustic_codes, semantic_codes, token_lengths)
        q = self.spec.num_quantizers
        if acoustic_codes.shape != semantic_codes.shape or acoustic_codes.dim() != 3 or acoustic_codes.shape[1] != q:
            raise _lib.QuarkAudioError(-1, f"decode expects two [B,{q},N] code tensors, got "
                                           f"{tuple(acoustic_codes.shape)} / {tuple(semantic_codes.shape)}")
        ac = acoustic_codes.to(device=self.device, dtype=torch.int64).contiguous()
        sc = semantic_codes.to(device=self.device, dtype=torch.int64).contiguous()
        # -1 is legal: a dropped code, masked to a zero vector by the third-party get_output_from_indices (INTEGRATION.md "Edge semantics")
        pending = self._check_range_begin((ac, sc), self.spec.codebook_size, lo=-1)
        B, _, N = ac.shape
        wav = torch.empty((B, N * self.spec.dec_upsample * self.spec.hop), dtype=torch.float32, device=self.device)
        _lib.check(self._lib.qa_hcodec_decode(self._handle, ac.data_ptr(), sc.data_ptr(), B, N, wav.data_ptr(),
                                              _stream_ptr(self.device)))
        self._check_range_end(pending)  # IndexError like the reference's F.embedding, raised behind the decode's own synchronisation
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
        # length-injected: code + (len - 1) * K with len in 0..max_tokens (len 0 = the padding groups of shorter clips: [-K, 0))
        pending = self._check_range_begin((ac, sc), K * self.spec.max_tokens_per_group, lo=-K)
        frames = C.c_int64(0)
        _lib.check(self._lib.qa_hcodec_adaptive_frames(self._handle, sc.data_ptr(), B, G, C.byref(frames), _stream_ptr(self.device)))
        self._check_range_end(pending)  # the frame count's read-back just synchronised the stream: the counter is there, before any decode work
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


class HCodecTokenizer(torch.nn.Module):
    """Drop-in for the reference's three HCodecTokenizer classes, with their constructor signatures:

        HCodecTokenizer(pt_path)                                   H-Codec 1.0   HCodec-1.0/audio_tokenizer.py:18-33
        HCodecTokenizer(config=dict)   (config['ckpt_path'], ...)  H-Codec 1.5   HCodec-1.5/audio_tokenizer.py:38-51
        HCodecTokenizer(pt_path, config_path, device)              H-Codec 2.0   HCodec-2.0/audio_tokenizer.py:19-46

    plus keyword-only extras: `state_dict=` (instead of a checkpoint path), `spec=` (instead of a YAML config), `model=` (an already
    loaded `Codec`), `feature_extractor=`.  The reference downloads its SSL model (`AutoModel.from_pretrained`: hubert_base / XLSR-53); there is
    no network here, so the front-end is passed in: a `unified_audio_amd.SSLFeatureExtractor` (HuBERT / XLSR on the same HIP
    library - `tokenize(wav)` then never leaves the device path), or the reference's own PyTorch module
    (`feature_extractor(wav, output_hidden_states=True).hidden_states`), or precomputed features via `tokenize(wav, feats=...)`.
    Per version the tokenizer applies what the reference applies around it: mean of all hidden states (1.0, 2.0) or of states
    11 / 14 / 16 (1.5), |x|^0.3 compression, and for 2.0 the 48 kHz -> 16 kHz `Resample` in front (qa_resample).
    """

    def __init__(self, pt_path=None, config_path=None, device: str | torch.device = "cuda:0", *, config: Optional[dict] = None,
                 state_dict=None, feature_extractor: Optional[Callable] = None, spec: Optional[HCodecSpec] = None, model=None, **kwargs):
        super().__init__()
        self.config = config
        sampling_rate = 16000
        if config is None and config_path is not None:  # 2.0: YAML path
            import yaml

            with open(config_path, "r") as f:
                config = yaml.safe_load(f)
        if spec is None and config is not None:
            spec = _spec_from_config(config)
        if config is not None:
            sampling_rate = int(config.get("sampling_rate", 16000))
            if pt_path is None:
                pt_path = config.get("ckpt_path")  # 1.5: load_sub_weights(config['ckpt_path'], prefix=None)
        if spec is None and model is not None:
            spec = getattr(model, "spec", None)
        spec = spec or SPEC_10
        if spec.version == 20 and sampling_rate == 16000:
            sampling_rate = 48000  # large_12.5hz_config.yaml:1
        if model is not None:  # an already loaded Codec (or any object with its spec / encode / decode): nothing to read from disk
            self.device = torch.device(getattr(model, "device", device))
            self.model = model
        else:
            if state_dict is None:
                if pt_path is None:
                    raise ValueError("HCodecTokenizer needs pt_path, config['ckpt_path'] or state_dict")
                state_dict = torch.load(pt_path, map_location="cpu")  # audio_tokenizer.py:24
                if isinstance(state_dict, dict) and "state_dict" in state_dict:  # HCodec-1.5/audio_tokenizer.py:20-25
                    state_dict = state_dict["state_dict"]
            self.device = torch.device(device) if str(device) != "cpu" else torch.device("cuda:0")  # 2.0's default device='cpu': no CPU path
            self.model = Codec(None, None, None, spec=spec, device=self.device).load_state_dict(state_dict)
        self.feature_extractor = feature_extractor
        if isinstance(feature_extractor, torch.nn.Module):
            feature_extractor.eval()  # audio_tokenizer.py:28-29: the reference puts its SSL model in eval mode at construction
        self.sampling_rate = sampling_rate
        self.hop_length = spec.enc_hop  # 640 = 25 Hz (audio_tokenizer.py:31); 3840 = 12.5 Hz at 48 kHz (2.0 :46)
        # hidden states averaged by extract_wav2vec2_features when the extractor is the reference's own PyTorch module
        self.select_layers = (11, 14, 16) if spec.adaptive else None  # HCodec-1.5/audio_tokenizer.py:58-61

    def to(self, *args, **kwargs):
        self.model.to(*args, **kwargs)
        self.device = self.model.device
        fx = self.feature_extractor
        if fx is not None and hasattr(fx, "to"):
            self.feature_extractor = fx.to(self.device) or fx
        return self

    def train(self, mode: bool = True):
        """eval() must reach a PyTorch feature_extractor registered as a submodule (dropout / layerdrop of a HuBERT built in train
        mode); train(True) is refused like Codec's."""
        if mode:
            raise _lib.QuarkAudioError(-4, "unified_audio_amd.HCodecTokenizer is the inference path (the reference calls .eval())")
        return super().train(False)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """The tokenizer's weights are the codec's (audio_tokenizer.py:24-25 loads them into self.model): forward to it."""
        self.model.load_state_dict(state_dict)
        return self

    def resample(self, wavs: torch.Tensor) -> torch.Tensor:
        """torchaudio.transforms.Resample(sampling_rate, 16000) (HCodec-2.0/audio_tokenizer.py:44,51) on the device."""
        if self.sampling_rate == 16000:
            return wavs
        x = wavs.to(device=self.device, dtype=torch.float32).contiguous()
        B, T = x.shape
        n = int(self.model._lib.qa_resample_length(T, self.sampling_rate, 16000))
        out = torch.empty((B, n), dtype=torch.float32, device=self.device)
        _lib.check(self.model._lib.qa_resample(x.data_ptr(), B, T, self.sampling_rate, 16000, out.data_ptr(), _stream_ptr(self.device)))
        return out

    @torch.no_grad()
    def extract_wav2vec2_features(self, wavs: torch.Tensor) -> torch.Tensor:
        """1.0: audio_tokenizer.py:35-48 (pad (160,160), mean of all hidden states, sign*|x|^0.3); 1.5: hidden states 11 / 14 / 16
        (HCodec-1.5/audio_tokenizer.py:53-67); 2.0 (`extract_ssl_features`): Resample first (HCodec-2.0/audio_tokenizer.py:48-64)."""
        if self.feature_extractor is None:
            raise _lib.QuarkAudioError(-3, "no feature_extractor was given; pass feats= to tokenize()")
        from .ssl import SSLFeatureExtractor

        wavs = self.resample(wavs)
        if isinstance(self.feature_extractor, SSLFeatureExtractor):  # padding, averaging and compression happen in the library
            fx = self.feature_extractor
            want = tuple(self.select_layers or ())
            have = tuple(fx.spec.select or ())
            if want != have:
                raise _lib.QuarkAudioError(-1, f"this tokenizer averages hidden states {want or 'all'} but the SSLFeatureExtractor was built "
                                               f"with select={have or 'all'} (H-Codec 1.5 needs SPEC_XLSR53, 1.0 / 2.0 SPEC_HUBERT_BASE)")
            return fx(wavs)
        wavs = torch.nn.functional.pad(wavs, (160, 160))
        feats = self.feature_extractor(wavs, output_hidden_states=True)
        hs = feats.hidden_states
        if self.select_layers is not None:
            feats_mix = sum(hs[i] for i in self.select_layers) / len(self.select_layers)
        else:
            feats_mix = torch.stack(hs, dim=1).mean(1)
        symbol = (feats_mix > 0).float() * 2 - 1
        return symbol * feats_mix.abs() ** 0.3

    extract_ssl_features = extract_wav2vec2_features  # the 2.0 tokenizer's name for it

    def pad_wav(self, wav: torch.Tensor) -> torch.Tensor:
        pad = math.ceil(wav.size(-1) / self.hop_length) * self.hop_length - wav.size(-1)
        return torch.nn.functional.pad(wav, (0, pad))

    @torch.no_grad()
    def tokenize(self, wav: torch.Tensor, feats: Optional[torch.Tensor] = None, threshold: float = 0.0):
        wav = self.pad_wav(wav.to(self.device))
        if feats is None:
            feats = self.extract_wav2vec2_features(wav)  # (b, t, d)
        feats = feats.to(self.device).transpose(-2, -1)  # (b, d, t) view; the library reads it through its strides
        if self.model.spec.adaptive:
            return self.model.encode(wav.unsqueeze(1), feats, threshold=threshold)
        return self.model.encode(wav.unsqueeze(1), feats)

    @torch.no_grad()
    def detokenize(self, acoustic_codes: torch.Tensor, semantic_codes: torch.Tensor, token_lengths=None):
        """1.0: audio_tokenizer.py:64-66; 1.5: HCodec-1.5/audio_tokenizer.py:83-86 (call as detokenize(**codes))."""
        if self.model.spec.adaptive:
            return self.model.decode(acoustic_codes, semantic_codes, token_lengths)
        return self.model.decode(acoustic_codes, semantic_codes)
