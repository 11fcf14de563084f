"""TEST INFRASTRUCTURE ONLY - CPU restatement of H-Codec 2.0 `Codec.encode` / `Codec.decode`
(48 kHz, 12.5 Hz frames, STFT-domain ConvNeXt encoder, 16+16 codebooks).  Paths relative to
/root/reference/QuarkAudio-HCodec/HCodec-2.0/.

    Codec.encode / decode          vq/codec.py:76-99
    CodecEncoder.forward           vq/codec_encoder.py:12-79   (STFT 1920/960 -> log-mag | phase/pi -> Conv1d k3 -> LN -> ConvNeXt x L
                                                                -> Transformer(+LSTM) -> LN -> Conv1d k=2s+1 stride s)
    CodecDecoder.forward           vq/codec_decoder.py:14-72   (repeat_interleave(s) -> Conv1d k=s+1 -> prior_net -> LN -> ConvNeXt -> ISTFT)
    shared blocks                  same files as H-Codec 1.0 (restated in oracle/hcodec_ref.py)

Pinned against the reference's own modules (constructed from a reduced YAML config, which is the reference's own code path) by
tests/test_oracle_cpu.py and tests/golden/hcodec20_*.npz.  RVQ: third-party, pinned to vq/core_vq.py (see hcodec_ref.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

from . import hcodec_ref as R

Tensor = torch.Tensor
SD = Dict[str, Tensor]


@dataclass(frozen=True)
class HCodec20Spec:
    """conf/large_12.5hz_config.yaml."""

    enc_dim: int = 1536
    enc_inter: int = 4608
    enc_convnext_layers: int = 24
    enc_transformer_layers: int = 2
    dimension: int = 512  # code_dim
    n_fft: int = 1920
    hop: int = 960
    stride: int = 4  # int(50 / target_frame_rate)
    sem_in: int = 768
    sem_ch: int = 1536
    sem_strides: Tuple[int, ...] = (2, 1, 2)
    codebook_size: int = 1024
    num_quantizers: int = 16
    dec_dim: int = 1536
    dec_inter: int = 4608
    dec_convnext_layers: int = 32
    dec_transformer_layers: int = 2
    gn_groups: int = 32
    tr_inter_cap: int = 4096  # min(dim * 4, 4096), codec_encoder.py:49
    causal: bool = False  # `causal` of encoder_config / decoder_config (codec_encoder.py:23, codec_decoder.py:25; the YAML ships false)

    @property
    def frame_hop(self) -> int:  # samples per code frame (audio_tokenizer.py:41)
        return self.hop * self.stride

    def tr_inter(self, d: int) -> int:
        return min(4 * d, self.tr_inter_cap)

    def as_10(self, d: int, layers: int) -> R.HCodecSpec:
        """view used to reuse the shared-block restatements"""
        return R.HCodecSpec(sem_in=self.sem_in, sem_ch=self.sem_ch, sem_strides=self.sem_strides, code_dim=self.dimension,
                            codebook_size=self.codebook_size, num_quantizers=self.num_quantizers, dec_dim=d,
                            dec_heads=d // 64, dec_layers=layers, convnext_layers=self.dec_convnext_layers,
                            n_fft=self.n_fft, hop=self.hop, gn_groups=self.gn_groups)


SPEC_20 = HCodec20Spec()


def _transformer(sd: SD, p: str, x: Tensor, layers: int, heads: int, taps=None, causal: bool = False) -> Tensor:
    return R.transformer(sd, p, x, layers, heads, taps, causal)  # MLP width is read off the weights (w1: [inter, d])


def _conv(x: Tensor, w: Tensor, b: Tensor, stride: int, causal: bool) -> Tensor:
    """vq/conv.py Conv1d (:32-56): ConstantPad1d (k - stride, 0) if causal else (k // 2, k // 2), then a plain strided Conv1d."""
    k = w.shape[-1]
    return F.conv1d(F.pad(x, (k - stride, 0) if causal else (k // 2, k // 2)), w, b, stride=stride)


def codec_encoder(sd: SD, wav: Tensor, spec: HCodec20Spec = SPEC_20, taps=None) -> Tensor:
    """codec_encoder.py:62-79.  wav [B, T] (T multiple of hop) -> [B, dimension, T / (hop*stride)]."""
    pad = (spec.n_fft - spec.hop) // 2
    x = F.pad(wav, (pad, pad))
    st = torch.stft(x, spec.n_fft, spec.hop, window=torch.hann_window(spec.n_fft), center=False, return_complex=True)
    mag, phase = st.abs(), st.angle()
    x = torch.cat([torch.log(torch.clip(mag, min=1e-5)), phase / torch.pi], dim=1)
    if taps is not None:
        taps["enc.stft"] = x
    p = "encoder"
    cz = spec.causal
    x = _conv(x, sd[p + ".embed.conv.weight"], sd[p + ".embed.conv.bias"], 1, cz)
    c = x.shape[1]
    x = F.layer_norm(x.transpose(1, 2), (c,), sd[p + ".norm.weight"], sd[p + ".norm.bias"], eps=1e-6).transpose(1, 2)
    for i in range(spec.enc_convnext_layers):
        x = R.convnext_block(sd, f"{p}.prior_net.{i}", x, cz)
    if taps is not None:
        taps["enc.prior"] = x
    x = _transformer(sd, p + ".post_net.1", x.transpose(1, 2), spec.enc_transformer_layers, c // 64, taps, cz).transpose(1, 2)
    x = F.layer_norm(x.transpose(1, 2), (c,), sd[p + ".final_layer_norm.weight"], sd[p + ".final_layer_norm.bias"], eps=1e-6).transpose(1, 2)
    return _conv(x, sd[p + ".out.conv.weight"], sd[p + ".out.conv.bias"], spec.stride, cz)


def encode(sd: SD, wav: Tensor, feat: Tensor, spec: HCodec20Spec = SPEC_20, taps=None):
    """Codec.encode, codec.py:76-87.  wav [B, T] (no channel dim, audio_tokenizer.py:73), feat [B, sem_in, N50]."""
    emb = codec_encoder(sd, wav, spec, taps)
    sem = R.semantic_encoder(sd, feat, spec.as_10(spec.dec_dim, spec.dec_transformer_layers))
    if taps is not None:
        taps["enc.emb"], taps["enc.sem"] = emb, sem
    ac, _ = R.rvq_search(emb.transpose(1, 2), R.rvq_codebooks(sd, "quantizer", spec.num_quantizers))
    sc, _ = R.rvq_search(sem.transpose(1, 2), R.rvq_codebooks(sd, "semantic_quantizer", spec.num_quantizers))
    return ac.transpose(1, 2).contiguous(), sc.transpose(1, 2).contiguous()


def codec_decoder(sd: SD, x: Tensor, spec: HCodec20Spec = SPEC_20, taps=None) -> Tensor:
    """codec_decoder.py:61-72."""
    p = "decoder"
    s10 = spec.as_10(spec.dec_dim, spec.dec_transformer_layers)
    cz = spec.causal
    x = x.repeat_interleave(spec.stride, dim=-1)
    x = _conv(x, sd[p + ".embed.conv.weight"], sd[p + ".embed.conv.bias"], 1, cz)
    if taps is not None:
        taps["dec.embed"] = x
    x = R.resnet_block(sd, p + ".prior_net.0", x, spec.gn_groups, cz)
    x = R.resnet_block(sd, p + ".prior_net.1", x, spec.gn_groups, cz)
    c = x.shape[1]
    x = _transformer(sd, p + ".prior_net.3", x.transpose(1, 2), spec.dec_transformer_layers, c // 64, taps, cz).transpose(1, 2)
    x = R.resnet_block(sd, p + ".prior_net.5", x, spec.gn_groups, cz)
    x = R.resnet_block(sd, p + ".prior_net.6", x, spec.gn_groups, cz)
    x = F.group_norm(x, spec.gn_groups, sd[p + ".prior_net.7.weight"], sd[p + ".prior_net.7.bias"], eps=1e-6)
    x = F.layer_norm(x.transpose(1, 2), (c,), sd[p + ".norm.weight"], sd[p + ".norm.bias"], eps=1e-6).transpose(1, 2)
    for i in range(spec.dec_convnext_layers):
        x = R.convnext_block(sd, f"{p}.post_net.{i}", x, cz)
    x = F.layer_norm(x.transpose(1, 2), (c,), sd[p + ".final_layer_norm.weight"], sd[p + ".final_layer_norm.bias"], eps=1e-6)
    if taps is not None:
        taps["dec.backbone"] = x
    return R.istft_head(sd, x, s10.n_fft, s10.hop, taps)


def decode(sd: SD, acoustic_codes: Tensor, semantic_codes: Tensor, spec: HCodec20Spec = SPEC_20, taps=None) -> Tensor:
    """Codec.decode, codec.py:90-99."""
    a = R.rvq_lookup(acoustic_codes.transpose(1, 2), R.rvq_codebooks(sd, "quantizer", spec.num_quantizers))
    s = R.rvq_lookup(semantic_codes.transpose(1, 2), R.rvq_codebooks(sd, "semantic_quantizer", spec.num_quantizers))
    return codec_decoder(sd, torch.cat([a.transpose(1, 2), s.transpose(1, 2)], dim=1), spec, taps)
