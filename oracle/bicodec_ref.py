"""TEST INFRASTRUCTURE ONLY - CPU restatement (plain PyTorch fp32) of `BiCodec.detokenize`, the stage the reference's UniSE
test path ends with (QuarkAudio-UniSE/model/bicodec/bicodec.py:182-199, called at model/model.py:193,223).

    FactorizedVectorQuantize.detokenize   modules/vq/factorized_vector_quantize.py:154-172   codebook look-up -> out_project (WN k1)
    SpeakerEncoder.detokenize             modules/speaker/speaker_encoder.py:111-116          ResidualFSQ look-up -> project
    ResidualFSQ.get_output_from_indices   modules/fsq/residual_fsq.py:112-156                 implicit codebook (finite_scalar_quantization.py:108-112,165-183)
    Decoder.forward (prenet)              modules/encoder_decoder/feat_decoder.py:29-97       linear_pre -> 2 x (SamplingBlock + Vocos) -> AdaLN-Vocos -> linear
    SamplingBlock.forward (ratio 1)       modules/blocks/samper.py:74-97                      3 * x  (conv_res + skip1 + skip2 of the identity paths)
    VocosBackbone / ConvNeXtBlock / AdaLayerNorm   modules/blocks/vocos.py:38-110,113-136,287-335
    WaveGenerator / DecoderBlock          modules/encoder_decoder/wave_generator.py:32-91     Snake -> weight-normed ConvTranspose1d -> 3 dilated ResidualUnits
    Snake1d / ResidualUnit                modules/blocks/layers.py:31-67

PINNED: tests/test_bicodec_oracle_cpu.py runs the reference's own modules (oracle/ref_bicodec_shim.py, /root/reference only)
against this restatement on seeded weights - same state_dict keys and shapes, waveform equal to fp32 round-off - and
tests/golden/bicodec_*.npz hold waveforms those modules produced (oracle/gen_golden_bicodec.py).  Spark-TTS's `BiCodec/config.yaml`
is not in the reference tree, so the shapes come from BiCodecSpec (the published configuration by default).
Nothing in the product path may import this module.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


@dataclass(frozen=True)
class BiCodecSpec:
    """Spark-TTS BiCodec `audio_tokenizer` configuration (config.yaml of the checkpoint the reference downloads to
    `codec_ckpt_dir`, QuarkAudio-UniSE/conf/config.yaml:120): 16 kHz, 50 Hz semantic tokens (8192 codes), 32 global tokens (FSQ 4^6)."""

    latent_dim: int = 1024        # quantizer.input_dim = prenet in / out = speaker_encoder.out_dim = decoder.input_channel
    codebook_size: int = 8192
    codebook_dim: int = 8
    mel_dim: int = 128            # speaker_encoder.input_dim (encoder side only)
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


SPEC_BICODEC = BiCodecSpec()


def _wn(sd: SD, p: str) -> Tensor:
    """torch.nn.utils.weight_norm (dim 0): w = g * v / ||v|| over all dims but the first (for ConvTranspose1d the first dim is C_in)."""
    if p + ".weight" in sd:
        return sd[p + ".weight"]
    v, g = sd[p + ".weight_v"], sd[p + ".weight_g"]
    return g * v / v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))


def fsq_codes(indices: Tensor, levels) -> Tensor:
    """FSQ.indices_to_codes without projection (finite_scalar_quantization.py:165-183): digit d = (idx // prod(levels[:d])) % L_d,
    code = (digit - L_d // 2) / (L_d // 2)."""
    lv = torch.tensor(levels, dtype=torch.int64)
    basis = torch.cumprod(torch.tensor([1] + list(levels[:-1]), dtype=torch.int64), dim=0)
    digits = (indices.unsqueeze(-1) // basis) % lv
    half = lv // 2
    return (digits - half).float() / half.float()


def semantic_detokenize(sd: SD, tokens: Tensor) -> Tensor:
    """FactorizedVectorQuantize.detokenize: tokens [B, T] -> z_q [B, latent, T]."""
    z = F.embedding(tokens, sd["quantizer.codebook.weight"]).transpose(1, 2)
    return F.conv1d(z, _wn(sd, "quantizer.out_project"), sd["quantizer.out_project.bias"])


def global_detokenize(sd: SD, tokens: Tensor, spec: BiCodecSpec) -> Tensor:
    """SpeakerEncoder.detokenize: tokens [B, 1, token_num] -> d_vector [B, latent]."""
    idx = tokens.transpose(1, 2)                                    # [B, token_num, q = 1]
    codes = fsq_codes(idx[..., 0], spec.fsq_levels)                 # one quantizer: scale (levels - 1)^0 = 1, sum over q is the code itself
    zq = F.linear(codes, sd["speaker_encoder.quantizer.project_out.weight"], sd["speaker_encoder.quantizer.project_out.bias"])  # [B, n, latent]
    x = zq.transpose(1, 2).reshape(zq.shape[0], -1)                 # (B, latent_dim, token_num) flattened channel-major
    return F.linear(x, sd["speaker_encoder.project.weight"], sd["speaker_encoder.project.bias"])


def _ada_or_ln(sd: SD, p: str, x: Tensor, cond) -> Tensor:
    """nn.LayerNorm(eps 1e-6) or AdaLayerNorm (vocos.py:113-136): LN without affine, then * scale(c) + shift(c)."""
    c = x.shape[-1]
    if p + ".scale.weight" in sd:
        scale = F.linear(cond, sd[p + ".scale.weight"], sd[p + ".scale.bias"])
        shift = F.linear(cond, sd[p + ".shift.weight"], sd[p + ".shift.bias"])
        return F.layer_norm(x, (c,), eps=1e-6) * scale.unsqueeze(1) + shift.unsqueeze(1)
    return F.layer_norm(x, (c,), sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)


def vocos_backbone(sd: SD, p: str, x: Tensor, n_layers: int, cond=None) -> Tensor:
    """VocosBackbone.forward (vocos.py:323-335): x [B, C, T] -> [B, T, C]."""
    x = F.conv1d(x, sd[p + ".embed.weight"], sd[p + ".embed.bias"], padding=3)
    x = _ada_or_ln(sd, p + ".norm", x.transpose(1, 2), cond).transpose(1, 2)
    for i in range(n_layers):
        q = f"{p}.convnext.{i}"
        y = F.conv1d(x, sd[q + ".dwconv.weight"], sd[q + ".dwconv.bias"], padding=3, groups=x.shape[1])
        y = _ada_or_ln(sd, q + ".norm", y.transpose(1, 2), cond)
        y = F.linear(F.gelu(F.linear(y, sd[q + ".pwconv1.weight"], sd[q + ".pwconv1.bias"])), sd[q + ".pwconv2.weight"], sd[q + ".pwconv2.bias"])
        x = x + (sd[q + ".gamma"] * y).transpose(1, 2)
    return F.layer_norm(x.transpose(1, 2), (x.shape[1],), sd[p + ".final_layer_norm.weight"], sd[p + ".final_layer_norm.bias"], eps=1e-6)


def prenet(sd: SD, z_q: Tensor, d_vector: Tensor, spec: BiCodecSpec, taps=None) -> Tensor:
    """Decoder.forward with sample_ratios [1, 1] (feat_decoder.py:79-96): [B, latent, T] -> [B, latent, T]."""
    x = F.linear(z_q.transpose(1, 2), sd["prenet.linear_pre.weight"], sd["prenet.linear_pre.bias"])   # [B, T, C]
    for i in range(2):
        x = 3.0 * x.transpose(1, 2)  # SamplingBlock with both scales 1: conv_res + skip1_res + skip2_res = 3 x (samper.py:78-95)
        x = vocos_backbone(sd, f"prenet.downsample.{i}.1", x, 2)                                         # [B, T, C]
    if taps is not None:
        taps["prenet.down"] = x
    x = vocos_backbone(sd, "prenet.vocos_backbone", x.transpose(1, 2), spec.vocos_layers, d_vector)
    if taps is not None:
        taps["prenet.backbone"] = x
    return F.linear(x, sd["prenet.linear.weight"], sd["prenet.linear.bias"]).transpose(1, 2)


def snake(x: Tensor, alpha: Tensor) -> Tensor:
    """layers.py:31-36: x + sin^2(alpha x) / (alpha + 1e-9), alpha [1, C, 1]."""
    return x + (alpha + 1e-9).reciprocal() * torch.sin(alpha * x).pow(2)


def residual_unit(sd: SD, p: str, x: Tensor, dilation: int) -> Tensor:
    y = F.conv1d(snake(x, sd[p + ".block.0.alpha"]), _wn(sd, p + ".block.1"), sd[p + ".block.1.bias"], dilation=dilation, padding=3 * dilation)
    y = F.conv1d(snake(y, sd[p + ".block.2.alpha"]), _wn(sd, p + ".block.3"), sd[p + ".block.3.bias"])
    return x + y  # equal lengths: the crop of layers.py:64-66 never triggers with "same" padding


def wave_generator(sd: SD, x: Tensor, spec: BiCodecSpec, taps=None) -> Tensor:
    """WaveGenerator.forward (wave_generator.py:61-91): [B, latent, T] -> [B, 1, T * prod(rates)]."""
    x = F.conv1d(x, _wn(sd, "decoder.model.0"), sd["decoder.model.0.bias"], padding=3)
    for i, (k, s) in enumerate(zip(spec.kernel_sizes, spec.rates)):
        p = f"decoder.model.{i + 1}"
        x = snake(x, sd[p + ".block.0.alpha"])
        x = F.conv_transpose1d(x, _wn(sd, p + ".block.1"), sd[p + ".block.1.bias"], stride=s, padding=(k - s) // 2)
        for j, dil in enumerate((1, 3, 9)):
            x = residual_unit(sd, f"{p}.block.{j + 2}", x, dil)
        if taps is not None:
            taps[f"gen.block{i}"] = x
    n = len(spec.rates)
    x = snake(x, sd[f"decoder.model.{n + 1}.alpha"])
    x = F.conv1d(x, _wn(sd, f"decoder.model.{n + 2}"), sd[f"decoder.model.{n + 2}.bias"], padding=3)
    return torch.tanh(x)


@torch.no_grad()
def detokenize(sd: SD, semantic_tokens: Tensor, global_tokens: Tensor, spec: BiCodecSpec = SPEC_BICODEC, taps=None) -> Tensor:
    """BiCodec.detokenize (bicodec.py:193-199): semantic_tokens [B, T] int64, global_tokens [B, 1, token_num] int64 -> wav [B, 1, T * hop]."""
    z_q = semantic_detokenize(sd, semantic_tokens)
    d = global_detokenize(sd, global_tokens, spec)
    x = prenet(sd, z_q, d, spec, taps)
    x = x + d.unsqueeze(-1)
    if taps is not None:
        taps["z_q"], taps["d_vector"], taps["prenet.out"] = z_q, d, x
    return wave_generator(sd, x, spec, taps)
