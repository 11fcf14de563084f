"""TEST INFRASTRUCTURE - CPU oracle of the SSL feature extraction in front of H-Codec.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

What it restates
    HCodecTokenizer.extract_wav2vec2_features
        QuarkAudio-HCodec/HCodec-1.0/audio_tokenizer.py:35-48  (bosonai/hubert_base: mean over ALL hidden states)
        QuarkAudio-HCodec/HCodec-1.5/audio_tokenizer.py:53-67  (facebook/wav2vec2-large-xlsr-53: hidden states 11, 14, 16)
    Model.extract_semantic_features
        QuarkAudio-UniSE/model/model.py:38-51                  (microsoft/wavlm-base-plus: mean over all hidden states, no compression)
    The model body is THIRD-PARTY: `transformers` HubertModel / Wav2Vec2Model / WavLMModel (requirements pin 4.49.0 / 4.57.1; container has
    5.15.0), absent from /root/reference.  Its published algorithm is restated below with plain torch ops, module by module
    (transformers/models/hubert/modeling_hubert.py: HubertFeatureEncoder, HubertFeatureProjection,
    HubertPositionalConvEmbedding, HubertEncoder / HubertEncoderStableLayerNorm and their layers), driven by the HF
    state_dict.

Pinning
    tests/test_ssl_oracle_cpu.py runs this restatement against the container's own `transformers.HubertModel` /
    `Wav2Vec2Model` (random init, both flavours) - outputs agree to fp32 rounding.  There are no pretrained weights and no
    golden vectors for this stage anywhere in the reference (no network) => PARITY UNPINNED with respect to the published
    checkpoints; pinned with respect to the published architecture.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F


@dataclasses.dataclass(frozen=True)
class SSLSpec:
    conv_dim: Tuple[int, ...] = (512,) * 7
    conv_kernel: Tuple[int, ...] = (10, 3, 3, 3, 3, 2, 2)
    conv_stride: Tuple[int, ...] = (5, 2, 2, 2, 2, 2, 2)
    conv_bias: bool = False
    feat_extract_norm: str = "group"
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    do_stable_layer_norm: bool = False
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16
    layer_norm_eps: float = 1e-5
    pad: int = 160
    select: Tuple[int, ...] = ()
    compress_exponent: float = 0.3
    num_buckets: int = 0            # WavLM: 320 (gated relative position bias); 0 = HuBERT / wav2vec 2.0
    max_bucket_distance: int = 800


SPEC_HUBERT_BASE = SSLSpec()
# UniSE: Model.extract_semantic_features (QuarkAudio-UniSE/model/model.py:30,38-51): wavlm-base-plus, mean of all hidden states,
# the compression lines are commented out in the reference
SPEC_WAVLM_BASE_PLUS = SSLSpec(num_buckets=320, max_bucket_distance=800, compress_exponent=0.0)
SPEC_XLSR53 = SSLSpec(conv_bias=True, feat_extract_norm="layer", hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                      intermediate_size=4096, do_stable_layer_norm=True, select=(11, 14, 16))


def hf_config(spec: SSLSpec, kind: str = "hubert"):
    """The transformers config whose model this spec describes (used by the pinning test and the weight synthesiser)."""
    from transformers import HubertConfig, Wav2Vec2Config, WavLMConfig

    cls = {"hubert": HubertConfig, "wav2vec2": Wav2Vec2Config, "wavlm": WavLMConfig}[kind]
    extra = dict(num_buckets=spec.num_buckets, max_bucket_distance=spec.max_bucket_distance) if kind == "wavlm" else {}
    return cls(**extra, conv_dim=list(spec.conv_dim), conv_kernel=list(spec.conv_kernel), conv_stride=list(spec.conv_stride),
               conv_bias=spec.conv_bias, feat_extract_norm=spec.feat_extract_norm, hidden_size=spec.hidden_size,
               num_hidden_layers=spec.num_hidden_layers, num_attention_heads=spec.num_attention_heads,
               intermediate_size=spec.intermediate_size, do_stable_layer_norm=spec.do_stable_layer_norm,
               num_conv_pos_embeddings=spec.num_conv_pos_embeddings, num_conv_pos_embedding_groups=spec.num_conv_pos_embedding_groups,
               layer_norm_eps=spec.layer_norm_eps, num_feat_extract_layers=len(spec.conv_dim), hidden_dropout=0.0, attention_dropout=0.0,
               activation_dropout=0.0, feat_proj_dropout=0.0, layerdrop=0.0, mask_time_prob=0.0, mask_feature_prob=0.0)


def synth_state_dict(seed: int, spec: SSLSpec, kind: str = "hubert") -> Dict[str, torch.Tensor]:
    """Seeded random weights in the HF key layout: the HF module's own initialisation under torch.manual_seed, with the
    norm affine parameters and biases perturbed so that every term of the computation is exercised."""
    from transformers import HubertModel, Wav2Vec2Model, WavLMModel

    torch.manual_seed(seed)
    model = {"hubert": HubertModel, "wav2vec2": Wav2Vec2Model, "wavlm": WavLMModel}[kind](hf_config(spec, kind)).eval()
    g = torch.Generator().manual_seed(seed + 1)
    sd = {}
    for k, v in model.state_dict().items():
        v = v.detach().clone().float()
        if k == "masked_spec_embed":
            continue
        if k.endswith("layer_norm.weight") or k.endswith("final_layer_norm.weight"):
            v = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith(".bias"):
            v = 0.05 * torch.randn(v.shape, generator=g)
        elif k.endswith("gru_rel_pos_const"):
            v = 1.0 + 0.3 * torch.randn(v.shape, generator=g)
        elif k.endswith("rel_attn_embed.weight") or k.endswith("gru_rel_pos_linear.weight"):
            v = 0.5 * torch.randn(v.shape, generator=g)
        sd[k] = v
    return sd


def _pos_conv_weight(sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """weight_norm(dim=2) of HubertPositionalConvEmbedding: w = v * g / ||v|| with the norm over (out, in) per kernel tap."""
    pre = "encoder.pos_conv_embed.conv."
    if pre + "parametrizations.weight.original0" in sd:
        g, v = sd[pre + "parametrizations.weight.original0"], sd[pre + "parametrizations.weight.original1"]
    elif pre + "weight_g" in sd:
        g, v = sd[pre + "weight_g"], sd[pre + "weight_v"]
    else:
        return sd[pre + "weight"]
    return v * (g / v.norm(dim=(0, 1), keepdim=True))


def hidden_states(sd: Dict[str, torch.Tensor], wavs: torch.Tensor, spec: SSLSpec, taps: dict | None = None) -> List[torch.Tensor]:
    """HubertModel(wavs, output_hidden_states=True).hidden_states for un-padded-batch input (no attention mask)."""
    eps = spec.layer_norm_eps
    x = wavs[:, None]  # HubertFeatureEncoder.forward: input_values[:, None]
    n_conv = len(spec.conv_dim)
    for i in range(n_conv):
        pre = f"feature_extractor.conv_layers.{i}."
        x = F.conv1d(x, sd[pre + "conv.weight"], sd.get(pre + "conv.bias"), stride=spec.conv_stride[i])
        if spec.feat_extract_norm == "group":
            if i == 0:  # HubertGroupNormConvLayer: GroupNorm(num_groups = C) = per-channel normalisation over time
                x = F.group_norm(x, spec.conv_dim[0], sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"], 1e-5)
        else:       # HubertLayerNormConvLayer: LayerNorm over channels
            x = F.layer_norm(x.transpose(1, 2), (spec.conv_dim[i],), sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"],
                             1e-5).transpose(1, 2)
        x = F.gelu(x)
        if taps is not None and i == 0:
            taps["ssl.conv0"] = x.transpose(1, 2).contiguous()
    x = x.transpose(1, 2)  # [B, N, C]
    if taps is not None:
        taps["ssl.extract"] = x.contiguous()
    # HubertFeatureProjection
    x = F.layer_norm(x, (spec.conv_dim[-1],), sd["feature_projection.layer_norm.weight"], sd["feature_projection.layer_norm.bias"], eps)
    x = F.linear(x, sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"])
    # HubertPositionalConvEmbedding + HubertSamePadLayer
    k = spec.num_conv_pos_embeddings
    pos = F.conv1d(x.transpose(1, 2), _pos_conv_weight(sd), sd["encoder.pos_conv_embed.conv.bias"], padding=k // 2,
                   groups=spec.num_conv_pos_embedding_groups)
    if k % 2 == 0:
        pos = pos[:, :, :-1]
    x = x + F.gelu(pos).transpose(1, 2)
    d, H = spec.hidden_size, spec.num_attention_heads
    hd = d // H

    position_bias = None
    if spec.num_buckets:
        # WavLMAttention.compute_bias + _relative_positions_bucket (transformers/models/wavlm/modeling_wavlm.py): only layer 0
        # owns rel_attn_embed, every layer re-uses its bias
        n = x.shape[1]
        rel = torch.arange(n)[None, :] - torch.arange(n)[:, None]  # memory_position - context_position
        nb = spec.num_buckets // 2
        bucket = (rel > 0).long() * nb
        a = rel.abs()
        max_exact = nb // 2
        large = torch.log(a.float() / max_exact) / math.log(spec.max_bucket_distance / max_exact) * (nb - max_exact)
        large = torch.min((max_exact + large).long(), torch.full_like(a, nb - 1))
        bucket = bucket + torch.where(a < max_exact, a, large)
        position_bias = F.embedding(bucket, sd["encoder.layers.0.attention.rel_attn_embed.weight"]).permute(2, 0, 1)  # [H, N, N]

    def attention(h, pre):
        B, N, _ = h.shape
        q = F.linear(h, sd[pre + "q_proj.weight"], sd[pre + "q_proj.bias"]).view(B, N, H, hd).transpose(1, 2)
        kk = F.linear(h, sd[pre + "k_proj.weight"], sd[pre + "k_proj.bias"]).view(B, N, H, hd).transpose(1, 2)
        v = F.linear(h, sd[pre + "v_proj.weight"], sd[pre + "v_proj.bias"]).view(B, N, H, hd).transpose(1, 2)
        scores = (q * hd ** -0.5) @ kk.transpose(-1, -2)
        if position_bias is not None:  # WavLMAttention.forward: gate from the layer input, per (batch, head, query)
            gh = h.view(B, N, H, hd).permute(0, 2, 1, 3)
            proj = F.linear(gh, sd[pre + "gru_rel_pos_linear.weight"], sd[pre + "gru_rel_pos_linear.bias"])
            proj = proj.view(B, H, N, 2, 4).sum(-1)
            gate_a, gate_b = torch.sigmoid(proj).chunk(2, dim=-1)
            gate = gate_a * (gate_b * sd[pre + "gru_rel_pos_const"] - 1.0) + 2.0  # [B, H, N, 1]
            scores = scores + gate * position_bias[None]
        a = torch.softmax(scores, dim=-1) @ v
        return F.linear(a.transpose(1, 2).reshape(B, N, d), sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"])

    def ffn(h, pre):
        u = F.gelu(F.linear(h, sd[pre + "intermediate_dense.weight"], sd[pre + "intermediate_dense.bias"]))
        return F.linear(u, sd[pre + "output_dense.weight"], sd[pre + "output_dense.bias"])

    def ln(h, name):
        return F.layer_norm(h, (d,), sd[name + ".weight"], sd[name + ".bias"], eps)

    out: List[torch.Tensor] = []
    if not spec.do_stable_layer_norm:  # HubertEncoder / HubertEncoderLayer
        x = ln(x, "encoder.layer_norm")
        for i in range(spec.num_hidden_layers):
            out.append(x)
            pre = f"encoder.layers.{i}."
            x = ln(x + attention(x, pre + "attention."), pre + "layer_norm")
            x = ln(x + ffn(x, pre + "feed_forward."), pre + "final_layer_norm")
        out.append(x)
    else:                              # HubertEncoderStableLayerNorm / HubertEncoderLayerStableLayerNorm
        for i in range(spec.num_hidden_layers):
            out.append(x)
            pre = f"encoder.layers.{i}."
            x = x + attention(ln(x, pre + "layer_norm"), pre + "attention.")
            x = x + ffn(ln(x, pre + "final_layer_norm"), pre + "feed_forward.")
        out.append(ln(x, "encoder.layer_norm"))
    return out


def extract_features(sd: Dict[str, torch.Tensor], wavs: torch.Tensor, spec: SSLSpec, taps: dict | None = None) -> torch.Tensor:
    """audio_tokenizer.py:35-48 / 1.5:53-67 - wavs [B, T] -> feats_mix [B, frames, hidden]."""
    wavs = F.pad(wavs, (spec.pad, spec.pad))
    hs = hidden_states(sd, wavs, spec, taps)
    sel = list(spec.select) if spec.select else list(range(len(hs)))
    mix = torch.stack([hs[i] for i in sel], dim=1).mean(1)
    if taps is not None:
        taps["ssl.hidden0"] = hs[0].contiguous()
        taps["ssl.mean"] = mix
    if spec.compress_exponent <= 0:
        return mix
    symbol = (mix > 0).float() * 2 - 1
    return symbol * mix.abs() ** spec.compress_exponent
