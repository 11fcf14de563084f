"""Seeded synthetic weights and inputs (no checkpoints / datasets exist offline) - DATA GENERATION ONLY, no model arithmetic.

Used by bench.py, tools/ and (through the re-export in oracle/synth.py) by the tests and golden-vector generators.
Everything is drawn from numpy's PCG64 (`np.random.default_rng(seed)`), whose stream is specified and portable, so this
container and the GPU box regenerate bit-identical tensors from a seed; the golden vectors in tests/golden/ only store
seeds + outputs.

`hcodec10_state_dict` produces a flat state_dict with exactly the key names / shapes of the reference's
`Codec(None, None, None).state_dict()` (HCodec-1.0/vq/codec.py:21-136; checked by tests/test_oracle_cpu.py against the
reference's own modules) minus the training-only `semantic_decoder.*` entries, which `Codec.encode/decode` never touch
(codec.py:166-187).  Spec arguments are duck-typed: the product's `unified_audio_amd.HCodecSpec` and the oracle's spec
dataclasses carry the same field names.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

import dataclasses
from typing import Tuple

from .hcodec import SPEC_10


def _t(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


class _Gen:
    def __init__(self, seed: int):
        self.rng = np.random.default_rng(seed)
        self.sd: Dict[str, torch.Tensor] = {}

    def uniform(self, shape, bound):
        return self.rng.uniform(-bound, bound, size=shape).astype(np.float32)

    def conv(self, name, cout, cin, k, bias=True, wn=False, gain=1.0):
        """PyTorch default conv/linear init: U(+-1/sqrt(fan_in)) for weight and bias."""
        bound = gain / math.sqrt(cin * k)
        w = self.uniform((cout, cin, k), bound)
        if wn:  # torch weight_norm init: g = ||v||; perturbed so that folding g/||v|| is really exercised
            self.sd[name + ".weight_v"] = _t(w)
            g = np.sqrt((w.reshape(cout, -1) ** 2).sum(1)) * self.rng.uniform(0.8, 1.25, size=cout)
            self.sd[name + ".weight_g"] = _t(g.reshape(cout, 1, 1))
        else:
            self.sd[name + ".weight"] = _t(w)
        if bias:
            self.sd[name + ".bias"] = _t(self.uniform((cout,), bound))

    def linear(self, name, cout, cin, bias=True, gain=1.0):
        bound = gain / math.sqrt(cin)
        self.sd[name + ".weight"] = _t(self.uniform((cout, cin), bound))
        if bias:
            self.sd[name + ".bias"] = _t(self.uniform((cout,), bound))

    def norm(self, name, c, bias=True):
        self.sd[name + ".weight"] = _t(1.0 + 0.1 * self.rng.standard_normal(c))
        if bias:
            self.sd[name + ".bias"] = _t(0.05 * self.rng.standard_normal(c))

    def transformer(self, p, d, n_layers, inter=None):
        inter = inter or 4 * d
        for i in range(n_layers):
            lp = f"{p}.layers.{i}"
            b = 1.0 / math.sqrt(d)
            for nm, shape in (("weight_ih_l0", (4 * d, d)), ("weight_hh_l0", (4 * d, d)),
                              ("bias_ih_l0", (4 * d,)), ("bias_hh_l0", (4 * d,))):
                self.sd[f"{lp}.self_attn.rnn.{nm}"] = _t(self.uniform(shape, b))
            for nm in ("q_proj", "k_proj", "v_proj"):
                self.linear(f"{lp}.self_attn.{nm}", d, d, bias=True, gain=2.0)
            self.linear(f"{lp}.self_attn.o_proj", d, d, bias=False)
            self.linear(f"{lp}.mlp.w1", inter, d, bias=False)
            self.linear(f"{lp}.mlp.w2", d, inter, bias=False)
            self.linear(f"{lp}.mlp.w3", inter, d, bias=False)
            self.norm(f"{lp}.input_layernorm", d, bias=False)
            self.norm(f"{lp}.post_attention_layernorm", d, bias=False)


def _mimi(self, p, d, n_layers, ff):
    for i in range(n_layers):
        lp = f"{p}.layers.{i}"
        self.linear(lp + ".self_attn.in_proj", 3 * d, d, bias=False, gain=1.5)
        self.sd[lp + ".self_attn.in_proj_weight"] = self.sd.pop(lp + ".self_attn.in_proj.weight")
        self.linear(lp + ".self_attn.out_proj", d, d, bias=False)
        self.norm(lp + ".norm1", d)
        self.norm(lp + ".norm2", d)
        self.linear(lp + ".linear1", ff, d, bias=False)
        self.linear(lp + ".linear2", d, ff, bias=False)
        # LayerScale init is 0.01 in the reference; a trained model moves it, and a larger value makes the test discriminating
        self.sd[lp + ".layer_scale_1.scale"] = _t(self.rng.uniform(0.05, 0.3, size=d))
        self.sd[lp + ".layer_scale_2.scale"] = _t(self.rng.uniform(0.05, 0.3, size=d))


_Gen.mimi = _mimi


def mimi_state_dict(seed: int, d: int, n_layers: int, ff: int, prefix: str = "transformer"):
    """Seeded weights of one StreamingTransformer (keys `<prefix>.layers.N...`, mimi/transformer.py:436-594)."""
    g = _Gen(seed)
    g.mimi(prefix, d, n_layers, ff)
    return g.sd


def hcodec10_state_dict(seed: int = 1234, spec=SPEC_10, head_logmag_bias: float = 1.5,
                        head_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    g = _Gen(seed)
    nf, dim = spec.n_filters, spec.dimension
    # --- SEANet encoder (seanet.py:121-187)
    p = "encoder.model"
    g.conv(f"{p}.0.conv.conv", nf, 1, 7, wn=True, gain=2.0)
    c = nf
    for i, r in enumerate(spec.ratios):
        rb = f"{p}.{1 + 3 * i}"
        g.conv(f"{rb}.block.1.conv.conv", c // 2, c, 3, wn=True, gain=1.5)
        g.conv(f"{rb}.block.3.conv.conv", c, c // 2, 1, wn=True, gain=1.5)
        g.conv(f"{rb}.shortcut.conv.conv", c, c, 1, wn=True, gain=1.5)
        g.conv(f"{p}.{3 + 3 * i}.conv.conv", 2 * c, c, 2 * r, wn=True, gain=1.5)
        c *= 2
    assert c == dim, (c, dim)
    n = len(spec.ratios)
    g.transformer(f"{p}.{3 * n + 2}", dim, spec.enc_layers)
    g.conv(f"{p}.{3 * n + 5}.conv.conv", dim, dim, 4, wn=True, gain=1.5)
    # --- decoder (codec_decoder.py:14-56)
    d = spec.dec_dim
    g.conv("decoder.embed.up", 2 * d, 2 * spec.code_dim, 1, gain=1.5)
    g.conv("decoder.embed.dw", d, 1, 5)
    g.norm("decoder.norm", d)
    for i in range(spec.convnext_layers):
        cp = f"decoder.post_net.{i}"
        g.sd[cp + ".gamma"] = _t(g.rng.uniform(0.5, 1.5, size=d) / spec.convnext_layers)
        g.conv(cp + ".dwconv.conv", d, 1, 7)
        g.norm(cp + ".norm", d)
        g.linear(cp + ".pwconv1.linear", spec.dec_inter, d)
        g.linear(cp + ".pwconv2.linear", d, spec.dec_inter)
    g.norm("decoder.final_layer_norm", d)
    for i in (0, 1, 5, 6):
        rp = f"decoder.prior_net.{i}"
        g.norm(rp + ".norm1", d)
        g.conv(rp + ".conv1.conv", d, d, 3)
        g.norm(rp + ".norm2", d)
        g.conv(rp + ".conv2.conv", d, d, 3)
    g.transformer("decoder.prior_net.3", d, spec.dec_layers)
    g.norm("decoder.prior_net.7", d)
    # ISTFT head: keep log-magnitude in the linear regime of exp()/clip(100) (heads.py:139-140) by default
    g.linear("decoder.head.out", spec.n_fft + 2, d, gain=head_gain)
    nb = spec.n_fft // 2 + 1
    g.sd["decoder.head.out.bias"][:nb] += head_logmag_bias
    g.sd["decoder.head.istft.window"] = torch.hann_window(spec.n_fft)
    # --- RVQ codebooks: stage q has std 0.6 * 0.5^q
    for name in ("quantizer", "semantic_quantizer"):
        for q in range(spec.num_quantizers):
            e = g.rng.standard_normal((1, spec.codebook_size, spec.code_dim)) * (0.6 * 0.5 ** q)
            g.sd[f"{name}.layers.{q}._codebook.embed"] = _t(e)
    # --- semantic encoder (semantic_module.py:157-201)
    sc = spec.sem_ch
    g.conv("semantic_encoder.conv.conv", sc, spec.sem_in, 3, bias=False)
    for i, s in enumerate(spec.sem_strides):
        bp = f"semantic_encoder.conv_blocks.{i}"
        for u in range(2):
            g.conv(f"{bp}.res_units.{u}.conv1.conv", sc, sc, 3, bias=False)
            g.conv(f"{bp}.res_units.{u}.conv2", sc, sc, 1, bias=False)
        g.conv(f"{bp}.conv.conv", sc, sc, 3 if s == 1 else 2 * s, bias=True)
    g.conv("semantic_encoder.conv2.conv", spec.code_dim, sc, 3, bias=False)
    if spec.adaptive:  # H-Codec 1.5 stacks (codec_adaptive.py:49-64; mimi/transformer.py:436-594,701-739)
        for name in ("semantic_aggregator", "acoustic_aggregator"):
            g.sd[name + ".query_embedding"] = _t(g.rng.standard_normal((1, spec.code_dim, 1)))
            g.mimi(name + ".transformer.transformer", spec.code_dim, spec.agg_layers, spec.agg_ff)
        g.mimi("bottleneck_transformer.transformer", 2 * spec.code_dim, spec.bt_layers, spec.bt_ff)
    return g.sd


def stress_state_dict(sd: Dict[str, torch.Tensor], lstm_gain: float = 4.0, layer_scale: float = 1.0, head_clip_bias: float = 4.6
                      ) -> Dict[str, torch.Tensor]:
    """RANGE-STRESS variant of a codec state_dict: the dynamic ranges a trained checkpoint can have and the default
    seeded weights do not - applied to the tensors of `sd` by name, so the reference (oracle/gen_golden.py) and every test regenerate the
    same weights from the seed:
      * nn.LSTM weights x `lstm_gain` (W_ih and W_hh): the gates saturate (|pre-activation| of tens: sigmoid / tanh in their flat ends);
      * LayerScale of the mimi stacks (reference init 0.01) and the ConvNeXt gammas (init 1 / n_layers) -> about `layer_scale`: the
        residual branches carry full-size updates, activations grow layer by layer;
      * ISTFT head: the log-magnitude bias of the magnitude bins raised to `head_clip_bias` (ln 100 = 4.6): about half of the bins run
        into the clip at 100 (heads.py:139-140), the rest stay in the exponential's steep part."""
    out = {}
    n_gamma = {}
    for k in sd:
        if k.endswith(".gamma"):
            pre = k.rsplit(".", 2)[0]
            n_gamma[pre] = n_gamma.get(pre, 0) + 1
    for k, v in sd.items():
        if k.endswith(("rnn.weight_hh_l0", "rnn.weight_ih_l0")):
            v = v * lstm_gain
        elif k.endswith((".layer_scale_1.scale", ".layer_scale_2.scale")):
            v = v / v.mean() * layer_scale
        elif k.endswith(".gamma"):
            v = v * (layer_scale * n_gamma[k.rsplit(".", 2)[0]])
        elif k == "decoder.head.out.bias":
            nb = v.numel() // 2
            v = v.clone()
            v[:nb] += head_clip_bias - float(v[:nb].mean())
        out[k] = v
    return out


def stress_lm_state_dict(sd: Dict[str, torch.Tensor], head_gain: float = 0.05, qk_gain: float = 3.0) -> Dict[str, torch.Tensor]:
    """Range-stress variant of the UniSE LM weights: the output head scaled DOWN (top-2 logit gaps of the greedy arg-max shrink by
    `head_gain`: many near-degenerate decisions, which the token goldens store gap by gap) and the q / k projections scaled up (peaky
    attention: softmax rows dominated by one key, exp() of large negative scores for the rest)."""
    out = {}
    for k, v in sd.items():
        if k == "output_head.weight":
            v = v * head_gain
        elif k.endswith(("self_attn.q_proj.weight", "self_attn.k_proj.weight")):
            v = v * qk_gain
        out[k] = v
    return out


def synth_wav(seed: int, batch: int, samples: int, sr: int = 16000) -> torch.Tensor:
    """band-limited noise (0.1*randn low-passed to ~4 kHz) + 3 sinusoids, peak 0.5."""
    rng = np.random.default_rng(seed)
    noise = 0.1 * rng.standard_normal((batch, samples + 32))
    kern = np.hanning(9)
    kern /= kern.sum()  # ~4 kHz low-pass at 16 kHz
    noise = np.stack([np.convolve(n, kern, mode="same") for n in noise])[:, 16:16 + samples]
    t = np.arange(samples) / sr
    tones = np.zeros((batch, samples))
    for b in range(batch):
        for f, a in zip(rng.uniform(80, 3000, size=3), (0.3, 0.2, 0.1)):
            tones[b] += a * np.sin(2 * np.pi * f * t + rng.uniform(0, 2 * np.pi))
    x = noise + tones
    x = 0.5 * x / np.abs(x).max(axis=1, keepdims=True)
    return _t(x)


def synth_feat(seed: int, batch: int, frames: int, dim: int = 768) -> torch.Tensor:
    """randn smoothed over time with a 15-tap moving average, then sign*|x|^0.3
    (audio_tokenizer.py:44-47).  Returned channel-first [B, dim, frames] like Codec.encode's `feat`."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((batch, dim, frames + 14))
    c = np.cumsum(np.concatenate([np.zeros((batch, dim, 1)), x], axis=2), axis=2)
    x = (c[:, :, 15:] - c[:, :, :-15]) / 15.0 * math.sqrt(15.0)
    x = np.sign(x) * np.abs(x) ** 0.3
    return _t(x)


def hcodec20_state_dict(seed: int, spec, head_logmag_bias: float = 1.5) -> Dict[str, torch.Tensor]:
    """H-Codec 2.0 keys (HCodec-2.0/vq/codec.py:17-56 with codec_encoder.py / codec_decoder.py), minus semantic_decoder.*"""
    g = _Gen(seed)
    nb = spec.n_fft // 2 + 1

    def convnext(p, d, inter, n):
        g.sd[p + ".gamma"] = _t(g.rng.uniform(0.5, 1.5, size=d) / n)
        g.conv(p + ".dwconv.conv", d, 1, 7)
        g.norm(p + ".norm", d)
        g.linear(p + ".pwconv1.linear", inter, d)
        g.linear(p + ".pwconv2.linear", d, inter)

    d = spec.enc_dim
    g.conv("encoder.embed.conv", d, 2 * nb, 3, gain=2.0)
    g.norm("encoder.norm", d)
    for i in range(spec.enc_convnext_layers):
        convnext(f"encoder.prior_net.{i}", d, spec.enc_inter, spec.enc_convnext_layers)
    g.transformer("encoder.post_net.1", d, spec.enc_transformer_layers, spec.tr_inter(d))
    g.norm("encoder.final_layer_norm", d)
    g.conv("encoder.out.conv", spec.dimension, d, 2 * spec.stride + 1, gain=1.5)
    d = spec.dec_dim
    g.conv("decoder.embed.conv", d, 2 * spec.dimension, spec.stride + 1, gain=1.5)
    g.norm("decoder.norm", d)
    for i in range(spec.dec_convnext_layers):
        convnext(f"decoder.post_net.{i}", d, spec.dec_inter, spec.dec_convnext_layers)
    g.norm("decoder.final_layer_norm", d)
    for i in (0, 1, 5, 6):
        rp = f"decoder.prior_net.{i}"
        g.norm(rp + ".norm1", d)
        g.conv(rp + ".conv1.conv", d, d, 3)
        g.norm(rp + ".norm2", d)
        g.conv(rp + ".conv2.conv", d, d, 3)
    g.transformer("decoder.prior_net.3", d, spec.dec_transformer_layers, spec.tr_inter(d))
    g.norm("decoder.prior_net.7", d)
    g.linear("decoder.head.out", spec.n_fft + 2, d)
    g.sd["decoder.head.out.bias"][:nb] += head_logmag_bias
    g.sd["decoder.head.istft.window"] = torch.hann_window(spec.n_fft)
    for name in ("quantizer", "semantic_quantizer"):
        for q in range(spec.num_quantizers):
            e = g.rng.standard_normal((1, spec.codebook_size, spec.dimension)) * (0.6 * 0.7 ** q)
            g.sd[f"{name}.layers.{q}._codebook.embed"] = _t(e)
    sc = spec.sem_ch
    g.conv("semantic_encoder.conv.conv", sc, spec.sem_in, 3, bias=False)
    for i, s in enumerate(spec.sem_strides):
        bp = f"semantic_encoder.conv_blocks.{i}"
        for u in range(2):
            g.conv(f"{bp}.res_units.{u}.conv1.conv", sc, sc, 3, bias=False)
            g.conv(f"{bp}.res_units.{u}.conv2", sc, sc, 1, bias=False)
        g.conv(f"{bp}.conv.conv", sc, sc, 3 if s == 1 else 2 * s, bias=True)
    g.conv("semantic_encoder.conv2.conv", spec.dimension, sc, 3, bias=False)
    return g.sd


def synth_wav_fullband(seed: int, batch: int, samples: int) -> torch.Tensor:
    """White noise + tones, peak 0.5: every STFT bin carries energy, so log-magnitude / phase are well conditioned
    (the phase of a near-empty bin is numerically arbitrary in ANY implementation, the reference's included)."""
    rng = np.random.default_rng(seed)
    x = 0.2 * rng.standard_normal((batch, samples))
    t = np.arange(samples) / 48000.0
    for b in range(batch):
        for f, a in zip(rng.uniform(100, 8000, size=3), (0.3, 0.2, 0.1)):
            x[b] += a * np.sin(2 * np.pi * f * t + rng.uniform(0, 2 * np.pi))
    return _t(0.5 * x / np.abs(x).max(axis=1, keepdims=True))


# ---- shape descriptors for callers that must not import the oracle (bench.py) ------------------------------------------------

@dataclasses.dataclass(frozen=True)
class Shapes20:
    """H-Codec 2.0 tensor shapes (HCodec-2.0/conf/large_12.5hz_config.yaml); same field names as oracle.hcodec20_ref.HCodec20Spec."""

    enc_dim: int = 1536
    enc_inter: int = 4608
    enc_convnext_layers: int = 24
    enc_transformer_layers: int = 2
    dimension: int = 512
    n_fft: int = 1920
    hop: int = 960
    stride: int = 4
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
    tr_inter_cap: int = 4096

    @property
    def frame_hop(self) -> int:
        return self.hop * self.stride

    def tr_inter(self, d: int) -> int:
        return min(4 * d, self.tr_inter_cap)


@dataclasses.dataclass(frozen=True)
class LMShapes:
    """UniSE LM tensor shapes (QuarkAudio-UniSE/conf/config.yaml:131-146); same field names as oracle.llm_ref.LMSpec."""

    hidden: int = 512
    n_layers: int = 12
    n_heads: int = 8
    global_size: int = 4096
    semantic_size: int = 8192
    feats_dim: int = 768
    num_tasks: int = 3

    @property
    def intermediate(self) -> int:
        return 4 * self.hidden

    @property
    def vocab(self) -> int:
        return 3 + self.global_size + self.semantic_size


LM_SHAPES_UNISE = LMShapes()


def lm_state_dict(seed: int, spec=None) -> Dict[str, torch.Tensor]:
    """Seeded random weights (numpy PCG64) with the reference's key names.  Embedding / head scales are chosen so the
    greedy arg-max is well separated most of the time, like a trained model's."""
    spec = spec or LM_SHAPES_UNISE
    rng = np.random.default_rng(seed)
    d, v = spec.hidden, spec.vocab

    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))

    def lin(o, i, gain=1.0):
        return t(rng.uniform(-gain / math.sqrt(i), gain / math.sqrt(i), size=(o, i)))

    sd = {
        "task_embedding.weight": t(rng.standard_normal((spec.num_tasks, d))),
        "enroll_sos_embedding.weight": t(rng.standard_normal((1, d))),
        "mix_sos_embedding.weight": t(rng.standard_normal((1, d))),
        "adapter.weight": lin(d, spec.feats_dim),
        "adapter.bias": t(rng.uniform(-0.03, 0.03, size=d)),
        "codec_embedding.weight": t(rng.standard_normal((v, d))),
        "output_head.weight": lin(v, d, gain=3.0),
        "norm.weight": t(1.0 + 0.1 * rng.standard_normal(d)),
    }
    for i in range(spec.n_layers):
        p = f"layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            sd[f"{p}.self_attn.{n}.weight"] = lin(d, d, gain=1.7)
        sd[f"{p}.mlp.gate_proj.weight"] = lin(spec.intermediate, d, gain=1.7)
        sd[f"{p}.mlp.up_proj.weight"] = lin(spec.intermediate, d, gain=1.7)
        sd[f"{p}.mlp.down_proj.weight"] = lin(d, spec.intermediate, gain=1.7)
        sd[f"{p}.input_layernorm.weight"] = t(1.0 + 0.1 * rng.standard_normal(d))
        sd[f"{p}.post_attention_layernorm.weight"] = t(1.0 + 0.1 * rng.standard_normal(d))
    return sd


def synth_feats(seed: int, batch: int, frames: int, dim: int = 768) -> torch.Tensor:
    """WavLM-like features [B, frames, dim]: smoothed noise (model.py:38-51 takes the mean of 13 hidden states)."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((batch, frames + 4, dim))
    x = (x[:, :-4] + x[:, 1:-3] + x[:, 2:-2] + x[:, 3:-1] + x[:, 4:]) / math.sqrt(5.0)
    return torch.from_numpy(x.astype(np.float32))




# ---- BiCodec detokenizer (QuarkAudio-UniSE/model/bicodec): quantizer look-up side, speaker detokenizer, prenet, wave generator ----

@dataclasses.dataclass(frozen=True)
class BiCodecShapes:
    """Shapes of the Spark-TTS BiCodec checkpoint the reference loads (model/bicodec/bicodec.py:70-115); same field names as
    oracle.bicodec_ref.BiCodecSpec and unified_audio_amd.BiCodecSpec."""

    latent_dim: int = 1024
    codebook_size: int = 8192
    codebook_dim: int = 8
    mel_dim: int = 128
    spk_latent_dim: int = 128
    token_num: int = 32
    fsq_levels: Tuple[int, ...] = (4, 4, 4, 4, 4, 4)
    vocos_dim: int = 384
    vocos_inter: int = 2048
    vocos_layers: int = 12
    gen_channels: int = 1536
    rates: Tuple[int, ...] = (8, 5, 4, 2)
    kernel_sizes: Tuple[int, ...] = (16, 11, 8, 4)


def bicodec_state_dict(seed: int, spec=None) -> Dict[str, torch.Tensor]:
    """Seeded weights with the key names / shapes of the parts of `BiCodec.state_dict()` that `detokenize` reads
    (bicodec.py:193-199): quantizer.codebook / out_project, speaker_encoder.quantizer.project_out / project, prenet.*, decoder.*
    (checked against the reference's own modules by tests/test_bicodec_oracle_cpu.py).  weight_norm parameters come as
    weight_g / weight_v like torch.nn.utils.weight_norm stores them (dim 0: for ConvTranspose1d that is the INPUT channel)."""
    spec = spec or BiCodecShapes()
    g = _Gen(seed)
    rng, sd = g.rng, g.sd
    L, C, I = spec.latent_dim, spec.vocos_dim, spec.vocos_inter

    def wn_generic(name, shape, fan_in, gain=1.0):
        w = g.uniform(shape, gain / math.sqrt(fan_in))
        sd[name + ".weight_v"] = _t(w)
        n0 = np.sqrt((w.reshape(shape[0], -1) ** 2).sum(1)) * rng.uniform(0.8, 1.25, size=shape[0])
        sd[name + ".weight_g"] = _t(n0.reshape((shape[0],) + (1,) * (len(shape) - 1)))

    sd["quantizer.codebook.weight"] = _t(rng.standard_normal((spec.codebook_size, spec.codebook_dim)))
    g.conv("quantizer.out_project", L, spec.codebook_dim, 1, wn=True, gain=2.0)
    g.linear("speaker_encoder.quantizer.project_out", spec.spk_latent_dim, len(spec.fsq_levels), gain=2.0)
    g.linear("speaker_encoder.project", L, spec.spk_latent_dim * spec.token_num, gain=2.0)

    def vocos(p, n_layers, cond):
        g.conv(p + ".embed", C, C, 7)
        if cond:
            for nm, base in (("scale", 1.0), ("shift", 0.0)):
                sd[f"{p}.norm.{nm}.weight"] = _t(g.uniform((C, cond), 0.5 / math.sqrt(cond)))
                sd[f"{p}.norm.{nm}.bias"] = _t(base + 0.1 * rng.standard_normal(C))
        else:
            g.norm(p + ".norm", C)
        for i in range(n_layers):
            q = f"{p}.convnext.{i}"
            sd[q + ".gamma"] = _t(rng.uniform(0.5, 1.5, size=C) / n_layers)
            g.conv(q + ".dwconv", C, 1, 7)
            if cond:
                for nm, base in (("scale", 1.0), ("shift", 0.0)):
                    sd[f"{q}.norm.{nm}.weight"] = _t(g.uniform((C, cond), 0.5 / math.sqrt(cond)))
                    sd[f"{q}.norm.{nm}.bias"] = _t(base + 0.1 * rng.standard_normal(C))
            else:
                g.norm(q + ".norm", C)
            g.linear(q + ".pwconv1", I, C)
            g.linear(q + ".pwconv2", C, I)
        g.norm(p + ".final_layer_norm", C)

    g.linear("prenet.linear_pre", C, L)
    for i in range(2):
        vocos(f"prenet.downsample.{i}.1", 2, 0)
    vocos("prenet.vocos_backbone", spec.vocos_layers, L)
    g.linear("prenet.linear", L, C)

    ch = spec.gen_channels
    g.conv("decoder.model.0", ch, L, 7, wn=True)
    for i, (k, s) in enumerate(zip(spec.kernel_sizes, spec.rates)):
        cin, cout = ch // 2 ** i, ch // 2 ** (i + 1)
        p = f"decoder.model.{i + 1}.block"
        sd[p + ".0.alpha"] = _t(rng.uniform(0.5, 1.5, size=(1, cin, 1)))
        wn_generic(p + ".1", (cin, cout, k), cin * k / s, gain=1.5)  # ConvTranspose1d weight [C_in, C_out, k]
        sd[p + ".1.bias"] = _t(g.uniform((cout,), 0.05))
        for j in range(3):
            u = f"{p}.{j + 2}.block"
            sd[u + ".0.alpha"] = _t(rng.uniform(0.5, 1.5, size=(1, cout, 1)))
            g.conv(u + ".1", cout, cout, 7, wn=True)
            sd[u + ".2.alpha"] = _t(rng.uniform(0.5, 1.5, size=(1, cout, 1)))
            g.conv(u + ".3", cout, cout, 1, wn=True)
    n = len(spec.rates)
    cl = ch // 2 ** n
    sd[f"decoder.model.{n + 1}.alpha"] = _t(rng.uniform(0.5, 1.5, size=(1, cl, 1)))
    g.conv(f"decoder.model.{n + 2}", 1, cl, 7, wn=True, gain=0.35)  # keeps tanh out of saturation
    return sd


def bicodec_tokens(seed: int, batch: int, frames: int, spec=None):
    """Seeded semantic tokens [B, frames] and global tokens [B, 1, token_num] (int64) in range."""
    spec = spec or BiCodecShapes()
    rng = np.random.default_rng(seed)
    n_glob = 1
    for lv in spec.fsq_levels:
        n_glob *= lv
    sem = torch.from_numpy(rng.integers(0, spec.codebook_size, size=(batch, frames)).astype(np.int64))
    glob = torch.from_numpy(rng.integers(0, n_glob, size=(batch, 1, spec.token_num)).astype(np.int64))
    return sem, glob


def ssl_state_dict(spec, seed: int = 21) -> Dict[str, torch.Tensor]:
    """Seeded random weights in the transformers HubertModel / Wav2Vec2Model key layout (shapes only depend on the spec)."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *shape, scale=0.02: torch.randn(*shape, generator=g) * scale  # noqa: E731
    sd = {}
    cin = 1
    for i, (c, k) in enumerate(zip(spec.conv_dim, spec.conv_kernel)):
        pre = f"feature_extractor.conv_layers.{i}."
        sd[pre + "conv.weight"] = rn(c, cin, k, scale=(2.0 / (cin * k)) ** 0.5)
        if spec.conv_bias:
            sd[pre + "conv.bias"] = rn(c)
        if spec.feat_extract_norm == "layer" or i == 0:
            sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"] = 1.0 + rn(c, scale=0.1), rn(c)
        cin = c
    d, inter = spec.hidden_size, spec.intermediate_size
    sd["feature_projection.layer_norm.weight"], sd["feature_projection.layer_norm.bias"] = 1.0 + rn(cin, scale=0.1), rn(cin)
    sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"] = rn(d, cin, scale=cin ** -0.5), rn(d)
    cg = d // spec.num_conv_pos_embedding_groups
    sd["encoder.pos_conv_embed.conv.weight"] = rn(d, cg, spec.num_conv_pos_embeddings, scale=(cg * spec.num_conv_pos_embeddings) ** -0.5)
    sd["encoder.pos_conv_embed.conv.bias"] = rn(d)
    sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"] = 1.0 + rn(d, scale=0.1), rn(d)
    for i in range(spec.num_hidden_layers):
        pre = f"encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[pre + f"attention.{nm}.weight"], sd[pre + f"attention.{nm}.bias"] = rn(d, d, scale=d ** -0.5), rn(d)
        sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"] = 1.0 + rn(d, scale=0.1), rn(d)
        sd[pre + "feed_forward.intermediate_dense.weight"], sd[pre + "feed_forward.intermediate_dense.bias"] = rn(inter, d, scale=d ** -0.5), rn(inter)
        sd[pre + "feed_forward.output_dense.weight"], sd[pre + "feed_forward.output_dense.bias"] = rn(d, inter, scale=inter ** -0.5), rn(d)
        sd[pre + "final_layer_norm.weight"], sd[pre + "final_layer_norm.bias"] = 1.0 + rn(d, scale=0.1), rn(d)
        if spec.num_buckets:  # WavLM
            hd = d // spec.num_attention_heads
            sd[pre + "attention.gru_rel_pos_linear.weight"], sd[pre + "attention.gru_rel_pos_linear.bias"] = rn(8, hd, scale=hd ** -0.5), rn(8)
            sd[pre + "attention.gru_rel_pos_const"] = 1.0 + rn(1, spec.num_attention_heads, 1, 1, scale=0.1)
            if i == 0:
                sd[pre + "attention.rel_attn_embed.weight"] = rn(spec.num_buckets, spec.num_attention_heads, scale=0.5)
    return sd
