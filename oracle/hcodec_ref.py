"""TEST INFRASTRUCTURE ONLY - CPU restatement (plain PyTorch fp32 ops) of the reference's H-Codec 1.0
`Codec.encode` / `Codec.decode` hot path.  Nothing in the product path may import this module; only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg do.

Every function cites the reference lines it follows (paths relative to
/root/reference/QuarkAudio-HCodec/HCodec-1.0/).  The restatement is functional (driven by a flat
state_dict with the reference's own key names), so that the same seeded weights can be handed to
(a) the reference's modules in this container (oracle/ref_shim.py, used to validate this file and to
generate tests/golden/*), (b) this file on the GPU box, where /root/reference does not exist, and
(c) the HIP library under test.

Pinning status: validated against the reference's own modules on seeded weights by
tests/test_oracle_vs_reference.py (runs where /root/reference exists) and against the committed
vectors in tests/golden/ everywhere.  The RVQ stage is third-party in the reference
(vector-quantize-pytorch==1.22.15, not vendored, not installed); `rvq_search` / `rvq_lookup` here are pinned bit-exact to
the reference's in-tree statement of the same algorithm, vq/core_vq.py:223-231,394-412 (tests/test_rvq_pin_cpu.py, live
and through tests/golden/rvq_corevq_*.npz); the pip package itself has never been diffed.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


@dataclass(frozen=True)
class HCodecSpec:
    """Architecture constants of H-Codec 1.0 (hard-coded in vq/codec.py:30-136)."""

    n_filters: int = 32  # codec.py:33
    ratios: Tuple[int, ...] = (2, 4, 5, 8)  # codec.py:33 gives [8,5,4,2]; seanet.py:114 reverses it
    dimension: int = 512  # codec.py:33
    enc_heads: int = 8  # seanet.py:166
    enc_layers: int = 2  # seanet.py:167
    sem_in: int = 768  # codec.py:122
    sem_ch: int = 768  # codec.py:123
    sem_strides: Tuple[int, ...] = (2, 1)  # codec.py:126
    code_dim: int = 512  # codec.py:102
    codebook_size: int = 1024  # codec.py:103
    num_quantizers: int = 4  # codec.py:104
    dec_dim: int = 768  # codec.py:44
    dec_inter: int = 2304  # codec.py:45
    dec_heads: int = 8  # codec_decoder.py:44
    dec_layers: int = 2  # codec_decoder.py:45
    convnext_layers: int = 12  # codec_decoder.py:20
    n_fft: int = 1280  # codec_decoder.py:21
    hop: int = 320  # codec_decoder.py:22
    gn_groups: int = 32  # conv.py:259
    # --- H-Codec 1.5 only (HCodec-1.5/conf/config_adaptive_v3.yaml:65-111); adaptive=False is H-Codec 1.0
    adaptive: bool = False
    agg_layers: int = 32  # aggregators.*.num_layers
    agg_heads: int = 8
    agg_ff: int = 2048
    bt_layers: int = 32  # transformer_kwargs.num_layers (bottleneck, d_model = 2 * code_dim)
    bt_heads: int = 8
    bt_ff: int = 2048
    threshold: float = 0.6  # manual_threshold
    max_tokens_per_group: int = 8
    # causal=True: the variant every block parameterises (SConv1d conv.py:203-206, Conv1d / ConvTranspose1d vq/conv.py:44-47,76-79,
    # Transformer transformer.py:470-475); vq/codec.py:31 hard-codes False, so only reference MODULES built with causal=True pin it
    causal: bool = False
    # H-Codec 1.5 stacks: config_adaptive_v3.yaml:84,87,93,96 (aggregators causal / context_frames), :103,105 (bottleneck)
    agg_causal: bool = False
    agg_context: int = 16
    bt_causal: bool = False
    bt_context: int = 16

    @property
    def enc_hop(self) -> int:
        return int(math.prod(self.ratios)) * 2  # strides + the final stride-2 conv (seanet.py:173-178)


SPEC_10 = HCodecSpec()
# H-Codec 1.5 (config_adaptive_v3.yaml): SEANet stride order 8,5,4,2 (the config lists [2,4,5,8] and seanet.py:114 reverses
# it), XLSR features (1024), decoder width 1024, plus the adaptive-frame-rate stacks.
SPEC_15 = HCodecSpec(ratios=(8, 5, 4, 2), sem_in=1024, sem_ch=1024, dec_dim=1024, dec_inter=2304, adaptive=True)


# ----------------------------------------------------------------------------- padding / convs

def _fold_weight_norm(sd: SD, p: str) -> Tensor:
    """torch.nn.utils.weight_norm (dim=0): w = g * v / ||v||, norm over all dims but 0 (conv.py:25-26 of
    encoder_modules)."""
    v, g = sd[p + ".weight_v"], sd[p + ".weight_g"]
    return v * (g / v.flatten(1).norm(dim=1).view(-1, 1, 1))


def _extra_padding(length: int, k_eff: int, stride: int, padding_total: int) -> int:
    """encoder_modules/conv.py:54-61."""
    n_frames = (length - k_eff + padding_total) / stride + 1
    ideal = (math.ceil(n_frames) - 1) * stride + (k_eff - padding_total)
    return ideal - length


def _pad1d_reflect(x: Tensor, left: int, right: int) -> Tensor:
    """encoder_modules/conv.py:79-96 (reflect branch incl. the short-input case)."""
    length = x.shape[-1]
    max_pad = max(left, right)
    extra = 0
    if length <= max_pad:
        extra = max_pad - length + 1
        x = F.pad(x, (0, extra))
    y = F.pad(x, (left, right), mode="reflect")
    return y[..., : y.shape[-1] - extra]


def sconv1d(x: Tensor, w: Tensor, b: Optional[Tensor], stride: int, dilation: int = 1, causal: bool = False) -> Tensor:
    """SConv1d.forward, encoder_modules/conv.py:195-211 (causal: the whole padding goes to the left, :203-206)."""
    k = w.shape[-1]
    k_eff = (k - 1) * dilation + 1
    padding_total = k_eff - stride
    extra = _extra_padding(x.shape[-1], k_eff, stride, padding_total)
    if causal:
        x = _pad1d_reflect(x, padding_total, extra)
    else:
        right = padding_total // 2
        left = padding_total - right
        x = _pad1d_reflect(x, left, right + extra)
    return F.conv1d(x, w, b, stride=stride, dilation=dilation)


def _wn_sconv(sd: SD, p: str, x: Tensor, stride: int = 1, causal: bool = False) -> Tensor:
    return sconv1d(x, _fold_weight_norm(sd, p + ".conv.conv"), sd[p + ".conv.conv.bias"], stride, causal=causal)


def seanet_resblock(sd: SD, p: str, x: Tensor, causal: bool = False) -> Tensor:
    """SEANetResnetBlock.forward, seanet.py:34-76: shortcut_1x1(x) + 1x1(ELU(k3(ELU(x))))."""
    h = _wn_sconv(sd, p + ".block.1", F.elu(x), causal=causal)
    h = _wn_sconv(sd, p + ".block.3", F.elu(h), causal=causal)
    return _wn_sconv(sd, p + ".shortcut", x, causal=causal) + h


# ----------------------------------------------------------------------------- transformer (+LSTM)

def rms_norm(x: Tensor, w: Tensor, eps: float = 1e-6) -> Tensor:
    """encoder_modules/transformer.py:77-96 (native F.rms_norm branch)."""
    return F.rms_norm(x, (x.shape[-1],), weight=w, eps=eps)


def lstm_forward(x: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor) -> Tensor:
    """nn.LSTM(d, d, 1, batch_first=True) with zero initial state (transformer.py:115,133).
    Gate order i, f, g, o (PyTorch)."""
    d = w_hh.shape[1]
    m = torch.nn.LSTM(x.shape[-1], d, 1, batch_first=True)
    with torch.no_grad():
        m.weight_ih_l0.copy_(w_ih)
        m.weight_hh_l0.copy_(w_hh)
        m.bias_ih_l0.copy_(b_ih)
        m.bias_hh_l0.copy_(b_hh)
        y, _ = m(x)
    return y


def rope_tables(n: int, head_dim: int, theta: float = 10000.0) -> Tuple[Tensor, Tensor]:
    """transformer.py:33-44,69-74: inv_freq = theta^(-2i/d); emb = cat(freqs, freqs)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    freqs = torch.arange(n).float()[:, None] * inv_freq[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def _rotate_half(x: Tensor) -> Tensor:  # transformer.py:211-215
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def attention_block(sd: SD, p: str, x: Tensor, n_heads: int, taps=None, causal: bool = False, left_context: int = 0) -> Tensor:
    """Attention.forward, transformer.py:121-180: LSTM -> q/k/v (+bias) -> rotate-half RoPE ->
    softmax(QK^T * d^-0.5) in fp32 -> O (no bias).  Non-causal: mask is None; causal: tril mask, with left_context > 0 the
    sliding window tril * triu(-left_context + 1) (transformer.py:437-475, added as a -inf bias :169-175)."""
    b, n, d = x.shape
    hd = d // n_heads
    x = lstm_forward(x, sd[p + ".rnn.weight_ih_l0"], sd[p + ".rnn.weight_hh_l0"],
                     sd[p + ".rnn.bias_ih_l0"], sd[p + ".rnn.bias_hh_l0"])
    if taps is not None:
        taps[p + ".rnn"] = x
    q = F.linear(x, sd[p + ".q_proj.weight"], sd[p + ".q_proj.bias"]).view(b, n, n_heads, hd).transpose(1, 2)
    k = F.linear(x, sd[p + ".k_proj.weight"], sd[p + ".k_proj.bias"]).view(b, n, n_heads, hd).transpose(1, 2)
    v = F.linear(x, sd[p + ".v_proj.weight"], sd[p + ".v_proj.bias"]).view(b, n, n_heads, hd).transpose(1, 2)
    cos, sin = rope_tables(n, hd)
    q = q * cos + _rotate_half(q) * sin
    k = k * cos + _rotate_half(k) * sin
    w = torch.matmul(q, k.transpose(2, 3)) * hd ** -0.5
    if causal:
        seen = torch.tril(torch.ones(n, n, dtype=torch.bool))
        if left_context > 0:
            seen = seen & torch.triu(torch.ones(n, n, dtype=torch.bool), diagonal=-left_context + 1)
        w = w + torch.zeros_like(w).masked_fill_(~seen, float("-inf"))
    w = F.softmax(w, dim=-1, dtype=torch.float32)
    o = torch.matmul(w, v).transpose(1, 2).reshape(b, n, d)
    return F.linear(o, sd[p + ".o_proj.weight"])


def transformer(sd: SD, p: str, x: Tensor, n_layers: int, n_heads: int, taps=None, causal: bool = False, left_context: int = 0) -> Tensor:
    """Transformer.forward / TransformerLayer.forward, transformer.py:338-393,444-489."""
    for i in range(n_layers):
        lp = f"{p}.layers.{i}"
        h = rms_norm(x, sd[lp + ".input_layernorm.weight"])
        x = x + attention_block(sd, lp + ".self_attn", h, n_heads, taps, causal, left_context)
        h = rms_norm(x, sd[lp + ".post_attention_layernorm.weight"])
        h = F.linear(F.silu(F.linear(h, sd[lp + ".mlp.w1.weight"])) * F.linear(h, sd[lp + ".mlp.w3.weight"]),
                     sd[lp + ".mlp.w2.weight"])  # MLP.forward, transformer.py:218-226
        x = x + h
        if taps is not None:
            taps[lp] = x
    return x


# ----------------------------------------------------------------------------- encode side

def seanet_encoder(sd: SD, wav: Tensor, spec: HCodecSpec = SPEC_10, taps=None) -> Tensor:
    """SEANetEncoder.forward, seanet.py:121-187,206-208.  wav [B,1,T] -> emb [B,dimension,N25]."""
    p = "encoder.model"
    cz = spec.causal
    x = _wn_sconv(sd, f"{p}.0", wav, causal=cz)
    if taps is not None:
        taps["enc.conv0"] = x
    for i, r in enumerate(spec.ratios):
        x = seanet_resblock(sd, f"{p}.{1 + 3 * i}", x, cz)
        x = _wn_sconv(sd, f"{p}.{3 + 3 * i}", F.elu(x), stride=r, causal=cz)
        if taps is not None:
            taps[f"enc.stage{i}"] = x
    n = len(spec.ratios)
    x = transformer(sd, f"{p}.{3 * n + 2}", x.transpose(1, 2), spec.enc_layers, spec.enc_heads, taps, cz).transpose(1, 2)
    if taps is not None:
        taps["enc.transformer"] = x
    x = _wn_sconv(sd, f"{p}.{3 * n + 5}", F.elu(x), stride=2, causal=cz)
    return x


def semantic_encoder(sd: SD, feat: Tensor, spec: HCodecSpec = SPEC_10) -> Tensor:
    """semantic_module.Encoder.forward, semantic_module.py:157-201 (ResidualUnit :55-81, EncoderBlock
    :122-154).  Zero "same" padding (k-1)//2; strided conv uses k = 2*stride (k=3 when stride==1)."""
    p = "semantic_encoder"
    x = F.conv1d(feat, sd[p + ".conv.conv.weight"], None, padding=1)
    for i, s in enumerate(spec.sem_strides):
        bp = f"{p}.conv_blocks.{i}"
        for u in range(2):
            up = f"{bp}.res_units.{u}"
            y = F.conv1d(F.elu(x), sd[up + ".conv1.conv.weight"], None, padding=1)
            y = F.conv1d(F.elu(y), sd[up + ".conv2.weight"], None)
            x = x + y
        w = sd[bp + ".conv.conv.weight"]
        x = F.conv1d(x, w, sd[bp + ".conv.conv.bias"], stride=s, padding=(w.shape[-1] - 1) // 2)
    return F.conv1d(x, sd[p + ".conv2.conv.weight"], None, padding=1)


def rvq_codebooks(sd: SD, p: str, nq: int) -> Tensor:
    return torch.stack([sd[f"{p}.layers.{q}._codebook.embed"][0] for q in range(nq)])  # [Q,K,D]


def rvq_search(x: Tensor, codebooks: Tensor) -> Tuple[Tensor, Tensor]:
    """Residual nearest-codebook search, vq/core_vq.py:223-231 (distance) and :394-404 (residual loop);
    call sites codec.py:171-172.  x [..., D] -> (indices [..., Q] int64, quantized [..., D])."""
    r = x
    out = torch.zeros_like(x)
    idx = []
    for e in codebooks:
        flat = r.reshape(-1, r.shape[-1])
        dist = -(flat.pow(2).sum(1, keepdim=True) - 2 * flat @ e.t() + e.pow(2).sum(1)[None, :])
        i = dist.max(dim=-1).indices.view(r.shape[:-1])
        q = e[i]
        r = r - q
        out = out + q
        idx.append(i)
    return torch.stack(idx, dim=-1), out


def rvq_search_upstream_association(x: Tensor, codebooks: Tensor) -> Tuple[Tensor, Tensor]:
    """The same residual search with the distance formed the way the THIRD-PARTY package forms it [upstream-memory:
    vector_quantize_pytorch 1.22.15, `cdist` + `EuclideanCodebook.forward`; the package is pinned in requirements.txt:54 and absent here]:
        cdist = ((|x|^2 + |e|^2) + (-2) * x.e).clamp(min=0).sqrt();  index = argmax(-cdist)
    against the in-tree statement rvq_search follows (core_vq.py:223-231: `-(|x|^2 - 2 x.e + |e|^2)`, argmax).  Mathematically the same
    arg-min; in fp32 the two associations round differently and the square root can merge neighbouring distances, so decisions at
    near-ties can differ.  tests/test_rvq_pin_cpu.py counts those and shows each one lies inside tests/util.CODE_TIE_TOL."""
    r = x
    out = torch.zeros_like(x)
    idx = []
    for e in codebooks:
        flat = r.reshape(-1, r.shape[-1])
        x2 = flat.pow(2).sum(-1)
        y2 = e.pow(2).sum(-1)
        xy = (flat @ e.t()) * -2
        dist = -((x2[:, None] + y2[None, :]) + xy).clamp(min=0).sqrt()
        i = dist.argmax(dim=-1).view(r.shape[:-1])
        q = e[i]
        r = r - q
        out = out + q
        idx.append(i)
    return torch.stack(idx, dim=-1), out


def rvq_lookup(indices: Tensor, codebooks: Tensor) -> Tensor:
    """get_output_from_indices (codec.py:183-184; core_vq.py:406-412): sum_q E_q[idx_q].  idx == -1 = a dropped code: upstream
    vector_quantize_pytorch masks it to a zero vector (get_codes_from_indices: `mask = indices == -1` ... `masked_fill(mask, 0.)`); the
    in-tree core_vq.py has no such case (it would index from the end) - the third-party call site is what the product mirrors."""
    out = 0
    for q, e in enumerate(codebooks):
        idx = indices[..., q]
        dropped = idx == -1
        out = out + e[idx.masked_fill(dropped, 0)].masked_fill(dropped[..., None], 0.0)
    return out


def encode(sd: SD, wav: Tensor, feat: Tensor, spec: HCodecSpec = SPEC_10, taps=None) -> Tuple[Tensor, Tensor]:
    """Codec.encode, codec.py:166-175.  wav [B,1,T], feat [B,sem_in,N50] -> two int64 [B,nq,N25]."""
    emb = seanet_encoder(sd, wav, spec, taps)
    sem = semantic_encoder(sd, feat, spec)
    if taps is not None:
        taps["enc.emb"] = emb
        taps["enc.sem"] = sem
    ac, _ = rvq_search(emb.transpose(1, 2), rvq_codebooks(sd, "quantizer", spec.num_quantizers))
    sc, _ = rvq_search(sem.transpose(1, 2), rvq_codebooks(sd, "semantic_quantizer", spec.num_quantizers))
    return ac.transpose(1, 2).contiguous(), sc.transpose(1, 2).contiguous()


# ----------------------------------------------------------------------------- decode side

def _zero_pad(x: Tensor, k: int, causal: bool) -> Tensor:
    """the ConstantPad1d of vq/conv.py:44-47 (Conv1d, stride 1) and :76-79 (ConvTranspose1d): (k - 1, 0) if causal else (k//2, k//2)"""
    return F.pad(x, (k - 1, 0) if causal else (k // 2, k // 2))


def subpixel_upsample(sd: SD, p: str, x: Tensor, stride: int = 2, causal: bool = False) -> Tensor:
    """vq/conv.py:58-91 ("ConvTranspose1d" = 1x1 conv -> pixel shuffle -> zero-pad -> depthwise k5)."""
    x = F.conv1d(x, sd[p + ".up.weight"], sd[p + ".up.bias"])
    b, _, t = x.shape
    d = x.shape[1] // stride
    x = x.unflatten(1, (stride, d)).permute(0, 2, 3, 1).flatten(-2, -1)  # sample t*stride+j <- block j
    w = sd[p + ".dw.weight"]
    return F.conv1d(_zero_pad(x, w.shape[-1], causal), w, sd[p + ".dw.bias"], groups=d)


def _swish(x: Tensor) -> Tensor:
    return x * torch.sigmoid(x)


def resnet_block(sd: SD, p: str, x: Tensor, groups: int, causal: bool = False) -> Tensor:
    """vq/conv.py:263-304 (eval: dropout off): x + k3(swish(GN(k3(swish(GN(x)))))).  The GroupNorm is over the whole clip in
    the causal variant too (the flag only moves the conv padding)."""
    h = F.group_norm(x, groups, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps=1e-6)
    h = F.conv1d(_zero_pad(_swish(h), 3, causal), sd[p + ".conv1.conv.weight"], sd[p + ".conv1.conv.bias"])
    h = F.group_norm(h, groups, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps=1e-6)
    h = F.conv1d(_zero_pad(_swish(h), 3, causal), sd[p + ".conv2.conv.weight"], sd[p + ".conv2.conv.bias"])
    return x + h


def convnext_block(sd: SD, p: str, x: Tensor, causal: bool = False) -> Tensor:
    """vq/conv.py:168-211: dw k7 -> LN -> Linear -> GELU(erf) -> Linear -> *gamma -> +res."""
    c = x.shape[1]
    h = F.conv1d(_zero_pad(x, 7, causal), sd[p + ".dwconv.conv.weight"], sd[p + ".dwconv.conv.bias"], groups=c)
    h = F.layer_norm(h.transpose(1, 2), (c,), sd[p + ".norm.weight"], sd[p + ".norm.bias"], eps=1e-6)
    h = F.linear(h, sd[p + ".pwconv1.linear.weight"], sd[p + ".pwconv1.linear.bias"])
    h = F.linear(F.gelu(h), sd[p + ".pwconv2.linear.weight"], sd[p + ".pwconv2.linear.bias"])
    h = sd[p + ".gamma"] * h
    return x + h.transpose(1, 2)


def istft_head(sd: SD, x: Tensor, n_fft: int, hop: int, taps=None) -> Tensor:
    """ISTFTHead.forward (vq/heads.py:137-150) + ISTFT.forward "same" (vq/spectral_ops.py:48-73)."""
    y = F.linear(x, sd["decoder.head.out.weight"], sd["decoder.head.out.bias"]).transpose(1, 2)
    mag, ph = y.chunk(2, dim=1)
    mag = torch.clip(torch.exp(mag), max=1e2)
    spec = mag * (torch.cos(ph) + 1j * torch.sin(ph))
    if taps is not None:
        taps["dec.spec_re"], taps["dec.spec_im"] = spec.real, spec.imag
    window = torch.hann_window(n_fft)
    pad = (n_fft - hop) // 2
    _, _, t = spec.shape
    frames = torch.fft.irfft(spec, n_fft, dim=1, norm="backward") * window[None, :, None]
    out_size = (t - 1) * hop + n_fft
    y = F.fold(frames, output_size=(1, out_size), kernel_size=(1, n_fft), stride=(1, hop))[:, 0, 0, pad:-pad]
    env = F.fold(window.square().expand(1, t, -1).transpose(1, 2), output_size=(1, out_size),
                 kernel_size=(1, n_fft), stride=(1, hop)).squeeze()[pad:-pad]
    assert (env > 1e-11).all()
    return y / env


def codec_decoder(sd: SD, x: Tensor, spec: HCodecSpec = SPEC_10, taps=None) -> Tensor:
    """CodecDecoder.forward, vq/codec_decoder.py:58-67."""
    p = "decoder"
    cz = spec.causal
    x = subpixel_upsample(sd, p + ".embed", x, causal=cz)
    if taps is not None:
        taps["dec.embed"] = x
    x = resnet_block(sd, p + ".prior_net.0", x, spec.gn_groups, cz)
    x = resnet_block(sd, p + ".prior_net.1", x, spec.gn_groups, cz)
    if taps is not None:
        taps["dec.prior_res1"] = x
    x = transformer(sd, p + ".prior_net.3", x.transpose(1, 2), spec.dec_layers, spec.dec_heads, taps, cz).transpose(1, 2)
    if taps is not None:
        taps["dec.transformer"] = x
    x = resnet_block(sd, p + ".prior_net.5", x, spec.gn_groups, cz)
    x = resnet_block(sd, p + ".prior_net.6", x, spec.gn_groups, cz)
    x = F.group_norm(x, spec.gn_groups, sd[p + ".prior_net.7.weight"], sd[p + ".prior_net.7.bias"], eps=1e-6)
    c = x.shape[1]
    x = F.layer_norm(x.transpose(1, 2), (c,), sd[p + ".norm.weight"], sd[p + ".norm.bias"], eps=1e-6).transpose(1, 2)
    if taps is not None:
        taps["dec.prior"] = x
    for i in range(spec.convnext_layers):
        x = convnext_block(sd, f"{p}.post_net.{i}", x, cz)
    x = F.layer_norm(x.transpose(1, 2), (c,), sd[p + ".final_layer_norm.weight"],
                     sd[p + ".final_layer_norm.bias"], eps=1e-6)
    if taps is not None:
        taps["dec.backbone"] = x
    return istft_head(sd, x, spec.n_fft, spec.hop, taps)


def decode(sd: SD, acoustic_codes: Tensor, semantic_codes: Tensor, spec: HCodecSpec = SPEC_10, taps=None) -> Tensor:
    """Codec.decode, codec.py:178-187.  codes int64 [B,nq,N25] -> wav [B, N25*2*hop]."""
    a = rvq_lookup(acoustic_codes.transpose(1, 2), rvq_codebooks(sd, "quantizer", spec.num_quantizers))
    s = rvq_lookup(semantic_codes.transpose(1, 2), rvq_codebooks(sd, "semantic_quantizer", spec.num_quantizers))
    x = torch.cat([a.transpose(1, 2), s.transpose(1, 2)], dim=1)
    return codec_decoder(sd, x, spec, taps)


# ----------------------------------------------------------------------------- facade helpers

def pad_wav(wav: Tensor, hop: int = 640) -> Tensor:
    """HCodecTokenizer.pad_wav, audio_tokenizer.py:50-53: right zero-pad to a multiple of hop."""
    t = wav.shape[-1]
    return F.pad(wav, (0, math.ceil(t / hop) * hop - t))
