"""TEST INFRASTRUCTURE ONLY - CPU restatement of the H-Codec 1.5 deltas on top of oracle/hcodec_ref.py
(adaptive frame rate: similarity grouping, two QueryTokenAggregators on encode, length-injected codes, index
de-aggregation and a bottleneck transformer on decode).  Paths relative to /root/reference/QuarkAudio-HCodec/HCodec-1.5/.

    Codec.encode / decode                          vq/codec_adaptive.py:150-199
    similarity alignment                           adaptive/modeling_flexicodec_new.py:828-921
    index de-aggregation from token lengths        adaptive/modeling_flexicodec_new.py:1007-1041
    QueryTokenAggregator / ProjectedTransformer    adaptive/model_blocks/mimi/transformer.py:701-880
    StreamingTransformer(Layer) / attention        adaptive/model_blocks/mimi/transformer.py:294-698
    interleaved-pair RoPE                          adaptive/model_blocks/mimi/module/rope.py:13-69

Pinned against the reference's own modules by tests/test_oracle_cpu.py (where /root/reference exists) and by the
committed golden vectors tests/golden/hcodec15_*.npz.  RVQ stage: third-party, pinned to vq/core_vq.py (see hcodec_ref.py).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

from . import hcodec_ref as R
from .hcodec_ref import HCodecSpec, SPEC_15

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ------------------------------------------------------------------------------- mimi transformer

def rope_interleaved(q: Tensor, k: Tensor, max_period: float = 10000.0, offset: int = 0) -> Tuple[Tensor, Tensor]:
    """apply_rope, module/rope.py:13-69 ([B,H,T,D] layout): pairs (2i, 2i+1) rotated by (offset + t) * P^(-2i/D)."""
    b, h, t, d = q.shape
    ds = torch.arange(d // 2, dtype=torch.float32)
    freqs = torch.exp(ds * (-math.log(max_period) * 2 / d))
    ts = (float(offset) + torch.arange(t, dtype=torch.float32)).view(1, -1, 1)
    rotr, roti = torch.cos(freqs * ts), torch.sin(freqs * ts)

    def rot(x):
        x = x.reshape(b, h, t, d // 2, 2)
        xr, xi = x[..., 0], x[..., 1]
        return torch.stack([xr * rotr - xi * roti, xr * roti + xi * rotr], dim=-1).reshape(b, h, t, d)

    return rot(q), rot(k)


class MimiStreamState:
    """The streaming state of a StreamingTransformer (`with model.streaming(B)`): one RingKVCache of capacity `context` per layer
    (mimi/transformer.py:212-281,345-370) and the running offset (:284-293,624-626)."""

    def __init__(self, batch: int, n_layers: int, n_heads: int, head_dim: int, capacity: int):
        self.capacity = capacity
        self.cache = [torch.zeros(2, batch, n_heads, capacity, head_dim) for _ in range(n_layers)]
        self.offset = 0

    def reset(self):  # reset_streaming(): the caches are NOT cleared, `end_offset` alone invalidates them (:239-241)
        self.offset = 0

    def complete(self, layer: int, k: Tensor, v: Tensor):
        """RingKVCache.complete (:243-281): write the chunk, return (keys, values, positions [capacity], -1 = never written)."""
        t = k.shape[2]
        assert t <= self.capacity
        idx = (torch.arange(t) + self.offset) % self.capacity
        self.cache[layer][0].index_copy_(2, idx, k)
        self.cache[layer][1].index_copy_(2, idx, v)
        end = self.offset + t
        slots = torch.arange(self.capacity)
        delta = slots - end % self.capacity
        pos = torch.where(delta <= 0, end + delta, end + delta - self.capacity)
        pos = torch.where(slots >= end, torch.full_like(pos, -1), pos)
        return self.cache[layer][0], self.cache[layer][1], pos


def mimi_transformer(sd: SD, p: str, x: Tensor, n_layers: int, n_heads: int, causal: bool = False, context: int = 0,
                     state: "MimiStreamState" = None) -> Tensor:
    """StreamingTransformer.forward, positional_embedding='rope', norm='layer_norm' (eps 1e-5), gating='none' (GELU FFN, no
    biases), LayerScale.  x [B, T, C].  Non-causal: attn_bias None, `context` ignored (transformer.py:403-413).  causal: key j is
    visible to query i iff 0 <= i - j < context (context 0 / None = unbounded).  state: the streaming step (`with
    model.streaming(B)`): RoPE at the running offset, keys / values through the ring caches, offset advanced by T."""
    b, t, c = x.shape
    hd = c // n_heads
    off = state.offset if state is not None else 0
    assert state is None or causal, "Streaming only available for causal"  # transformer.py:382
    for i in range(n_layers):
        lp = f"{p}.layers.{i}"
        y = F.layer_norm(x, (c,), sd[lp + ".norm1.weight"], sd[lp + ".norm1.bias"], eps=1e-5)
        proj = F.linear(y, sd[lp + ".self_attn.in_proj_weight"])  # "b t (p h d) -> p b h t d"
        q, k, v = proj.view(b, t, 3, n_heads, hd).permute(2, 0, 3, 1, 4)
        q, k = rope_interleaved(q, k, offset=off)
        bias = None
        if state is not None:
            k, v, pos_k = state.complete(i, k, v)
        else:
            pos_k = torch.arange(t)
        if causal:
            delta = (off + torch.arange(t)).view(-1, 1) - pos_k.view(1, -1)
            bias = (pos_k.view(1, -1) >= 0) & (delta >= 0)
            if context:
                bias = bias & (delta < context)
        a = F.scaled_dot_product_attention(q, k, v, bias, dropout_p=0.0)
        a = a.transpose(1, 2).reshape(b, t, c)
        x = x + sd[lp + ".layer_scale_1.scale"] * F.linear(a, sd[lp + ".self_attn.out_proj.weight"])
        y = F.layer_norm(x, (c,), sd[lp + ".norm2.weight"], sd[lp + ".norm2.bias"], eps=1e-5)
        y = F.linear(F.gelu(F.linear(y, sd[lp + ".linear1.weight"])), sd[lp + ".linear2.weight"])
        x = x + sd[lp + ".layer_scale_2.scale"] * y
    if state is not None:
        state.offset += t
    return x


# ------------------------------------------------------------------------------- adaptive frame rate

def similarity_alignment(h: Tensor, threshold: float, max_tokens: int):
    """_perform_similarity_alignment_vectorized with x_lens = T for every item (codec_adaptive.py:156-161).
    h [B, T, D] -> (frame_to_segment [B, T] long, num_segments [B] long, sim [B, T-1])."""
    b, t, _ = h.shape
    if t <= 1:
        return torch.zeros(b, t, dtype=torch.long), torch.ones(b, dtype=torch.long), torch.ones(b, max(t - 1, 0))
    sim = F.cosine_similarity(h[:, :-1], h[:, 1:], dim=2)
    new_group = torch.cat([torch.ones(b, 1, dtype=torch.bool), sim <= threshold], dim=1)
    ar = torch.arange(t).unsqueeze(0)
    last_start = torch.cummax(ar * new_group.long(), dim=1).values
    split = ((ar - last_start) % max_tokens) == 0
    seg = torch.cumsum(split.long(), dim=1) - 1
    return seg, seg[:, -1] + 1, sim


def alignment_matrix(seg: Tensor) -> Tensor:
    b, t = seg.shape
    g = int(seg.max()) + 1
    a = torch.zeros(b, g, t)
    a[torch.arange(b)[:, None], seg, torch.arange(t)[None, :]] = 1.0
    return a


def query_token_aggregator(sd: SD, p: str, feats: Tensor, align: Tensor, nseg: Tensor, spec: HCodecSpec) -> Tensor:
    """QueryTokenAggregator.forward, transformer.py:741-826 (use_mean_pooling_init=True, query embedding added).
    feats [B, D, T], align [B, G, T] -> [B, D, G]."""
    b, d, t = feats.shape
    g = align.shape[1]
    group_mask = torch.arange(g)[None, :] < nseg[:, None]
    last = (align * torch.arange(t)).max(dim=2).values  # [B, G]
    frame_len = last.masked_fill(~group_mask, -1).max(dim=1).values + 1
    frame_mask = torch.arange(t)[None, :] < frame_len[:, None]
    last_cnt = last.clone()
    last_cnt[~group_mask] = t + 1
    n_before = (last_cnt.unsqueeze(2) < torch.arange(t)).sum(dim=1)
    frame_dest = torch.arange(t) + n_before
    query_dest = last + torch.arange(g) + 1
    summed = torch.einsum("bgt,bdt->bgd", align, feats)
    queries = (summed / align.sum(dim=2).clamp(min=1).unsqueeze(-1)).transpose(1, 2)
    queries = queries + sd[p + ".query_embedding"].expand(b, -1, g)
    src = torch.cat([feats, queries], dim=2)
    dest = torch.cat([frame_dest, query_dest], dim=1).to(torch.long)
    src_mask = torch.cat([frame_mask, group_mask], dim=1)
    perm = dest.masked_fill(~src_mask, t + g).argsort(dim=1, stable=True)
    inter = torch.gather(src, 2, perm.unsqueeze(1).expand(-1, d, -1))
    out = mimi_transformer(sd, p + ".transformer.transformer", inter.transpose(1, 2), spec.agg_layers, spec.agg_heads,
                           spec.agg_causal, spec.agg_context)
    out = out.transpose(1, 2)
    qpos = perm.argsort(dim=1, stable=True)[:, t:]
    agg = torch.gather(out, 2, qpos.unsqueeze(1).expand(-1, d, -1))
    return agg.masked_fill(~group_mask.unsqueeze(1), 0.0)


def encode(sd: SD, wav: Tensor, feat: Tensor, spec: HCodecSpec = SPEC_15, taps=None) -> Dict[str, Tensor]:
    """Codec.encode, codec_adaptive.py:150-178.  Returns {'acoustic_codes','semantic_codes'}: int64 [B, nq, G] with the
    group length injected: code' = (len - 1) * codebook_size + code (codec_adaptive.py:68-73)."""
    emb = R.seanet_encoder(sd, wav, spec, taps)
    sem = R.semantic_encoder(sd, feat, spec)
    seg, nseg, sim = similarity_alignment(sem.transpose(1, 2), spec.threshold, spec.max_tokens_per_group)
    align = alignment_matrix(seg)
    sem_a = query_token_aggregator(sd, "semantic_aggregator", sem, align, nseg, spec)
    emb_a = query_token_aggregator(sd, "acoustic_aggregator", emb, align, nseg, spec)
    if taps is not None:
        taps.update({"enc.emb": emb, "enc.sem": sem, "enc.sim": sim, "enc.seg": seg, "enc.emb_agg": emb_a, "enc.sem_agg": sem_a})
    ac, _ = R.rvq_search(emb_a.transpose(1, 2), R.rvq_codebooks(sd, "quantizer", spec.num_quantizers))
    sc, _ = R.rvq_search(sem_a.transpose(1, 2), R.rvq_codebooks(sd, "semantic_quantizer", spec.num_quantizers))
    tl = align.sum(dim=2).long().unsqueeze(1)  # [B,1,G]
    inject = lambda c: (tl - 1) * spec.codebook_size + c.transpose(1, 2)  # noqa: E731
    return {"acoustic_codes": inject(ac), "semantic_codes": inject(sc)}


def extract_lengths(codes: Tensor, codebook_size: int):
    """codec_adaptive.py:75-80."""
    length_id = torch.div(codes, codebook_size, rounding_mode="floor") + 1
    return codes % codebook_size, length_id[:, 0, :]


def deaggregate_indices(codes: Tensor, lengths: Tensor) -> Tensor:
    """_deaggregate_features_from_token_lengths applied to index tensors [B, nq, G] (codec_adaptive.py:188-189):
    per-item repeat_interleave, pad_sequence with 0."""
    items = [torch.repeat_interleave(codes[i].transpose(0, 1), lengths[i], dim=0) for i in range(codes.shape[0])]
    return torch.nn.utils.rnn.pad_sequence(items, batch_first=True, padding_value=0).transpose(1, 2)


def decode(sd: SD, acoustic_codes: Tensor, semantic_codes: Tensor, spec: HCodecSpec = SPEC_15, taps=None) -> Tensor:
    """Codec.decode with token_lengths=None, codec_adaptive.py:181-199."""
    ac, tl = extract_lengths(acoustic_codes, spec.codebook_size)
    sc, tl = extract_lengths(semantic_codes, spec.codebook_size)
    ac, sc = deaggregate_indices(ac, tl), deaggregate_indices(sc, tl)
    qa = R.rvq_lookup(ac.transpose(1, 2), R.rvq_codebooks(sd, "quantizer", spec.num_quantizers))
    qs = R.rvq_lookup(sc.transpose(1, 2), R.rvq_codebooks(sd, "semantic_quantizer", spec.num_quantizers))
    cat = torch.cat([qa, qs], dim=2)  # [B, T, 2*code_dim]
    bt = mimi_transformer(sd, "bottleneck_transformer.transformer", cat, spec.bt_layers, spec.bt_heads, spec.bt_causal, spec.bt_context)
    if taps is not None:
        taps["dec.bottleneck"] = bt
    return R.codec_decoder(sd, bt.transpose(1, 2), spec, taps)
