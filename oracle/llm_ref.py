"""TEST INFRASTRUCTURE ONLY - CPU restatement (plain PyTorch fp32) of the UniSE AR-LM generate path.

Follows /root/reference/QuarkAudio-UniSE/model/llm/llm_sft.py:93-195 (LLM_SFT.generate: prompt assembly, prefill,
33 global + N semantic steps with vocabulary-range masks) and model/llm/llm.py:253-288 (sample_logits, greedy branch).
The decoder body in the reference is HF transformers' LlamaDecoderLayer / LlamaRMSNorm / LlamaRotaryEmbedding
(third-party, pins 4.49.0 / 4.57.1; llm.py:63-79,182-211); it is restated here from the published Llama arithmetic and
validated against the container's transformers.LlamaModel by tests/test_llm_oracle_cpu.py.  The reference's own
llm.py cannot be constructed under the installed transformers 5.x, and it ships no tests for this path
-> PARITY UNPINNED beyond that cross-check.  Nothing in the product path may import this module.

State-dict keys are the reference's (`LLM_SFT.state_dict()` = the Lightning checkpoint's `dnn.*` with the prefix
stripped): task_embedding.weight, enroll_sos_embedding.weight, mix_sos_embedding.weight, adapter.{weight,bias},
codec_embedding.weight, output_head.weight, layers.{i}.self_attn.{q,k,v,o}_proj.weight,
layers.{i}.mlp.{gate,up,down}_proj.weight, layers.{i}.{input,post_attention}_layernorm.weight, norm.weight.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass(frozen=True)
class LMSpec:
    """QuarkAudio-UniSE/conf/config.yaml:131-146 (llm_config)."""

    hidden: int = 512
    n_layers: int = 12
    n_heads: int = 8
    global_size: int = 4096
    semantic_size: int = 8192
    feats_dim: int = 768
    num_tasks: int = 3
    rope_theta: float = 10000.0
    rms_eps: float = 1e-6

    @property
    def intermediate(self) -> int:
        return 4 * self.hidden  # llm.py:69

    @property
    def vocab(self) -> int:
        return 3 + self.global_size + self.semantic_size  # llm.py:42

    @property
    def global_offset(self) -> int:
        return 3  # llm.py:43

    @property
    def semantic_offset(self) -> int:
        return 3 + self.global_size  # llm.py:44


SPEC_UNISE = LMSpec()
TASK_MAP = {"se": 0, "tse": 1, "rtse": 2}  # config.yaml:132-136


from unified_audio_amd.synth import synth_feats  # noqa: E402,F401  (pure data generation lives on the product side)
from unified_audio_amd import synth as _synth  # noqa: E402


def lm_state_dict(seed: int, spec: LMSpec = SPEC_UNISE) -> Dict[str, Tensor]:
    """Seeded random weights (numpy PCG64) with the reference's key names (generator: unified_audio_amd/synth.py)."""
    return _synth.lm_state_dict(seed, spec)


# ------------------------------------------------------------------------------- Llama body

def _rms(x: Tensor, w: Tensor, eps: float) -> Tensor:
    var = x.float().pow(2).mean(-1, keepdim=True)  # LlamaRMSNorm
    return w * (x * torch.rsqrt(var + eps))


def _rope(n0: int, n: int, hd: int, theta: float) -> Tuple[Tensor, Tensor]:
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
    fr = torch.arange(n0, n0 + n).float()[:, None] * inv[None, :]
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos(), emb.sin()


def _rot(x: Tensor) -> Tensor:
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


class KVCache:
    def __init__(self, n_layers: int):
        self.k: List[Optional[Tensor]] = [None] * n_layers
        self.v: List[Optional[Tensor]] = [None] * n_layers

    def length(self) -> int:
        return 0 if self.k[0] is None else self.k[0].shape[2]

    def append(self, i: int, k: Tensor, v: Tensor):
        self.k[i] = k if self.k[i] is None else torch.cat([self.k[i], k], dim=2)
        self.v[i] = v if self.v[i] is None else torch.cat([self.v[i], v], dim=2)
        return self.k[i], self.v[i]


def llm_forward(sd: Dict[str, Tensor], x: Tensor, cache: KVCache, spec: LMSpec = SPEC_UNISE) -> Tensor:
    """CustomLlamaModel.llm_forward (llm.py:150-227) with use_cache=True: 12 Llama layers + final norm, causal."""
    b, n, d = x.shape
    h, hd = spec.n_heads, spec.hidden // spec.n_heads
    past = cache.length()
    cos, sin = _rope(past, n, hd, spec.rope_theta)
    for i in range(spec.n_layers):
        p = f"layers.{i}"
        y = _rms(x, sd[p + ".input_layernorm.weight"], spec.rms_eps)
        q = F.linear(y, sd[p + ".self_attn.q_proj.weight"]).view(b, n, h, hd).transpose(1, 2)
        k = F.linear(y, sd[p + ".self_attn.k_proj.weight"]).view(b, n, h, hd).transpose(1, 2)
        v = F.linear(y, sd[p + ".self_attn.v_proj.weight"]).view(b, n, h, hd).transpose(1, 2)
        q = q * cos + _rot(q) * sin
        k = k * cos + _rot(k) * sin
        kk, vv = cache.append(i, k, v)
        w = torch.matmul(q, kk.transpose(2, 3)) / math.sqrt(hd)
        if n > 1:  # causal over (past + n) keys
            mask = torch.ones(n, past + n, dtype=torch.bool).tril(diagonal=past)
            w = w.masked_fill(~mask, float("-inf"))
        w = F.softmax(w, dim=-1, dtype=torch.float32)
        o = torch.matmul(w, vv).transpose(1, 2).reshape(b, n, d)
        x = x + F.linear(o, sd[p + ".self_attn.o_proj.weight"])
        y = _rms(x, sd[p + ".post_attention_layernorm.weight"], spec.rms_eps)
        y = F.linear(F.silu(F.linear(y, sd[p + ".mlp.gate_proj.weight"])) * F.linear(y, sd[p + ".mlp.up_proj.weight"]),
                     sd[p + ".mlp.down_proj.weight"])
        x = x + y
    return _rms(x, sd["norm.weight"], spec.rms_eps)


def build_prompt(sd: Dict[str, Tensor], task: int, enroll_feats: Optional[Tensor], mix_feats: Tensor) -> Tensor:
    """llm_sft.py:110-128: [task, (enroll_sos, adapter(enroll)), mix_sos, adapter(mix)] continuous embeddings."""
    b = mix_feats.shape[0]
    parts = [sd["task_embedding.weight"][task].expand(b, 1, -1)]
    if enroll_feats is not None:
        parts += [sd["enroll_sos_embedding.weight"][0].expand(b, 1, -1),
                  F.linear(enroll_feats, sd["adapter.weight"], sd["adapter.bias"])]
    parts += [sd["mix_sos_embedding.weight"][0].expand(b, 1, -1),
              F.linear(mix_feats, sd["adapter.weight"], sd["adapter.bias"])]
    return torch.cat(parts, dim=1)


@torch.no_grad()
def generate(sd: Dict[str, Tensor], task_name: str, enroll_feats: Optional[Tensor], mix_feats: Tensor,
             semantic_length: int, global_length: int = 32, spec: LMSpec = SPEC_UNISE, forced: Optional[Tensor] = None):
    """LLM_SFT.generate with do_sample=False (model.py:173).  Returns (global_ids [B,G], semantic_ids [B,S],
    tokens [B, G+1+S] raw vocabulary ids, gaps [B, G+1+S] = top-1 minus top-2 logit inside the active range).
    `forced` (raw ids [B, G+1+S]) teacher-forces the fed-back tokens so a different implementation's stream can be
    audited step by step."""
    cache = KVCache(spec.n_layers)
    llm_forward(sd, build_prompt(sd, TASK_MAP[task_name], enroll_feats, mix_feats), cache, spec)
    b = mix_feats.shape[0]
    toks, gaps = [], []

    def phase(first_id: int, steps: int, lo: int, hi: int):
        ids = torch.full((b,), first_id, dtype=torch.long)
        for _ in range(steps):
            hs = llm_forward(sd, sd["codec_embedding.weight"][ids][:, None, :], cache, spec)
            logits = F.linear(hs[:, 0], sd["output_head.weight"])[:, lo:hi]  # range mask, llm_sft.py:150-153 / :180-182
            top2 = logits.topk(2, dim=-1).values
            nxt = logits.argmax(dim=-1) + lo  # sample_logits greedy branch (llm.py:286): top-k/top-p cannot move the arg-max
            toks.append(nxt)
            gaps.append(top2[:, 0] - top2[:, 1])
            ids = nxt if forced is None else forced[:, len(toks) - 1]

    phase(0, global_length + 1, spec.global_offset, spec.global_offset + spec.global_size)  # llm_sft.py:137-164
    phase(1, semantic_length, spec.semantic_offset, spec.semantic_offset + spec.semantic_size)  # :166-193
    tokens = torch.stack(toks, dim=1)
    src = tokens if forced is None else forced
    g = src[:, :global_length] - spec.global_offset
    s = src[:, global_length + 1:] - spec.semantic_offset
    return g, s, tokens, torch.stack(gaps, dim=1)
