"""Pins the LM oracle's Llama body against the container's transformers.LlamaModel (the reference lifts its layers out
of that class: QuarkAudio-UniSE/model/llm/llm.py:63-79) and checks cached-vs-full equality (llm.py:229-250)."""
import pytest
import torch

from oracle import llm_ref as L

SMALL = L.LMSpec(hidden=64, n_layers=2, n_heads=2, global_size=40, semantic_size=50, feats_dim=32, num_tasks=3)


def _hf_model(sd, spec):
    transformers = pytest.importorskip("transformers")
    cfg = transformers.LlamaConfig(vocab_size=spec.vocab, hidden_size=spec.hidden, num_hidden_layers=spec.n_layers,
                                   num_attention_heads=spec.n_heads, intermediate_size=spec.intermediate,
                                   max_position_embeddings=4096)
    m = transformers.LlamaModel(cfg).eval()
    body = {k: v for k, v in sd.items() if k.startswith("layers.") or k == "norm.weight"}
    body["embed_tokens.weight"] = sd["codec_embedding.weight"]
    m.load_state_dict(body, strict=True)
    return m


def test_llama_body_matches_transformers():
    sd = L.lm_state_dict(3, SMALL)
    m = _hf_model(sd, SMALL)
    x = torch.randn(2, 9, SMALL.hidden, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        ref = m(inputs_embeds=x).last_hidden_state
        mine = L.llm_forward(sd, x, L.KVCache(SMALL.n_layers), SMALL)
    assert float((ref - mine).abs().max()) < 2e-5


def test_cached_decode_equals_full_forward():
    sd = L.lm_state_dict(4, SMALL)
    x = torch.randn(2, 7, SMALL.hidden, generator=torch.Generator().manual_seed(1))
    full = L.llm_forward(sd, x, L.KVCache(SMALL.n_layers), SMALL)
    cache = L.KVCache(SMALL.n_layers)
    L.llm_forward(sd, x[:, :4], cache, SMALL)
    steps = [L.llm_forward(sd, x[:, i:i + 1], cache, SMALL) for i in range(4, 7)]
    assert float((torch.cat(steps, 1) - full[:, 4:]).abs().max()) < 2e-5


def test_generate_shapes_ranges_and_forcing():
    sd = L.lm_state_dict(5, SMALL)
    mix = L.synth_feats(1, 2, 6, SMALL.feats_dim)
    enr = L.synth_feats(2, 2, 5, SMALL.feats_dim)
    g, s, toks, gaps = L.generate(sd, "tse", enr, mix, semantic_length=6, global_length=4, spec=SMALL)
    assert g.shape == (2, 4) and s.shape == (2, 6) and toks.shape == (2, 11)
    assert int(g.min()) >= 0 and int(g.max()) < SMALL.global_size and int(s.min()) >= 0 and int(s.max()) < SMALL.semantic_size
    assert (gaps >= 0).all()
    g2, s2, toks2, _ = L.generate(sd, "tse", enr, mix, semantic_length=6, global_length=4, spec=SMALL, forced=toks)
    assert torch.equal(toks, toks2) and torch.equal(g, g2) and torch.equal(s, s2)
    # SE prompt has no enrollment part (llm_sft.py:127-128)
    assert L.build_prompt(sd, 0, None, mix).shape[1] == 2 + 6 and L.build_prompt(sd, 1, enr, mix).shape[1] == 3 + 5 + 6
