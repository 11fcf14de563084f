"""TEST INFRASTRUCTURE ONLY - run the reference's OWN `LLM_SFT` (QuarkAudio-UniSE/model/llm/llm_sft.py, llm.py) in this
container so that oracle/llm_ref.py's restatement of `generate` / `sample_logits` can be pinned to it and token-stream
golden vectors generated (oracle/gen_golden_lm.py).  The GPU box has no /root/reference: only the goldens travel.

The reference file targets transformers 4.49 / 4.57; the container has 5.x.  Three small compatibility
patches, none of which touches arithmetic:
  * `LlamaModel._update_causal_mask` (llm.py:77) no longer exists.  In 4.x, with SDPA attention, no attention_mask and
    no output_attentions it returned None (the SDPA kernel's own `is_causal` path); the patch returns None.
  * the decoder layers are called with the removed kwarg `past_key_value=` (llm.py:203) and their result is indexed
    `[0]` (llm.py:211): each layer instance of the constructed model gets a forward that renames the kwarg to
    `past_key_values` and wraps the returned tensor in a 1-tuple.
  * `x_transformers` (conformer.py:17; the condition encoder is constructed, never called on this path) is stubbed.
The package is imported as `model.llm` without executing model/__init__.py (which needs Lightning).
"""
from __future__ import annotations

import importlib
import os
import sys
import types
import warnings

from oracle.ref_shim import REFERENCE_ROOT

_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stubs")
_UNISE = os.path.join(REFERENCE_ROOT, "QuarkAudio-UniSE")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(_UNISE, "model", "llm", "llm_sft.py"))


def _import_llm_sft():
    import transformers

    if not hasattr(transformers.LlamaModel, "_update_causal_mask"):
        transformers.LlamaModel._update_causal_mask = lambda self, *a, **k: None
    if "model" not in sys.modules or not getattr(sys.modules["model"], "_qa_shim", False):
        pkg = types.ModuleType("model")
        pkg.__path__ = [os.path.join(_UNISE, "model")]
        pkg._qa_shim = True
        sys.modules["model"] = pkg
    sys.path.insert(0, _STUBS)
    try:
        return importlib.import_module("model.llm.llm_sft")
    finally:
        sys.path.remove(_STUBS)


def _patch_layer(layer):
    inner = layer.forward

    def forward(hidden_states, *args, past_key_value=None, **kwargs):
        if past_key_value is not None:
            kwargs["past_key_values"] = past_key_value
        out = inner(hidden_states, *args, **kwargs)
        return out if isinstance(out, tuple) else (out,)

    layer.forward = forward


def load_reference_llm(spec, task_map=None):
    """Construct the reference LLM_SFT (eval, random init) for an oracle LMSpec (QuarkAudio-UniSE/conf/config.yaml:131-146)."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found under {REFERENCE_ROOT}")
    mod = _import_llm_sft()
    base = dict(cond_dim=80, global_size=spec.global_size, semantic_size=spec.semantic_size, hidden_size=spec.hidden,
                num_layers=spec.n_layers, num_attention_heads=spec.n_heads, dropout_p=0.1, max_position_embeddings=4096,
                label_smoothing=0.1,
                conformer_params=dict(num_layers=1, dim=32, heads=2, dim_head=16, depthwise_conv_kernel_size=31, ff_mult=2,
                                      dropout=0.1, qk_norm=None, pe_attn_head=None))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = mod.LLM_SFT(num_tasks=spec.num_tasks, task_map=task_map or {"se": 0, "tse": 1, "rtse": 2},
                            feats_dim=spec.feats_dim, llm_base_config=base)
    for layer in model.layers:
        _patch_layer(layer)
    return model.eval()


def load_state(model, sd):
    """Load an oracle/synth state_dict; the only keys left at their random init are the never-called condition encoder
    (cond_input_layer / cond_encoder / cond_output_layer) and HF's rotary buffer."""
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    bad = [k for k in missing if not (k.startswith("cond_") or k.startswith("rotary_emb."))]
    assert not bad, bad
    return model
