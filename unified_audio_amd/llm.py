"""Python mirror of the reference's UniSE AR-LM interface, backed by libquarkaudio_hip.so.

    LLM_SFT.generate  <->  QuarkAudio-UniSE/model/llm/llm_sft.py:93-195   (greedy path: model/model.py:173)

Weights come in the reference's key layout (the Lightning checkpoint's `dnn.*` entries, prefix optional).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib

DEFAULT_TASK_MAP = {"se": 0, "tse": 1, "rtse": 2}  # conf/config.yaml:132-136


class LLM_SFT:
    def __init__(self, num_tasks: int = 3, task_map: Optional[Dict[str, int]] = None, feats_dim: int = 768,
                 llm_base_config: Optional[dict] = None, *, device: str | torch.device = "cuda:0"):
        cfg = dict(global_size=4096, semantic_size=8192, hidden_size=512, num_layers=12, num_attention_heads=8)
        cfg.update(llm_base_config or {})
        self.task_map = dict(task_map or DEFAULT_TASK_MAP)
        self.device = torch.device(device)
        s = _lib.qa_lm_spec()
        s.hidden, s.n_layers, s.n_heads = cfg["hidden_size"], cfg["num_layers"], cfg["num_attention_heads"]
        s.intermediate = 4 * cfg["hidden_size"]  # llm.py:69
        s.global_size, s.semantic_size = cfg["global_size"], cfg["semantic_size"]
        s.feats_dim, s.num_tasks = feats_dim, num_tasks
        s.rope_theta, s.rms_eps = 10000.0, 1e-6  # LlamaConfig defaults (llm.py:63-72)
        self._spec = s
        self.global_offset = 3
        self.semantic_offset = 3 + cfg["global_size"]
        self._lib = _lib.load_library()
        self._handle = C.c_void_p()

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        _lib.require_device()
        sd = {(k[4:] if k.startswith("dnn.") else k): v for k, v in state_dict.items()}
        self._free()
        table, n, keep = _lib.tensor_table(sd)
        handle = C.c_void_p()
        _lib.check(self._lib.qa_lm_create(C.byref(handle), C.byref(self._spec), table, n, self.device.index or 0))
        del keep
        self._handle = handle
        return self

    def eval(self):
        return self

    def _free(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.qa_lm_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    @torch.no_grad()
    def generate(self, task_name: str, enroll_mel, enroll_feats, mix_mel: torch.Tensor, mix_feats: torch.Tensor,
                 global_length: int = 32, temperature: float = 0.8, top_k: int = 50, top_p: float = 0.95,
                 do_sample: bool = True):
        """Returns (global_ids [B, global_length], semantic_ids [B, mix_mel.size(1)]) int64, offsets subtracted.
        Only `mix_mel.size(1)` is consumed from the mel inputs, exactly like the reference (llm_sft.py:108).
        do_sample=True (the reference's signature default, llm_sft.py:106) samples every token on the device with
        CustomLlamaModel.sample_logits' filters (llm.py:253-288); the draws come from a Philox stream seeded from torch's
        global generator (torch.manual_seed controls reproducibility, as in the reference), so the distribution - not
        the individual stream - matches the reference."""
        if not self._handle.value:
            raise _lib.QuarkAudioError(-3, "LLM_SFT has no weights: call load_state_dict first")
        task = self.task_map[task_name]  # KeyError like the reference
        mix = mix_feats.to(device=self.device, dtype=torch.float32).contiguous()
        B, n_mix, _ = mix.shape
        enr, n_enr = None, 0
        if enroll_mel is not None:
            enr = enroll_feats.to(device=self.device, dtype=torch.float32).contiguous()
            n_enr = enr.shape[1]
        S = int(mix_mel.size(1))
        gids = torch.empty((B, global_length), dtype=torch.int64, device=self.device)
        sids = torch.empty((B, S), dtype=torch.int64, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        enr_ptr = enr.data_ptr() if enr is not None else None
        if do_sample:
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())  # advances torch's CPU generator: new draws per call
            _lib.check(self._lib.qa_lm_generate_sampled(self._handle, task, enr_ptr, n_enr, mix.data_ptr(), n_mix, B, global_length,
                                                        S, temperature, top_k, top_p, seed, gids.data_ptr(), sids.data_ptr(), stream))
        else:
            _lib.check(self._lib.qa_lm_generate(self._handle, task, enr_ptr, n_enr, mix.data_ptr(), n_mix, B, global_length, S,
                                                temperature, top_k, top_p, gids.data_ptr(), sids.data_ptr(), stream))
        return gids, sids


def sample_logits(logits: torch.Tensor, temperature: float = 0.8, top_k: int = 50, top_p: float = 0.95,
                  do_sample: bool = True, seed: Optional[int] = None) -> torch.Tensor:
    """CustomLlamaModel.sample_logits (llm.py:253-288) on the device: logits [B, V] float32 cuda -> [B, 1] int64.
    Unlike the reference it does not filter `logits` in place."""
    lib = _lib.load_library()
    x = logits.to(dtype=torch.float32).contiguous()
    if not x.is_cuda:
        raise _lib.QuarkAudioError(-1, "sample_logits: logits must live on the HIP device (there is no CPU path)")
    out = torch.empty((x.shape[0],), dtype=torch.int64, device=x.device)
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
    _lib.check(lib.qa_sample_logits(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), top_k, top_p, temperature,
                                    1 if do_sample else 0, seed, out.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream))
    return out[:, None]
