"""Python mirror of the mimi StreamingTransformer with its causal / streaming behaviour, backed by libquarkaudio_hip.so.

    StreamingTransformer.forward / .streaming(B) / .streaming_forever(B) / .reset_streaming()
        <->  QuarkAudio-HCodec/HCodec-1.5/adaptive/model_blocks/mimi/transformer.py:605-698, module/streaming.py:87-123

Only the configuration H-Codec 1.5 instantiates (transformer.py:722-736) is offered: positional_embedding='rope', norm='layer_norm',
gating='none', layer_scale set, no biases; anything else raises.  Inside H-Codec 1.5 itself the same stacks run through `Codec`
(`HCodecSpec.agg_causal / agg_context / bt_causal / bt_context`); this class exposes one stack on its own, which is what the
streaming API needs.
"""
from __future__ import annotations

import ctypes as C
from contextlib import contextmanager
from typing import Dict, Optional

import torch

from . import _lib


class StreamingTransformer(torch.nn.Module):
    def __init__(self, d_model: int, num_heads: int, num_layers: int, dim_feedforward: int = 2048, causal: bool = False,
                 context: Optional[int] = None, positional_embedding: str = "rope", max_period: float = 10_000,
                 layer_scale: Optional[float] = 0.01, gating: str = "none", norm: str = "layer_norm", device="cuda:0",
                 prefix: str = "", **kwargs):
        super().__init__()
        if positional_embedding != "rope" or gating != "none" or norm != "layer_norm" or layer_scale is None or max_period != 10_000:
            raise _lib.QuarkAudioError(-3, "StreamingTransformer: only the H-Codec 1.5 configuration (rope, layer_norm, gating none, "
                                       "LayerScale, max_period 10000) is implemented")
        if kwargs.get("weights_per_step") or kwargs.get("skip_self_attn"):
            raise _lib.QuarkAudioError(-3, "StreamingTransformer: weights_per_step / skip_self_attn are not implemented")
        self.d_model, self.num_heads, self.num_layers, self.dim_feedforward = d_model, num_heads, num_layers, dim_feedforward
        self.causal, self.context = bool(causal), context
        self.device = torch.device(device)
        self._prefix = prefix
        self._lib = _lib.load_library()
        self._handle: Optional[C.c_void_p] = None
        self._sd: Optional[Dict[str, torch.Tensor]] = None
        self._streaming_batch = 0

    # ---- weights
    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        _lib.require_device()
        self._free()
        p = self._prefix + "." if self._prefix else ""
        sd = {"t." + k[len(p):]: v for k, v in state_dict.items() if k.startswith(p)}
        table, n, keep = _lib.tensor_table(sd)
        spec = _lib.qa_mimi_spec(self.d_model, self.num_heads, self.num_layers, self.dim_feedforward, int(self.causal),
                                 int(self.context or 0))
        handle = C.c_void_p()
        _lib.check(self._lib.qa_mimi_create(C.byref(handle), C.byref(spec), table, n, b"t", self.device.index or 0))
        self._handle = handle
        self._sd = {k[2:]: v for k, v in sd.items()}
        return self

    def state_dict(self, *args, destination=None, prefix: str = "", keep_vars: bool = False):
        out = destination if destination is not None else {}
        for k, v in (self._sd or {}).items():
            out[prefix + k] = v
        return out

    def _free(self):
        if getattr(self, "_handle", None):
            self._lib.qa_mimi_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    def _require(self):
        if not self._handle:
            raise _lib.QuarkAudioError(-1, "StreamingTransformer: no weights loaded (call load_state_dict first)")

    # ---- streaming API (module/streaming.py:87-123)
    @property
    def is_streaming(self) -> bool:
        return self._streaming_batch > 0

    def streaming_forever(self, batch_size: int):
        self._require()
        _lib.check(self._lib.qa_mimi_stream_begin(self._handle, int(batch_size)))
        self._streaming_batch = int(batch_size)

    def _stop_streaming(self):
        if self._handle:
            _lib.check(self._lib.qa_mimi_stream_end(self._handle))
        self._streaming_batch = 0

    @contextmanager
    def streaming(self, batch_size: int):
        self.streaming_forever(batch_size)
        try:
            yield
        finally:
            self._stop_streaming()

    def reset_streaming(self):
        self._require()
        _lib.check(self._lib.qa_mimi_stream_reset(self._handle))

    @property
    def streaming_offset(self) -> int:
        return int(self._lib.qa_mimi_stream_offset(self._handle)) if self._handle else -1

    # ---- forward
    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._require()
        if x.dim() != 3 or x.shape[-1] != self.d_model:
            raise _lib.QuarkAudioError(-1, f"StreamingTransformer expects [B, T, {self.d_model}], got {tuple(x.shape)}")
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        y = torch.empty_like(x)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        B, T, _ = x.shape
        with torch.cuda.device(self.device):
            if self.is_streaming:
                if B != self._streaming_batch:
                    raise _lib.QuarkAudioError(-1, f"streaming state was created for batch {self._streaming_batch}, got {B}")
                _lib.check(self._lib.qa_mimi_stream_step(self._handle, x.data_ptr(), T, y.data_ptr(), stream))
            else:
                _lib.check(self._lib.qa_mimi_forward(self._handle, x.data_ptr(), B, T, y.data_ptr(), stream))
        return y
