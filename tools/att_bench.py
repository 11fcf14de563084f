#!/usr/bin/env python
"""attention_kernel alone (qa_debug_attention) on the shapes the codecs launch: microseconds per launch."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unified_audio_amd import _lib  # noqa: E402

lib = _lib.load_library()
fn = lib.qa_debug_attention
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int,
               C.c_longlong, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
tot = 0.0
for name, B, N, H, hd, w in (("agg 32x283", 32, 283, 8, 64, 64), ("enc 32x500", 32, 500, 8, 64, 2), ("bt 32x250 hd128", 32, 250, 8, 128, 32),
                             ("dec 32x500 hd128", 32, 500, 8, 128, 2), ("h20 16x1500", 16, 1500, 24, 64, 0), ("wavlm 16x250 hd64 12h", 16, 250, 12, 64, 0),
                             ("dec10 32x500 hd96", 32, 500, 8, 96, 0)):
    d = H * hd
    qkv = torch.randn(B, N, 3 * d, generator=torch.Generator().manual_seed(5)).to(dev)
    out = torch.empty(B, N, d, device=dev)
    args = (qkv.data_ptr(), 3 * d, qkv.data_ptr() + 4 * d, qkv.data_ptr() + 8 * d, 3 * d, out.data_ptr(), d, B, N, N, N * 3 * d, H, hd, hd ** -0.5, 0, None)
    for _ in range(3):
        fn(*args)
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(20):
        fn(*args)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    tot += us * w
    import hashlib

    digest = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12]  # A/B builds (tools/variants.py) must agree bit for bit
    print(f"{name:24s} {us:8.1f} us  {4.0 * B * H * N * N * hd / us / 1e6:6.1f} TFLOP/s  digest={digest}")
print(f"weighted per H-Codec 1.5 step: {tot / 1e3:.2f} ms")
