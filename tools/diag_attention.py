#!/usr/bin/env python
"""Diagnostic: run-to-run determinism of attention_kernel alone over a sweep of shapes and debug bits."""
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


def run(B, N, H, hd, reps=4, dbg=0, scale_in=1.0):
    d = H * hd
    g = torch.Generator(device="cpu").manual_seed(B * 7 + N)
    qkv = (torch.randn(B, N, 3 * d, generator=g) * scale_in).to(dev)
    _lib.set_knob("QA_ATT_DEBUG", dbg)
    outs = []
    for _ in range(reps):
        out = torch.full((B, N, d), float("nan"), device=dev)
        st = fn(qkv.data_ptr(), 3 * d, qkv.data_ptr() + 4 * d, qkv.data_ptr() + 8 * d, 3 * d, out.data_ptr(), d, B, N, N, N * 3 * d, H, hd,
                hd ** -0.5, 0, None)
        assert st == 0, lib.qa_last_error()
        torch.cuda.synchronize()
        outs.append(out)
    diff = torch.stack([(outs[0] - o).abs() for o in outs[1:]]).amax(dim=0)
    blocks = (diff.view(B, N, H, hd).amax(dim=3) > 0)
    nb = int(blocks.any(dim=1).sum())  # (clip, head) pairs with any difference
    print(f"B={B:3d} N={N:5d} H={H:2d} hd={hd:3d} dbg={dbg:2d} scale={scale_in}: max diff {float(diff.max()):.3e}, differing elements {int((diff > 0).sum())}, "
          f"(clip, head) pairs touched {nb} of {B * H}, finite {bool(torch.isfinite(outs[0]).all())}", flush=True)


for shape in ((16, 1500, 24, 64), (16, 1500, 8, 64), (4, 1500, 24, 64), (1, 1500, 24, 64), (16, 500, 24, 64), (64, 500, 24, 64), (16, 1504, 24, 64),
              (16, 1536, 24, 64), (32, 283, 8, 64), (32, 500, 8, 128), (16, 1500, 12, 128), (16, 1500, 16, 96)):
    run(*shape)
for dbg in (1, 4, 8, 16, 29):
    run(16, 1500, 24, 64, dbg=dbg)
run(16, 1500, 24, 64, scale_in=0.1)
run(16, 1500, 24, 64, scale_in=3.0)
