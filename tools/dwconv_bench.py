#!/usr/bin/env python
"""Depthwise conv (+ LayerNorm) alone on the device: the row kernel against the row-strip kernel (QA_DWCONV_STRIP), next to qa_rownorm and a plain copy of
the same bytes.  usage: dwconv_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unified_audio_amd as qa  # noqa: E402
from unified_audio_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = qa.load_library()


def timed(fn, reps=100):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


st = torch.cuda.current_stream().cuda_stream
print("B x T x C  k  LN     MB in+out   row kernel us (TB/s)   strip kernel us (TB/s, frac of 8)   rownorm us   copy us")
for B, T, C, k, ln in ((32, 500, 1024, 7, True), (32, 500, 768, 7, True), (16, 1500, 1024, 7, True), (32, 500, 512, 5, False), (16, 250, 1024, 7, True)):
    x = torch.randn(B, T, C, device=dev)
    y = torch.empty_like(x)
    w = torch.randn(k, C, device=dev)
    b = torch.randn(C, device=dev)
    lw, lb = (torch.ones(C, device=dev), torch.zeros(C, device=dev)) if ln else (None, None)
    mb = 2 * x.numel() * 4 / 1e6

    def run():
        _lib.check(lib.qa_dwconv_cl(x.data_ptr(), w.data_ptr(), b.data_ptr(), lw.data_ptr() if ln else None, lb.data_ptr() if ln else None, y.data_ptr(),
                                    B, T, C, k, -1, 1e-6, st))

    _lib.set_knob("QA_DWCONV_STRIP", 0)
    t0 = timed(run)
    _lib.set_knob("QA_DWCONV_STRIP", 1)
    t1 = timed(run)
    ones = torch.ones(C, device=dev)
    tn = timed(lambda: _lib.check(lib.qa_rownorm(x.data_ptr(), ones.data_ptr(), ones.data_ptr(), y.data_ptr(), B * T, C, 1e-6, 2, st)))
    tc = timed(lambda: y.copy_(x))
    print(f"{B:2d} x {T:4d} x {C:4d}  {k}  {int(ln)}   {mb:8.1f}     {t0:7.1f} ({mb / t0:4.2f})          {t1:7.1f} ({mb / t1:4.2f}, {mb / t1 / 8:.2f})"
          f"              {tn:7.1f}     {tc:7.1f}", flush=True)
