#!/usr/bin/env python
"""What does moving N bytes in and N bytes out cost on this device when the kernel does nothing else?  (`roofline_hbm`.)
The byte-bound kernels of the codec graphs move 37 - 131 MB per launch; their HBM fractions (0.32 - 0.49 of 8 TB/s) are judged against what a plain
device-to-device copy of the same bytes achieves in one launch - the practical roofline of a launch of that size - next to qa_rownorm on the same rows.
usage: copy_floor.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unified_audio_amd as qa  # noqa: E402
from unified_audio_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = qa.load_library()


def timed(fn, reps=200):
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


print("rows x C         MB in+out   copy us   copy TB/s (frac of 8)   qa_rownorm(LN) us   TB/s (frac)")
for rows, C in ((9056, 512), (8000, 1024), (16000, 1024), (16000, 768), (48000, 1024)):
    x = torch.randn(rows, C, device=dev)
    y = torch.empty_like(x)
    w = torch.ones(C, device=dev)
    b = torch.zeros(C, device=dev)
    mb = 2 * x.numel() * 4 / 1e6
    t_c = timed(lambda: y.copy_(x))
    stream = torch.cuda.current_stream().cuda_stream
    t_n = timed(lambda: _lib.check(lib.qa_rownorm(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), rows, C, 1e-5, 2, stream)))
    # MB per us = TB/s
    print(f"{rows:6d} x {C:4d}   {mb:8.1f}   {t_c:7.1f}   {mb / t_c:5.2f} ({mb / t_c / 8:.2f})          {t_n:7.1f}          {mb / t_n:5.2f} ({mb / t_n / 8:.2f})", flush=True)
