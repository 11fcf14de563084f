#!/usr/bin/env python
"""A/B of attention_kernel's waves per workgroup (QA_ATT_NW = 4 | 3) on the codec's attention shapes: time per launch (HIP events).
usage: att_nw_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_kernels_gpu import _attention_alone  # noqa: E402
from unified_audio_amd import _lib, load_library  # noqa: E402

lib = load_library()
dev = torch.device("cuda:0")
# (name, B, N, H, hd): H-Codec 1.5 aggregators (250 frames + 33 groups), encoder (500 frames), bottleneck (250), H-Codec 1.0 decoder
SHAPES = [("agg15 N=283", 32, 283, 8, 64), ("agg15 x2 N=283", 64, 283, 8, 64), ("enc N=500", 32, 500, 8, 64), ("N=250", 32, 250, 8, 64),
          ("N=300", 32, 300, 8, 64), ("h20 N=1500 (forced)", 16, 1500, 24, 64)]
for name, B, N, H, hd in SHAPES:
    qkv = torch.randn(B, N, 3 * H * hd, device=dev)
    row = []
    for nw in (4, 3, 0):
        _lib.set_knob("QA_ATT_NW", nw)
        for _ in range(3):
            _attention_alone(lib, qkv, H, hd)
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(20):
            _attention_alone(lib, qkv, H, hd)
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 20 * 1e3)
    _lib.set_knob("QA_ATT_NW", 0)
    print(f"{name:22s} B={B:3d} N={N:5d} H={H:2d} hd={hd}:  NW=4 {row[0]:8.1f} us   NW=3 {row[1]:8.1f} us   rule {row[2]:8.1f} us", flush=True)
