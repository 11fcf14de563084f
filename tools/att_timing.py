#!/usr/bin/env python
"""Per-phase shader cycles of attention_kernel's key-tile loop (needs the -DQA_ATT_TIMING=1 build):
    QA_VARIANT_SOURCES=attention.hip python tools/variants.py att_timing "-DQA_ATT_TIMING=1"
    QA_LIBRARY=tools/_variants/att_timing/libquarkaudio_hip.so python tools/att_timing.py
Phases per 32-key tile and wave: [0] top barrier + LDS stores of the prefetched tile + second barrier (includes the wait for the tile's global
loads), [1] S = K Q^T (8 x (ds_read_b128, 4 MFMAs) at head_dim 64), [2] online softmax (VALU), [3] O += V P (16 x (LDS read, DT MFMAs)).
MFMA cycles of a tile are fixed (64 MFMAs x 64 cycles = 4096 at head_dim 64), so [1] + [3] - 4096 is what the wave waited inside its matrix phases."""
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
tm = lib.qa_debug_att_timing
tm.restype = C.c_int
tm.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
dev = torch.device("cuda:0")
print(f"{'shape':26s} {'us':>8s} {'tiles/wave':>10s} {'cyc/tile':>9s} | barrier+store  S=KQ^T  softmax  O+=VP | MFMA cycles/tile  share of the loop")
for name, B, N, H, hd in (("agg 32x283 hd64", 32, 283, 8, 64), ("enc 32x500 hd64", 32, 500, 8, 64), ("wavlm 16x250 hd64 12h", 16, 250, 12, 64),
                          ("h20 16x1500 hd64 24h", 16, 1500, 24, 64), ("bt 32x250 hd128", 32, 250, 8, 128), ("dec 32x500 hd128", 32, 500, 8, 128)):
    d = H * hd
    qkv = torch.randn(B, N, 3 * d, generator=torch.Generator().manual_seed(5)).to(dev)
    out = torch.empty(B, N, d, device=dev)
    args = (qkv.data_ptr(), 3 * d, qkv.data_ptr() + 4 * d, qkv.data_ptr() + 8 * d, 3 * d, out.data_ptr(), d, B, N, N, N * 3 * d, H, hd, hd ** -0.5, 0, None)
    for _ in range(2):
        fn(*args)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 8)()
    tm(buf, 1)
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    reps = 5
    for _ in range(reps):
        fn(*args)
    e1.record()
    torch.cuda.synchronize()
    tm(buf, 0)
    us = e0.elapsed_time(e1) / reps * 1e3
    ph, total, waves, tiles = [buf[i] for i in range(4)], buf[4], buf[5], max(1, buf[6])
    per = [p / tiles for p in ph]
    mfma = (hd // 8 * 4 + 16 * (hd // 32)) * 64  # S: hd/8 groups x 4, PV: 16 steps x hd/32, 64 cycles each
    loop = sum(per)
    print(f"{name:26s} {us:8.1f} {tiles / max(1, waves):10.1f} {loop:9.0f} | {per[0]:13.0f} {per[1]:7.0f} {per[2]:8.0f} {per[3]:6.0f} | "
          f"{mfma:16d}  {mfma / loop:.2f}")
