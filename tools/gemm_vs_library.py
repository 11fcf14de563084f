#!/usr/bin/env python
"""Calibration only (never a product path): conv_gemm's LINEAR launches against the vendor library's fp32 GEMM (torch.matmul -> rocBLAS / hipBLASLt)
on the shapes that carry the H-Codec FLOPs.  usage: gemm_vs_library.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unified_audio_amd as qa  # noqa: E402
from unified_audio_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = qa.load_library()
torch.backends.cuda.matmul.allow_tf32 = False


def timed(fn, reps=30):
    for _ in range(5):
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
print("M x N x K                 conv_gemm us (TFLOP/s)    torch.matmul fp32 us (TFLOP/s)   max |diff| / max |ref|")
for M, N, K in ((16000, 3072, 1024), (16000, 1024, 3072), (16000, 4096, 1024), (8000, 3072, 1024), (9056, 1536, 512), (9056, 2048, 512), (9056, 512, 2048), (9056, 512, 512)):
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    y = torch.empty(M, N, device=dev)
    y2 = torch.empty(M, N, device=dev)
    a = _lib.qa_conv_args()
    a.x, a.w, a.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
    a.B, a.T_in, a.C_in, a.T_out, a.N = 1, M, K, M, N
    a.ldx, a.ldy, a.ldr, a.ldg = K, N, N, N
    a.ksize, a.stride = 1, 1
    t0 = timed(lambda: _lib.check(lib.qa_conv1d_cl(C.byref(a), st)))
    wt = w.t().contiguous()
    t1 = min(timed(lambda: torch.matmul(x, w.t(), out=y2)), timed(lambda: torch.matmul(x, wt, out=y2)))
    fl = 2.0 * M * N * K
    torch.matmul(x, w.t(), out=y2)
    err = float((y - y2).abs().max() / y2.abs().max())
    print(f"{M:6d} x {N:5d} x {K:5d}      {t0:8.1f} ({fl / t0 / 1e6:6.1f})          {t1:8.1f} ({fl / t1 / 1e6:6.1f})                 {err:.2e}", flush=True)
