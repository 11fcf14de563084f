#!/usr/bin/env python
"""Per-phase shader-cycle breakdown of the implicit-GEMM main loop (needs a -DQA_TIMING build: tools/variants.py timing "-DQA_TIMING";
run with QA_LIBRARY=tools/_variants/timing/libquarkaudio_hip.so)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import conv1d_cl  # noqa: E402
from tools.gemm_bench import SHAPES  # noqa: E402
from unified_audio_amd import load_library  # noqa: E402


def main():
    lib = load_library()
    lib.qa_debug_timing.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    dev = torch.device("cuda:0")
    only = os.environ.get("QA_BENCH_ONLY")
    for name, M, N, Cc, k, s in SHAPES:
        if only and not any(name.startswith(o) for o in only.split(",")):
            continue
        x = torch.randn(1, M * s if (k == 1 and s == 1) else M * s + k, Cc, device=dev)
        w = torch.randn(N, k, Cc, device=dev) * 0.05
        b = torch.randn(N, device=dev)
        for _ in range(2):
            conv1d_cl(lib, x, w, b, stride=s, T_out=M)
        torch.cuda.synchronize()
        lib.qa_debug_timing(None, 1)
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        conv1d_cl(lib, x, w, b, stride=s, T_out=M)
        e1.record()
        torch.cuda.synchronize()
        t = (C.c_ulonglong * 10)()
        lib.qa_debug_timing(t, 0)
        waves, chunks = t[6], t[7]
        per = [t[i] / max(chunks, 1) for i in range(4)]
        print(f"{name:18s} {e0.elapsed_time(e1) * 1e3:8.1f} us | per chunk per wave (cycles): addr+issue {per[0]:7.0f}  mfma {per[1]:7.0f}  "
              f"wait+lds-store {per[2]:7.0f}  barrier {per[3]:7.0f} | per wave: epilogue {t[4] / waves:8.0f}  total {t[5] / waves:9.0f}  chunks {chunks / waves:.0f}  shader clock {t[5] / max(t[8], 1) * 100:.0f} MHz")


if __name__ == "__main__":
    main()
