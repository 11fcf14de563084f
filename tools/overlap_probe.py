#!/usr/bin/env python
"""Does the UniSE decode chain overlap with GEMM-bound work of another stream AT ALL on this device?  
LM generate (B sequences, SE prompt) on stream A from one host thread, a loop of large conv_gemm launches on stream B from a second host thread
(ctypes releases the GIL: two truly concurrent enqueuers), each alone and then both at once.  t_both ~ max(t_lm, t_gemm): the device overlaps
them and a pipelined driver is limited by its host side; t_both ~ t_lm + t_gemm: the device does not.
usage: overlap_probe.py [B] [gemm_launches]      knobs from the environment (QA_GEMM_MAX_WG, ...)"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unified_audio_amd as qa  # noqa: E402
import ctypes as C  # noqa: E402

from unified_audio_amd import _lib  # noqa: E402
from unified_audio_amd import synth as L  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NG = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = torch.device("cuda:0")
lib = qa.load_library()
lm = qa.LLM_SFT(device=dev).load_state_dict(L.lm_state_dict(4321))
mix = L.synth_feats(50, B, 250).to(dev)
mel = torch.zeros(B, 250, 80)
x = torch.randn(1, 16000, 1024, device=dev)
w = torch.randn(3072, 1, 1024, device=dev) / 32
y = torch.empty(1, 16000, 3072, device=dev)
ga = _lib.qa_conv_args()
ga.x, ga.w, ga.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
ga.B, ga.T_in, ga.C_in, ga.T_out, ga.N = 1, 16000, 1024, 16000, 3072
ga.ldx, ga.ldy, ga.ldr, ga.ldg = 1024, 3072, 3072, 3072
ga.ksize, ga.stride = 1, 1
sA = torch.cuda.Stream(dev, priority=-int(os.environ.get("QA_PIPE_LM_PRIO", "0")))
sB = torch.cuda.Stream(dev)


def run_lm():
    with torch.cuda.stream(sA):
        lm.generate("se", None, None, mel, mix, do_sample=False)
        sA.synchronize()


def run_gemm():
    with torch.cuda.stream(sB):
        for _ in range(NG):
            _lib.check(lib.qa_conv1d_cl(C.byref(ga), sB.cuda_stream))
        sB.synchronize()


def timed(*fns):
    torch.cuda.synchronize()
    ts = [threading.Thread(target=f) for f in fns]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


run_lm()
run_gemm()
t_lm = min(timed(run_lm) for _ in range(2))
t_g = min(timed(run_gemm) for _ in range(2))
t_both = min(timed(run_lm, run_gemm) for _ in range(2))
print(f"B={B}: LM alone {t_lm:.1f} ms, {NG} GEMMs alone {t_g:.1f} ms, both {t_both:.1f} ms  (max {max(t_lm, t_g):.1f}, sum {t_lm + t_g:.1f}; "
      f"overlap {100 * (t_lm + t_g - t_both) / min(t_lm, t_g):.0f} % of the shorter)", flush=True)
