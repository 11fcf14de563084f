#!/usr/bin/env python
"""Per-phase shader cycles of the LM decode GEMVs (tuning build -DQA_LM_TIMING, tools/variants.py lm_timing "-DQA_LM_TIMING" with
QA_VARIANT_SOURCES=lm_decode.hip).  usage: QA_LIBRARY=tools/_variants/lm_timing/libquarkaudio_hip.so python tools/lm_timing.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unified_audio_amd as qa  # noqa: E402
from unified_audio_amd import synth as L  # noqa: E402

dev = torch.device("cuda:0")
lib = qa.load_library()
lm = qa.LLM_SFT(device=dev).load_state_dict(L.lm_state_dict(4321))
mix = L.synth_feats(50, 16, 250).to(dev)
mel = torch.zeros(16, 250, 80)
lm.generate("se", None, None, mel, mix, do_sample=False)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 36)()
lib.qa_debug_lm_timing.argtypes = [C.c_void_p, C.c_int]
lib.qa_debug_lm_timing(None, 1)
lm.generate("se", None, None, mel, mix, do_sample=False)
torch.cuda.synchronize()
lib.qa_debug_lm_timing(buf, 0)
names = {0: "qkv", 1: "gate/up", 2: "down", 3: "head", 4: "o_proj(+merge)"}
print("kind            workgroups   loads->ready   MFMAs   reduce+barrier   epilogue   total cycles (wave 0 of a workgroup, mean)")
for k, nm in names.items():
    row = [buf[k * 6 + i] for i in range(6)]
    n = max(row[5], 1)
    print(f"{nm:15s} {row[5]:10d} {row[0] / n:12.0f} {row[1] / n:9.0f} {row[2] / n:14.0f} {row[3] / n:10.0f} {sum(row[:4]) / n:12.0f}")
