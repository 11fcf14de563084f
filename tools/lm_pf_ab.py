#!/usr/bin/env python
"""In-process A/B of the decode step's cross-launch prefetch planes (QA_LM_PF bit mask, csrc/lm_decode.h PfArgs): generate time per mask
at B segments, token checksums must not move.   usage: lm_pf_ab.py B TASK MASK[,MASK...] [REPS]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unified_audio_amd as qa  # noqa: E402
from unified_audio_amd import _lib, synth as L  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
TASK = sys.argv[2] if len(sys.argv) > 2 else "se"
MASKS = [int(m) for m in (sys.argv[3] if len(sys.argv) > 3 else "0,1,2,4,8,15").split(",")]
REPS = int(sys.argv[4]) if len(sys.argv) > 4 else 3
KNOB = sys.argv[5] if len(sys.argv) > 5 else "QA_LM_PF"
S = 250
dev = torch.device("cuda:0")
lm = qa.LLM_SFT(device=dev).load_state_dict(L.lm_state_dict(4321))
mix = L.synth_feats(50, B, 250).to(dev)
enr = L.synth_feats(51, B, 250).to(dev) if TASK != "se" else None
mel = torch.zeros(B, S, 80)
w = torch.arange(1, S + 1, device=dev)


def run():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g, s = lm.generate(TASK, None if enr is None else mel, enr, mel, mix, do_sample=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, int((g * w[: g.shape[1]]).sum()), int((s * w).sum())


run()
ref = None
for rnd in range(2):  # two rounds over the masks: drift shows up as a difference between the rounds
    for m in MASKS:
        _lib.set_knob(KNOB, m)
        ts = []
        for _ in range(REPS):
            t, cg, cs = run()
            ts.append(t)
        ref = ref or (cg, cs)
        print(f"B={B} {TASK} {KNOB}={m:3d} round {rnd}: min {min(ts):7.2f} ms  median {sorted(ts)[len(ts) // 2]:7.2f} ms  "
              f"{B * (33 + S) / min(ts) * 1e3:8.0f} tok/s  tokens {'same' if (cg, cs) == ref else 'DIFFERENT'}", flush=True)
