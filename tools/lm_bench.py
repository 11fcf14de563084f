#!/usr/bin/env python
"""UniSE LM generate micro-benchmark (B segments, SE prompt 252, 33 + 250 greedy steps)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unified_audio_amd as qa  # noqa: E402
from unified_audio_amd import synth as L  # noqa: E402  (seeded weights / features: data generation only)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
S = int(sys.argv[3]) if len(sys.argv) > 3 else 250  # semantic steps
TASK = sys.argv[4] if len(sys.argv) > 4 else "se"  # "tse": 250-frame enrollment in the prompt (BASELINE configs[3])
dev = torch.device("cuda:0")
lm = qa.LLM_SFT(device=dev).load_state_dict(L.lm_state_dict(4321))
mix = L.synth_feats(50, B, 250).to(dev)
enr = L.synth_feats(51, B, 250).to(dev) if TASK != "se" else None
mel = torch.zeros(B, S, 80)
for i in range(REPS):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g, s = lm.generate(TASK, None if enr is None else mel, enr, mel, mix, do_sample=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"B={B} generate {dt * 1e3:.1f} ms  {B * (33 + S) / dt:.0f} tok/s", flush=True)
# checksum of the token streams: A/B builds that must not change a single token (tools/variants.py) are compared on it
w = torch.arange(1, s.shape[1] + 1, device=s.device)
print(f"ids checksum global {int((g * w[: g.shape[1]]).sum())} semantic {int((s * w).sum())}", flush=True)
