#!/usr/bin/env python
"""Experiment: the BASELINE step (H-Codec 1.5 encode+decode, 32 x 10 s) as ONE handle over the whole batch vs TWO handles, each
over half of the batch, driven by two host threads on two streams (their kernels co-run: one half's GEMM prologues / epilogues and
LSTM steps under the other half's MFMA phases).  usage: python tools/split_bench.py [steps]"""
import os
import sys
import threading
import time

import torch

os.environ.setdefault("QA_LSTM_PERSISTENT", "0")  # concurrent handles on one GPU: no kernel that needs the whole device resident
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unified_audio_amd as qa  # noqa: E402
from unified_audio_amd import synth  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    ways = int(os.environ.get("QA_SPLIT_WAYS", "2"))
    dev = torch.device("cuda:0")
    spec = qa.SPEC_15
    sd = synth.hcodec10_state_dict(1234, spec)
    B, T = 32, 160000
    wav = synth.synth_wav(7, B, T).to(dev)
    feats = synth.synth_feat(9, B, T // 320, spec.sem_in).transpose(1, 2).contiguous().to(dev)
    codecs = [qa.Codec(None, None, None, spec=spec, device=dev).load_state_dict(sd) for _ in range(ways)]
    streams = [torch.cuda.Stream(dev) for _ in range(ways)]

    def step(codec, w, f):
        codes = codec.encode(w.unsqueeze(1), f.transpose(1, 2))
        return codec.decode(**codes)

    def run_whole(n):
        for _ in range(n):
            out = step(codecs[0], wav, feats)
        torch.cuda.synchronize(dev)
        return out

    def run_split(n):
        outs = [None] * ways
        per = B // ways

        def worker(i):
            with torch.cuda.stream(streams[i]):
                for _ in range(n):
                    outs[i] = step(codecs[i], wav[i * per:(i + 1) * per], feats[i * per:(i + 1) * per])

        th = [threading.Thread(target=worker, args=(i,)) for i in range(ways)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize(dev)
        return torch.cat(outs, 0)

    ref = run_whole(2)
    t0 = time.perf_counter()
    run_whole(steps)
    whole = (time.perf_counter() - t0) / steps
    got = run_split(2)
    t0 = time.perf_counter()
    run_split(steps)
    split = (time.perf_counter() - t0) / steps
    same = torch.equal(ref, got)
    err = float((ref - got).abs().max())
    print(f"whole batch: {whole * 1e3:.2f} ms/step   {ways}-way split on {ways} streams: {split * 1e3:.2f} ms/step   "
          f"waveforms identical: {same} (max |diff| {err:.3g})")


if __name__ == "__main__":
    main()
