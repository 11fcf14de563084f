"""TEST INFRASTRUCTURE ONLY - the reference's OWN `vq.Codec` (imported from /root/reference through oracle/ref_shim.py) timed on
this container's host cores beside the oracle restatement ("port") that bench.py's cpu_baseline leg times on the GPU box, on the
same seeded weights and clips, so that the GPU / CPU ratio in the bench line has a reference-code anchor.

    python -m oracle.time_reference_cpu [--model 1.5] [--clips 8] [--seconds 10] [--reps 2]

Runs only where /root/reference exists (the build container); prints one JSON line; results are recorded in BASELINE.md section 2a.
"""
from __future__ import annotations

import argparse
import json
import os
import time

import torch

from . import hcodec15_ref as R15
from . import hcodec_ref as R
from . import ref_shim, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=["1.0", "1.5"], default="1.5")
    ap.add_argument("--clips", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    cores = len(os.sched_getaffinity(0))
    torch.set_num_threads(cores)
    spec = R.SPEC_15 if args.model == "1.5" else R.SPEC_10
    sd = synth.hcodec10_state_dict(1234, spec)
    T = int(round(args.seconds * 16000 / spec.enc_hop)) * spec.enc_hop
    wav = synth.synth_wav(101, args.clips, T)
    feat = synth.synth_feat(102, args.clips, T // 320, spec.sem_in)
    model = ref_shim.load_state(ref_shim.load_reference_codec(args.model, spec if args.model == "1.5" else None), sd)

    def run_reference():
        if spec.adaptive:
            codes = model.encode(wav.unsqueeze(1), feat)
            return codes["acoustic_codes"], model.decode(codes["acoustic_codes"], codes["semantic_codes"])
        ac, sc = model.encode(wav.unsqueeze(1), feat)
        return ac, model.decode(ac, sc)

    def run_port():
        if spec.adaptive:
            codes = R15.encode(sd, wav.unsqueeze(1), feat, spec)
            return codes["acoustic_codes"], R15.decode(sd, codes["acoustic_codes"], codes["semantic_codes"], spec)
        ac, sc = R.encode(sd, wav.unsqueeze(1), feat, spec)
        return ac, R.decode(sd, ac, sc, spec)

    out = {}
    results = {}
    with torch.no_grad():
        for kind, fn in (("reference", run_reference), ("port", run_port)):
            best = float("inf")
            for i in range(args.reps + 1):  # first pass = warm-up
                t0 = time.perf_counter()
                results[kind] = fn()
                dt = time.perf_counter() - t0
                if i:
                    best = min(best, dt)
            out[kind] = {"value": args.clips * T / 16000 / best, "unit": "audio-seconds/sec", "seconds_per_pass": best}
    same_codes = bool(torch.equal(results["reference"][0], results["port"][0]))
    wav_err = float((results["reference"][1] - results["port"][1]).pow(2).mean().sqrt() / results["reference"][1].pow(2).mean().sqrt())
    print(json.dumps({"model": f"H-Codec {args.model}", "cores": cores, "clips": args.clips, "clip_seconds": T / 16000, "reps": args.reps,
                      "reference_vq_Codec": out["reference"], "oracle_port": out["port"], "port_over_reference": out["port"]["value"] / out["reference"]["value"],
                      "codes_identical": same_codes, "waveform_rel_rms_port_vs_reference": wav_err}))


if __name__ == "__main__":
    main()
