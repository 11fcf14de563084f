"""In-process A/B of run-time knobs (csrc/knobs.h) on the codec hot path: ONE process, one set of weights and inputs, the knob flipped
with qa_set_knob between timed blocks, and every output of every setting compared BIT FOR BIT with the first setting's.

    python tools/knob_ab.py --models 1.0,1.5 --knob QA_LSTM_XCD=1,2,0 [--steps 4] [--batch 32] [--seconds 10]

Prints one line per (model, value): ms per encode+decode step (best and mean of --repeats blocks), `identical` = codes and
waveform equal to the first value's, and a digest of the outputs.  Several --knob arguments are swept one after the other (not as a
product).  --libs default,tools/_variants/X/libquarkaudio_hip.so repeats the sweep per library BUILD (tools/variants.py) and compares the
digests across builds.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default="1.5")
    ap.add_argument("--knob", action="append", required=True, help="NAME=v0,v1,... (v0 is the reference setting)")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--repeats", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--libs", default="", help="comma-separated library builds (tools/variants.py; 'default' = the product library): the whole "
                    "sweep is repeated in one child process per build (QA_LIBRARY) and the digests of the outputs are compared")
    args = ap.parse_args()
    if args.libs:
        import subprocess

        digests = {}
        for lib in args.libs.split(","):
            env = dict(os.environ)
            if lib != "default":
                env["QA_LIBRARY"] = os.path.abspath(lib)
            else:
                env.pop("QA_LIBRARY", None)
            argv = [a for i, a in enumerate(sys.argv) if a != "--libs" and (i == 0 or sys.argv[i - 1] != "--libs") and not a.startswith("--libs=")]
            out = subprocess.run([sys.executable] + argv, env=env, capture_output=True, text=True).stdout
            for line in out.splitlines():
                if line.startswith("H-Codec"):
                    print(f"[{lib}] {line}", flush=True)
                    key = line.split(": best")[0]
                    digests.setdefault(key, []).append((lib, line.rsplit("digest=", 1)[-1]))
        for key, lst in digests.items():
            print(f"{key}: outputs of all builds identical = {len({d for _, d in lst}) == 1}")
        return

    import unified_audio_amd as qa
    from unified_audio_amd import _lib, synth

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    for model in args.models.split(","):
        if model == "2.0":
            sr, spec = 48000, synth.Shapes20()
            codec = qa.Codec(None, None, None, spec=qa.SPEC_20, device=dev).load_state_dict(synth.hcodec20_state_dict(1234, spec))
            hop_in, frame_hop, sem_in = spec.hop, spec.frame_hop, spec.sem_in
        else:
            sr, spec = 16000, (qa.SPEC_15 if model == "1.5" else qa.SPEC_10)
            codec = qa.Codec(None, None, None, spec=spec, device=dev).load_state_dict(synth.hcodec10_state_dict(1234, spec))
            hop_in, frame_hop, sem_in = 320, spec.enc_hop, spec.sem_in
        adaptive = getattr(spec, "adaptive", False)
        B = args.batch
        T = int(round(args.seconds * sr / frame_hop)) * frame_hop
        wav = (synth.synth_wav_fullband(7, B, T) if model == "2.0" else synth.synth_wav(7, B, T)).to(dev)
        feats = synth.synth_feat(9, B, T // hop_in, sem_in).transpose(1, 2).contiguous().to(dev)

        def step():
            if adaptive:
                codes = codec.encode(wav.unsqueeze(1), feats.transpose(1, 2))
                return [codes["acoustic_codes"], codes["semantic_codes"], codec.decode(**codes)]
            ac, sc = codec.encode(wav.unsqueeze(1), feats.transpose(1, 2))
            return [ac, sc, codec.decode(ac, sc)]

        for kv in args.knob:
            name, vals = kv.split("=")
            ref = None
            default = _lib.get_knob(name)
            for v in vals.split(","):
                _lib.set_knob(name, int(v))
                for _ in range(args.warmup):
                    out = step()
                torch.cuda.synchronize(dev)
                times = []
                for _ in range(args.repeats):
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        out = step()
                    torch.cuda.synchronize(dev)
                    times.append(1e3 * (time.perf_counter() - t0) / args.steps)
                if ref is None:
                    ref = [o.clone() for o in out]
                same = all(a.shape == b.shape and torch.equal(a, b) for a, b in zip(out, ref))
                note = ""
                if not same and all(a.shape == b.shape for a, b in zip(out, ref)):  # another summation order: how far apart
                    codes = sum(int((a != b).sum()) for a, b in zip(out[:2], ref[:2])) / sum(a.numel() for a in out[:2])
                    wrel = float((out[2] - ref[2]).pow(2).mean().sqrt() / ref[2].pow(2).mean().sqrt())
                    note = f"  codes differing {codes:.5f}, waveform rel-RMS diff {wrel:.2e}, finite={bool(torch.isfinite(out[2]).all())}"
                import hashlib

                dig = hashlib.sha256(b"".join(o.cpu().numpy().tobytes() for o in out)).hexdigest()[:16]
                note += f"  digest={dig}"
                print(f"H-Codec {model} {B} x {args.seconds:g} s  {name}={v}: best {min(times):.2f} ms, mean {sum(times) / len(times):.2f} ms per step"
                      f"  ({B * T / sr / (min(times) * 1e-3):.0f} audio-s/s)  identical={same}{note}", flush=True)
            _lib.set_knob(name, default)
        del codec
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
