#!/usr/bin/env python
"""Inference entry point in the shape of the reference's `test.py` (QuarkAudio-UniSE/test.py -> Model.test_step): read wav files,
run UniSE 'se' / 'tse' / 'ss' on the MI355X path (WavLM front-end -> AR-LM -> BiCodec detokenize), write the enhanced wav files.

    python tools/unise_infer.py --mode se  --ckpt model.ckpt --wavlm wavlm_state.pt --bicodec BiCodec_dir --out outdir a.wav b.wav
    python tools/unise_infer.py --mode tse --enroll spk.wav ... mix.wav
    python tools/unise_infer.py --mode se --synthetic --out outdir a.wav      # seeded random weights: plumbing / timing only

Checkpoints: `--ckpt` the reference's Lightning checkpoint (its `dnn.*` entries are the LM, model.py:82-91), `--wavlm` a
`WavLMModel.state_dict()` of microsoft/wavlm-base-plus saved with torch.save, `--bicodec` the Spark-TTS BiCodec directory with
model.safetensors.  All utterances of one call form ONE batch of 5 s segments (the reference handles one file per step)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unified_audio_amd as qa  # noqa: E402
from unified_audio_amd import audio_io, synth  # noqa: E402
from unified_audio_amd.unise import UniSE  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("wavs", nargs="+")
    ap.add_argument("--mode", choices=["se", "tse", "ss"], default="se")
    ap.add_argument("--enroll", default=None, help="tse: enrollment wav (one for all inputs, or a comma-separated list)")
    ap.add_argument("--ckpt"), ap.add_argument("--wavlm"), ap.add_argument("--bicodec")
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--out", default="enhanced")
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args()
    dev = torch.device(a.device)
    if a.synthetic:
        lm_sd, ssl_sd, bic = synth.lm_state_dict(4321), None, qa.BiCodec(device=dev).load_state_dict(synth.bicodec_state_dict(77))
        from bench import _ssl_state_dict  # seeded HF-layout weights

        ssl_sd = _ssl_state_dict(qa.SPEC_WAVLM_BASE_PLUS)
    else:
        ck = torch.load(a.ckpt, map_location="cpu")
        lm_sd = ck.get("state_dict", ck)
        ssl_sd = torch.load(a.wavlm, map_location="cpu")
        bic = qa.BiCodec.load_from_checkpoint(a.bicodec, device=dev)
    lm = qa.LLM_SFT(device=dev).load_state_dict(lm_sd)  # `dnn.` prefix stripped, other entries ignored
    fx = qa.SSLFeatureExtractor(qa.SPEC_WAVLM_BASE_PLUS, device=dev).load_state_dict(ssl_sd)
    drv = UniSE(lm, fx, tokenizer=qa.BiCodecTokenizer(model=bic))
    srcs = [audio_io.load_audio(p, 16000, dev) for p in a.wavs]
    enrolls = None
    if a.mode == "tse":
        paths = a.enroll.split(",")
        paths = paths * len(srcs) if len(paths) == 1 else paths
        enrolls = [audio_io.load_audio(p, 16000, dev) for p in paths]
        n = min(e.shape[-1] for e in enrolls)
        enrolls = [e[:, :n] for e in enrolls]
    outs = drv.enhance(a.mode, srcs, enrolls)
    os.makedirs(a.out, exist_ok=True)
    for p, o in zip(a.wavs, outs):
        stem = os.path.splitext(os.path.basename(p))[0]
        if a.mode == "ss":
            audio_io.write_wav(os.path.join(a.out, stem + "_s1.wav"), o[0], 16000)
            audio_io.write_wav(os.path.join(a.out, stem + "_s2.wav"), o[1], 16000)
        else:
            audio_io.write_wav(os.path.join(a.out, stem + ".wav"), o, 16000)
    print(f"wrote {len(outs)} result(s) to {a.out}")


if __name__ == "__main__":
    main()
