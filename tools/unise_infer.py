#!/usr/bin/env python
"""Inference entry point with the command line of the reference's `test.py` (QuarkAudio-UniSE/test.py:11-39 -> Model.test_step), on the
MI355X path (WavLM front-end -> AR-LM -> BiCodec detokenize):

    python tools/unise_infer.py --config conf/config.yaml --save_enhanced DIR

reads the reference's YAML: `codec_ckpt_dir` (BiCodec/config.yaml + BiCodec/model.safetensors), `llm_config`, `ckpt_path` (the Lightning
checkpoint, its `dnn.*` entries), `dataset_config.test_kwargs` (mode, data_src_dir, data_enroll_dir, enroll_duration) - plus ONE added key,
`semantic_model_path`: a local snapshot of microsoft/wavlm-base-plus (the reference downloads it; there is no network here).  Every file
of `data_src_dir` is one test batch `(mode, enroll, src, tgt, fs, lengths, names)`; all of them go through `Model.test_steps` together (the
reference runs one file per step, data_module.py:340 - same results, segments of all files share the launches).  With several ranks
(`torch.distributed.run`), files are dealt rank-strided like data_module.py:364.

Positional wav files replace the dataset section, `--mode` / `--enroll` replace its mode and enrollment directory:

    python tools/unise_infer.py --config conf.yaml --save_enhanced DIR --mode tse --enroll spk1.wav,spk2.wav mix1.wav mix2.wav
    python tools/unise_infer.py --synthetic --mode se --save_enhanced DIR a.wav      # seeded random weights at the published sizes: plumbing / timing only
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unified_audio_amd as qa  # noqa: E402
from unified_audio_amd import audio_io, synth  # noqa: E402
from unified_audio_amd.unise import Model, TestDataset, wrap_pad  # noqa: E402


def build_model(a, config, dev):
    if a.synthetic:
        lm = qa.LLM_SFT(device=dev).load_state_dict(synth.lm_state_dict(4321))
        fx = qa.SSLFeatureExtractor(qa.SPEC_WAVLM_BASE_PLUS, device=dev).load_state_dict(synth.ssl_state_dict(qa.SPEC_WAVLM_BASE_PLUS))
        bic = qa.BiCodec(device=dev).load_state_dict(synth.bicodec_state_dict(77))
        return Model(config, device=dev, semantic_model=fx, tokenizer=qa.BiCodecTokenizer(model=bic), dnn=lm)
    return Model(config, device=dev)


def batches_from_files(a, dev):
    """The batch tuples of data_module.py:338-384 for wav files named on the command line."""
    enrolls = None
    if a.mode == "tse":
        if not a.enroll:
            raise SystemExit("--mode tse needs --enroll (one wav for all inputs, or a comma-separated list, one per input)")
        paths = a.enroll.split(",")
        paths = paths * len(a.wavs) if len(paths) == 1 else paths
        if len(paths) != len(a.wavs):
            raise SystemExit(f"{len(paths)} enrollments for {len(a.wavs)} inputs")
        n = int(a.enroll_duration * 16000)
        enrolls = []
        for p in paths:  # data_module.py:346-352: wrapped or cut to enroll_duration, peak 0.99 - every file keeps ITS OWN samples
            e = audio_io.load_audio(p, 16000, dev)
            e = wrap_pad(e, n)[..., :n] if e.shape[-1] < n else e[..., :n]
            enrolls.append(e / (e.abs().max() + 1e-5) * 0.99)
    for i, p in enumerate(a.wavs):
        src = audio_io.load_audio(p, 16000, dev)
        yield (a.mode, None if enrolls is None else enrolls[i], src, src, torch.tensor([16000]), torch.tensor([src.shape[-1]]),
               [os.path.splitext(os.path.basename(p))[0]])


def main():
    ap = argparse.ArgumentParser("test model")
    ap.add_argument("wavs", nargs="*", help="wav files (instead of config['dataset_config']['test_kwargs'])")
    ap.add_argument("--config", type=str, default=None, help="the reference's YAML (test.py --config)")
    ap.add_argument("--save_enhanced", "--out", dest="save_enhanced", type=str, default=None, help="The dir path to save enhanced wavs.")
    ap.add_argument("--mode", choices=["se", "tse", "ss"], default=None)
    ap.add_argument("--enroll", default=None, help="tse with positional wavs: enrollment wav (one for all inputs, or a comma-separated list)")
    ap.add_argument("--enroll_duration", type=float, default=5.0)
    ap.add_argument("--synthetic", action="store_true", help="seeded random weights at the published sizes instead of checkpoints")
    ap.add_argument("--device", default=None)
    a = ap.parse_args()
    config = {}
    if a.config:
        import yaml

        with open(a.config, "r") as f:
            config = yaml.safe_load(f)
    elif not a.synthetic:
        raise SystemExit("--config (the reference's YAML) or --synthetic is required")
    if a.save_enhanced is not None:  # test.py:15-17
        config["save_enhanced"] = a.save_enhanced
        os.makedirs(a.save_enhanced, exist_ok=True)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    dev = torch.device(a.device or f"cuda:{int(os.environ.get('LOCAL_RANK', 0))}")
    model = build_model(a, config, dev)
    if a.wavs:
        a.mode = a.mode or "se"
        batches = list(batches_from_files(a, dev))[rank::world]
    else:
        kw = dict(((config.get("dataset_config") or {}).get("test_kwargs")) or {})
        if not kw:
            raise SystemExit("no wav files given and the config has no dataset_config.test_kwargs")
        if a.mode:
            kw["mode"] = a.mode
        batches = list(TestDataset(**kw, device=dev, rank=rank, world_size=world))
    model.test_steps(batches)
    print(f"rank {rank}: {len(batches)} file(s) done" + (f", written to {config['save_enhanced']}" if config.get("save_enhanced") else ""))


if __name__ == "__main__":
    sys.exit(main())
