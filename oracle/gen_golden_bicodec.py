"""TEST INFRASTRUCTURE ONLY - golden waveforms of `BiCodec.detokenize`, produced by the reference's OWN modules
(QuarkAudio-UniSE/model/bicodec/modules/*, assembled as bicodec.py:193-199 by oracle/ref_bicodec_shim.py) on seeded weights / tokens
(unified_audio_amd/synth.py generators: the GPU box regenerates the same tensors from the seeds).

Run in the build container:  python -m oracle.gen_golden_bicodec
"""
from __future__ import annotations

import os

import numpy as np
import torch

from oracle import bicodec_ref as B
from oracle import ref_bicodec_shim as S
from unified_audio_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
SMALL = dict(latent_dim=64, codebook_size=128, codebook_dim=8, spk_latent_dim=32, token_num=4, vocos_dim=32, vocos_inter=64, vocos_layers=2,
             gen_channels=256, rates=(4, 5, 2), kernel_sizes=(8, 11, 4))
# full widths of the published configuration with a short backbone and few frames: every kernel shape of the real model, small file
WIDE = dict(vocos_layers=2)
CASES = {  # name -> (spec kwargs, seed, batch, frames)
    "bicodec_small": (SMALL, 3, 2, 9),
    "bicodec_wide": (WIDE, 4, 1, 6),
    # the published configuration itself (12 AdaLN-Vocos blocks, 96 M parameters on the decode side), one segment of 1 s
    "bicodec_published_1s": ({}, 5, 1, 50),
}


def case_tensors(name):
    kw, seed, batch, frames = CASES[name]
    spec = B.BiCodecSpec(**kw)
    sd = synth.bicodec_state_dict(seed, spec)
    sem, glob = synth.bicodec_tokens(seed + 100, batch, frames, spec)
    return spec, sd, sem, glob


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    for name in CASES:
        spec, sd, sem, glob = case_tensors(name)
        model = S.load_reference_detokenizer(spec, sd)
        wav = model.detokenize(sem, glob)
        mine = B.detokenize(sd, sem, glob, spec)
        assert float((wav - mine).abs().max()) < 5e-5, name  # the pin itself (also a test)
        np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), wav=wav.numpy().astype(np.float32))
        print(name, tuple(wav.shape), "rms %.3f" % float(wav.pow(2).mean().sqrt()), "peak %.3f" % float(wav.abs().max()))


if __name__ == "__main__":
    main()
