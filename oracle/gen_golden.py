"""TEST INFRASTRUCTURE ONLY - generate tests/golden/*.npz by running the REFERENCE's own modules
(/root/reference/QuarkAudio-HCodec/HCodec-1.0/vq, imported through oracle/ref_shim.py) on seeded inputs.

Run in the build container (the GPU box has no /root/reference):   python -m oracle.gen_golden
The fixtures only hold seeds, integer codes, the reconstructed waveform and a few activation samples; weights and
inputs are regenerated from the seeds (oracle/synth.py, numpy PCG64).  The RVQ stage of the reference is the third-party
vector_quantize_pytorch package, replaced here by oracle/stubs (PARITY UNPINNED for that stage).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import hcodec_ref as R
from . import ref_shim, synth

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

CASES = [
    # name, weight seed, head_logmag_bias, batch, samples (before pad_wav)
    ("hcodec10_b2_ragged", 1234, 1.5, 2, 640 * 12 + 123),
    ("hcodec10_b1_short", 77, 1.5, 1, 640 * 2),
    ("hcodec10_b1_saturating_head", 78, 4.0, 1, 640 * 4),  # exp(log-mag) runs into the clip at 100 (heads.py:140)
]


def run_case(name, seed, head_bias, batch, samples):
    sd = synth.hcodec10_state_dict(seed, head_logmag_bias=head_bias)
    model = ref_shim.load_state(ref_shim.load_reference_codec("1.0"), sd)
    wav = R.pad_wav(synth.synth_wav(seed + 1, batch, samples))  # HCodecTokenizer.pad_wav
    feat = synth.synth_feat(seed + 2, batch, wav.shape[-1] // 320)
    with torch.no_grad():
        emb = model.encoder(wav.unsqueeze(1))
        sem = model.semantic_encoder(feat)
        ac, sc = model.encode(wav.unsqueeze(1), feat)
        rec = model.decode(ac, sc)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        seed=seed, head_bias=head_bias, batch=batch, samples=samples,
        acoustic_codes=ac.numpy().astype(np.int16), semantic_codes=sc.numpy().astype(np.int16),
        wav_rec=rec.numpy().astype(np.float32),
        emb_sample=emb[:, ::37, ::3].numpy().astype(np.float32), sem_sample=sem[:, ::37, ::3].numpy().astype(np.float32),
        emb_abs_mean=float(emb.abs().mean()), sem_abs_mean=float(sem.abs().mean()),
    )
    print(name, tuple(ac.shape), tuple(rec.shape), "wav rms %.4f max %.3f" % (rec.pow(2).mean().sqrt(), rec.abs().max()))


def main():
    os.makedirs(OUT, exist_ok=True)
    for c in CASES:
        run_case(*c)


if __name__ == "__main__":
    main()
