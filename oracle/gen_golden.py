"""TEST INFRASTRUCTURE ONLY - generate tests/golden/*.npz by running the REFERENCE's own modules
(/root/reference/QuarkAudio-HCodec/HCodec-1.0/vq, imported through oracle/ref_shim.py) on seeded inputs.

Run in the build container (the GPU box has no /root/reference):   python -m oracle.gen_golden
The fixtures only hold seeds, integer codes, the reconstructed waveform and a few activation samples; weights and
inputs are regenerated from the seeds (oracle/synth.py, numpy PCG64).  The RVQ stage of the reference is the third-party
vector_quantize_pytorch package, replaced here by oracle/stubs (pinned to vq/core_vq.py by tests/test_rvq_pin_cpu.py).
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


CASES_CAUSAL = [  # the reference's own encoder / decoder classes built with causal=True (ref_shim.make_causal_10)
    ("hcodec10_b2_causal", 4321, 1.5, 2, 640 * 9 + 77),
]


# RANGE STRESS: one case per codec version with synth.stress_state_dict applied - LSTM weights x 4 (saturated gates),
# LayerScale / ConvNeXt gamma ~ 1 (full-size residual updates), ISTFT log-magnitude bias at ln 100 (half of the bins in the clip)
CASES_STRESS = [("hcodec10_b1_stress", 79, 1.5, 1, 640 * 6 + 100)]
CASES_15_STRESS = [("hcodec15_b2_stress", 1530, 2, 640 * 20 + 30, 0.6)]


def run_case(name, seed, head_bias, batch, samples, causal=False, stress=False):
    sd = synth.hcodec10_state_dict(seed, head_logmag_bias=head_bias)
    if stress:
        sd = synth.stress_state_dict(sd)
    model = ref_shim.load_reference_codec("1.0")
    if causal:
        model = ref_shim.make_causal_10(model)
    model = ref_shim.load_state(model, sd)
    wav = R.pad_wav(synth.synth_wav(seed + 1, batch, samples))  # HCodecTokenizer.pad_wav
    feat = synth.synth_feat(seed + 2, batch, wav.shape[-1] // 320)
    with torch.no_grad():
        emb = model.encoder(wav.unsqueeze(1))
        sem = model.semantic_encoder(feat)
        ac, sc = model.encode(wav.unsqueeze(1), feat)
        rec = model.decode(ac, sc)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        seed=seed, head_bias=head_bias, batch=batch, samples=samples, causal=int(causal), stress=int(stress),
        acoustic_codes=ac.numpy().astype(np.int16), semantic_codes=sc.numpy().astype(np.int16),
        wav_rec=rec.numpy().astype(np.float32),
        emb_sample=emb[:, ::37, ::3].numpy().astype(np.float32), sem_sample=sem[:, ::37, ::3].numpy().astype(np.float32),
        emb_abs_mean=float(emb.abs().mean()), sem_abs_mean=float(sem.abs().mean()),
    )
    print(name, tuple(ac.shape), tuple(rec.shape), "wav rms %.4f max %.3f" % (rec.pow(2).mean().sqrt(), rec.abs().max()))


CASES_15 = [
    # name, seed, batch, samples, threshold  (full-width H-Codec 1.5 with 2-layer adaptive stacks: the layer count is a YAML knob
    # of the reference - config_adaptive_v3.yaml:86,95,103 - so this is the reference's own code path)
    ("hcodec15_b2_thr060", 1500, 2, 640 * 24 + 50, 0.6),
    ("hcodec15_b2_thr072", 1501, 2, 640 * 24, 0.72),
]


CASES_15_CAUSAL = [  # the YAML's `causal: true` + context windows for the aggregators and the bottleneck (config_adaptive_v3.yaml:84-105)
    ("hcodec15_b2_causal_stacks", 1510, 2, 640 * 20, 0.66, dict(agg_causal=True, agg_context=6, bt_causal=True, bt_context=9)),
]


CASES_15_FULL = [  # the published depth itself: 32-layer aggregators and bottleneck (654 M parameters), the reference's vq.Codec on CPU
    ("hcodec15_b2_full_depth", 1520, 2, 640 * 48 + 50, 0.72, None, 32),
]


def run_case_15(name, seed, batch, samples, threshold, flags=None, layers=2, stress=False):
    import dataclasses

    from . import hcodec15_ref  # noqa: F401  (same spec object the tests use)

    spec = dataclasses.replace(R.SPEC_15, agg_layers=layers, bt_layers=layers, threshold=threshold, **(flags or {}))
    sd = synth.hcodec10_state_dict(seed, spec)
    if stress:
        sd = synth.stress_state_dict(sd)
    model = ref_shim.load_state(ref_shim.load_reference_codec("1.5", spec), sd)
    wav = R.pad_wav(synth.synth_wav(seed + 1, batch, samples))
    feat = synth.synth_feat(seed + 2, batch, wav.shape[-1] // 320, spec.sem_in)
    with torch.no_grad():
        codes = model.encode(wav.unsqueeze(1), feat)
        rec = model.decode(codes["acoustic_codes"], codes["semantic_codes"])
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), seed=seed, batch=batch, samples=samples, threshold=threshold, layers=layers, stress=int(stress),
        **{k: int(v) for k, v in (flags or {}).items()},
        acoustic_codes=codes["acoustic_codes"].numpy().astype(np.int32), semantic_codes=codes["semantic_codes"].numpy().astype(np.int32),
        wav_rec=rec.numpy().astype(np.float32))
    print(name, tuple(codes["acoustic_codes"].shape), tuple(rec.shape), "lens", (codes["semantic_codes"][0, 0] // 1024 + 1).tolist())


SPEC20_SMALL = dict(enc_dim=256, enc_inter=512, enc_convnext_layers=2, enc_transformer_layers=1, dimension=128, sem_in=64, sem_ch=128,
                    codebook_size=64, num_quantizers=5, dec_dim=256, dec_inter=512, dec_convnext_layers=2, dec_transformer_layers=1)


def run_case_20(name, seed, batch, samples, causal=False, full=False, stress=False):
    """H-Codec 2.0 built by the reference from a reduced YAML (same code path as the 1.28 B-parameter configuration), or - `full` -
    from the shipped large_12.5hz_config.yaml shapes themselves (24 + 32 ConvNeXt blocks at width 1536, 16 + 16 codebooks)."""
    from . import hcodec20_ref as R20

    spec = R20.HCodec20Spec(causal=causal) if full else R20.HCodec20Spec(**SPEC20_SMALL, causal=causal)
    sd = synth.hcodec20_state_dict(seed, spec)
    if stress:
        sd = synth.stress_state_dict(sd)
    model = ref_shim.load_state(ref_shim.load_reference_codec("2.0", spec), sd)
    wav = R.pad_wav(synth.synth_wav_fullband(seed + 1, batch, samples), spec.frame_hop)
    feat = synth.synth_feat(seed + 2, batch, wav.shape[-1] // spec.hop, spec.sem_in)
    with torch.no_grad():
        ac, sc = model.encode(wav, feat)
        rec = model.decode(ac, sc)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, batch=batch, samples=samples, causal=int(causal), full=int(full), stress=int(stress),
                        acoustic_codes=ac.numpy().astype(np.int16), semantic_codes=sc.numpy().astype(np.int16),
                        wav_rec=rec.numpy().astype(np.float32))
    print(name, tuple(ac.shape), tuple(rec.shape))


def main():
    os.makedirs(OUT, exist_ok=True)
    import sys

    if "--full" in sys.argv:  # the two published configurations at full size (minutes of CPU time, ~10 GB of host memory)
        for c in CASES_15_FULL:
            run_case_15(*c)
        run_case_20("hcodec20_b1_full", 2010, 1, 3840 * 12 + 700, full=True)
        return
    if "--stress" in sys.argv:
        for c in CASES_STRESS:
            run_case(*c, stress=True)
        for c in CASES_15_STRESS:
            run_case_15(*c, stress=True)
        run_case_20("hcodec20_small_b2_stress", 2002, 2, 3840 * 4 + 300, stress=True)
        return
    run_case_20("hcodec20_small_b2", 2000, 2, 3840 * 5 + 1000)
    run_case_20("hcodec20_small_b2_causal", 2001, 2, 3840 * 4 + 500, causal=True)
    for c in CASES:
        run_case(*c)
    for c in CASES_CAUSAL:
        run_case(*c, causal=True)
    for c in CASES_15:
        run_case_15(*c)
    for c in CASES_15_CAUSAL:
        run_case_15(*c)


if __name__ == "__main__":
    main()
