"""Pins oracle/bicodec_ref.py (BiCodec.detokenize, QuarkAudio-UniSE/model/bicodec/bicodec.py:182-199) to the reference's OWN modules
(live, through oracle/ref_bicodec_shim.py) and to the golden waveforms those modules produced (everywhere)."""
import os

import numpy as np
import pytest
import torch

from oracle import bicodec_ref as B
from oracle import gen_golden_bicodec as G
from oracle import ref_bicodec_shim as S
from unified_audio_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
live = pytest.mark.skipif(not S.reference_available(), reason="/root/reference is only mounted in the build container")


@pytest.mark.parametrize("name", list(G.CASES))
def test_oracle_reproduces_reference_waveform_goldens(name):
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    spec, sd, sem, glob = G.case_tensors(name)
    wav = B.detokenize(sd, sem, glob, spec)
    assert wav.shape == g["wav"].shape == (sem.shape[0], 1, sem.shape[1] * spec.hop)
    rms = float(np.sqrt(np.mean(g["wav"] ** 2)))
    assert 0.05 < rms < 0.6 and float(np.abs(g["wav"]).max()) < 0.999  # a live, unsaturated waveform
    assert float(np.abs(wav.numpy() - g["wav"]).max()) < 5e-5


@live
def test_synthetic_state_dict_has_the_reference_key_layout():
    spec = B.BiCodecSpec(**G.SMALL)
    ref = S.load_reference_detokenizer(spec).state_dict()
    want = {k: tuple(v.shape) for k, v in ref.items() if k.startswith(S.DETOK_PREFIXES)}
    got = {k: tuple(v.shape) for k, v in synth.bicodec_state_dict(1, spec).items()}
    assert got == want
    full = {k: tuple(v.shape) for k, v in synth.bicodec_state_dict(1).items()}  # the published shapes
    assert full["quantizer.codebook.weight"] == (8192, 8) and full["decoder.model.1.block.1.weight_v"] == (1536, 768, 16)
    assert full["speaker_encoder.project.weight"] == (1024, 128 * 32) and full["prenet.vocos_backbone.norm.scale.weight"] == (384, 1024)


@live
@pytest.mark.parametrize("kw,batch,frames", [(G.SMALL, 3, 7), (dict(G.SMALL, rates=(2, 3), kernel_sizes=(4, 7), gen_channels=64), 1, 1)])
def test_restatement_matches_reference_modules(kw, batch, frames):
    spec = B.BiCodecSpec(**kw)
    sd = synth.bicodec_state_dict(11, spec)
    sem, glob = synth.bicodec_tokens(12, batch, frames, spec)
    model = S.load_reference_detokenizer(spec, sd)
    ref = model.detokenize(sem, glob)
    taps = {}
    mine = B.detokenize(sd, sem, glob, spec, taps)
    assert float((ref - mine).abs().max()) < 5e-5
    # stage by stage against the reference's own sub-modules
    assert float((model.quantizer.detokenize(sem) - taps["z_q"]).abs().max()) < 1e-5
    assert float((model.speaker_encoder.detokenize(glob) - taps["d_vector"]).abs().max()) < 1e-5


def test_fsq_codes_enumerate_the_implicit_codebook():
    levels = (4, 4, 4, 4, 4, 4)
    codes = B.fsq_codes(torch.arange(4096), levels)
    assert codes.shape == (4096, 6) and set(codes.unique().tolist()) == {-1.0, -0.5, 0.0, 0.5}
    assert len({tuple(r) for r in codes.tolist()}) == 4096          # a bijection
    assert codes[1].tolist() == [-0.5, -1, -1, -1, -1, -1] and codes[4].tolist() == [-1, -0.5, -1, -1, -1, -1]  # least-significant digit first
