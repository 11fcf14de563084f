"""HIP path against the committed golden vectors produced by the reference's own modules (tests/golden, oracle/gen_golden.py)."""
import dataclasses
import glob
import os

import numpy as np
import pytest
import torch

from oracle import hcodec_ref as R
from oracle import synth
from tests.util import audit_codes_bnq

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hcodec10_*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_hip_path_reproduces_reference_golden(qa_lib, gpu_device, path):
    import unified_audio_amd as qa

    g = np.load(path)
    seed = int(g["seed"])
    sd = synth.hcodec10_state_dict(seed, head_logmag_bias=float(g["head_bias"]))
    if "stress" in g.files and int(g["stress"]):  # hcodec10_b1_stress (oracle/gen_golden.py CASES_STRESS)
        sd = synth.stress_state_dict(sd)
    causal = bool(int(g["causal"])) if "causal" in g.files else False  # hcodec10_b2_causal: the reference's blocks built causal=True
    ospec = dataclasses.replace(R.SPEC_10, causal=causal)
    tok = qa.HCodecTokenizer(state_dict=sd, device=gpu_device, spec=qa.HCodecSpec(causal=causal))
    wav = synth.synth_wav(seed + 1, int(g["batch"]), int(g["samples"]))  # un-padded: tokenize() pads like the reference
    padded = R.pad_wav(wav)
    feat = synth.synth_feat(seed + 2, int(g["batch"]), padded.shape[-1] // 320)
    ac, sc = tok.tokenize(wav.to(gpu_device), feats=feat.transpose(1, 2).contiguous().to(gpu_device))
    ref_ac = torch.from_numpy(g["acoustic_codes"].astype(np.int64))
    ref_sc = torch.from_numpy(g["semantic_codes"].astype(np.int64))
    assert ac.shape == ref_ac.shape and ac.dtype == torch.int64
    # integer output: EQUAL to the reference's codes except where the decision is a near-tie (tests/util.audit_codes); the
    # RVQ inputs the audit judges on are the oracle's, whose codes are the golden ones bit for bit (tests/test_oracle_cpu.py)
    taps = {}
    with torch.no_grad():
        ac_o, sc_o = R.encode(sd, padded.unsqueeze(1), feat, ospec, taps=taps)
    assert torch.equal(ac_o, ref_ac) and torch.equal(sc_o, ref_sc)
    audit_codes_bnq(taps["enc.emb"], R.rvq_codebooks(sd, "quantizer", 4), ac, ref_ac)
    audit_codes_bnq(taps["enc.sem"], R.rvq_codebooks(sd, "semantic_quantizer", 4), sc, ref_sc)
    rec = tok.detokenize(ref_ac.to(gpu_device), ref_sc.to(gpu_device)).cpu().numpy()
    rms = float(np.sqrt(np.mean((rec - g["wav_rec"]) ** 2)))
    assert rms < 1e-3 * max(1.0, float(np.sqrt(np.mean(g["wav_rec"] ** 2)))), rms  # north_star: <= 1e-3 RMS
    assert rms / float(np.sqrt(np.mean(g["wav_rec"] ** 2))) < 1e-4


GOLDEN_15 = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hcodec15_*.npz")))


@pytest.mark.parametrize("path", GOLDEN_15, ids=[os.path.basename(p)[:-4] for p in GOLDEN_15])
def test_hip_path_reproduces_reference_golden_15(qa_lib, gpu_device, path):
    """H-Codec 1.5 (full width; 2-layer adaptive stacks, or - hcodec15_b2_full_depth - the published 32-layer stacks) against vectors
    produced by the reference's vq.Codec."""
    import dataclasses

    import unified_audio_amd as qa

    g = np.load(path)
    seed = int(g["seed"])
    flags = {k: (bool(g[k]) if k.endswith("causal") else int(g[k])) for k in ("agg_causal", "agg_context", "bt_causal", "bt_context") if k in g.files}
    layers = int(g["layers"]) if "layers" in g.files else 2  # hcodec15_b2_full_depth: the published 32-layer stacks
    ospec = dataclasses.replace(R.SPEC_15, agg_layers=layers, bt_layers=layers, threshold=float(g["threshold"]), **flags)
    sd = synth.hcodec10_state_dict(seed, ospec)
    if "stress" in g.files and int(g["stress"]):
        sd = synth.stress_state_dict(sd)
    kw = {f: getattr(ospec, f) for f in ospec.__dataclass_fields__}
    tok = qa.HCodecTokenizer(state_dict=sd, device=gpu_device, spec=qa.HCodecSpec(**kw))
    wav = synth.synth_wav(seed + 1, int(g["batch"]), int(g["samples"]))
    feat = synth.synth_feat(seed + 2, int(g["batch"]), R.pad_wav(wav).shape[-1] // 320, ospec.sem_in)
    codes = tok.tokenize(wav.to(gpu_device), feats=feat.transpose(1, 2).contiguous().to(gpu_device))
    ref_ac = torch.from_numpy(g["acoustic_codes"].astype(np.int64))
    ref_sc = torch.from_numpy(g["semantic_codes"].astype(np.int64))
    assert codes["acoustic_codes"].shape == ref_ac.shape  # same number of groups
    assert torch.equal(codes["semantic_codes"].cpu() // 1024, ref_sc // 1024)  # identical token lengths
    assert torch.equal(codes["acoustic_codes"].cpu() // 1024, ref_ac // 1024)
    from oracle import hcodec15_ref as R15
    taps = {}
    with torch.no_grad():
        ref = R15.encode(sd, R.pad_wav(wav).unsqueeze(1), feat, ospec, taps)
    assert torch.equal(ref["acoustic_codes"], ref_ac) and torch.equal(ref["semantic_codes"], ref_sc)
    audit_codes_bnq(taps["enc.emb_agg"], R.rvq_codebooks(sd, "quantizer", 4), codes["acoustic_codes"] % 1024, ref_ac % 1024)
    audit_codes_bnq(taps["enc.sem_agg"], R.rvq_codebooks(sd, "semantic_quantizer", 4), codes["semantic_codes"] % 1024, ref_sc % 1024)
    rec = tok.detokenize(acoustic_codes=ref_ac.to(gpu_device), semantic_codes=ref_sc.to(gpu_device)).cpu().numpy()
    assert rec.shape == g["wav_rec"].shape
    rms = float(np.sqrt(np.mean((rec - g["wav_rec"]) ** 2)))
    assert rms < 1e-3 and rms / float(np.sqrt(np.mean(g["wav_rec"] ** 2))) < 1e-4, rms


@pytest.mark.parametrize("name", ["hcodec20_small_b2", "hcodec20_small_b2_causal", "hcodec20_b1_full", "hcodec20_small_b2_stress"])
def test_hip_path_reproduces_reference_golden_20(qa_lib, gpu_device, name):
    """H-Codec 2.0 against vectors produced by the reference's vq.Codec built from a reduced YAML (the second with `causal: true`) and -
    hcodec20_b1_full - from the shipped large_12.5hz_config.yaml shapes (24 + 32 ConvNeXt blocks at width 1536, 1.17 G parameters)."""
    import unified_audio_amd as qa
    from oracle import hcodec20_ref as R20
    from oracle.gen_golden import SPEC20_SMALL

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    causal = bool(int(g["causal"])) if "causal" in g.files else False
    full = bool(int(g["full"])) if "full" in g.files else False
    seed, o = int(g["seed"]), (R20.HCodec20Spec(causal=causal) if full else R20.HCodec20Spec(**SPEC20_SMALL, causal=causal))
    sd = synth.hcodec20_state_dict(seed, o)
    if "stress" in g.files and int(g["stress"]):
        sd = synth.stress_state_dict(sd)
    pspec = qa.HCodecSpec(version=20, enc_dim=o.enc_dim, enc_inter=o.enc_inter, enc_convnext_layers=o.enc_convnext_layers,
                          enc_layers=o.enc_transformer_layers, frame_stride=o.stride, tr_inter_cap=o.tr_inter_cap, dimension=o.dimension,
                          code_dim=o.dimension, sem_in=o.sem_in, sem_ch=o.sem_ch, sem_strides=o.sem_strides, codebook_size=o.codebook_size,
                          num_quantizers=o.num_quantizers, dec_dim=o.dec_dim, dec_inter=o.dec_inter, dec_heads=o.dec_dim // 64,
                          dec_layers=o.dec_transformer_layers, convnext_layers=o.dec_convnext_layers, n_fft=o.n_fft, hop=o.hop, causal=causal)
    tok = qa.HCodecTokenizer(state_dict=sd, device=gpu_device, spec=pspec)
    assert tok.hop_length == 3840  # int(sampling_rate / target_frame_rate), HCodec-2.0/audio_tokenizer.py:41
    wav = synth.synth_wav_fullband(seed + 1, int(g["batch"]), int(g["samples"]))
    feat = synth.synth_feat(seed + 2, int(g["batch"]), R.pad_wav(wav, 3840).shape[-1] // o.hop, o.sem_in)
    ac, sc = tok.tokenize(wav.to(gpu_device), feats=feat.transpose(1, 2).contiguous().to(gpu_device))
    ref_ac, ref_sc = torch.from_numpy(g["acoustic_codes"].astype(np.int64)), torch.from_numpy(g["semantic_codes"].astype(np.int64))
    assert ac.shape == ref_ac.shape
    taps = {}
    with torch.no_grad():
        ac_o, sc_o = R20.encode(sd, R.pad_wav(wav, 3840), feat, o, taps)
    cb_a, cb_s = R.rvq_codebooks(sd, "quantizer", o.num_quantizers), R.rvq_codebooks(sd, "semantic_quantizer", o.num_quantizers)
    if "stress" in name:
        # RANGE-STRESS case.  The chain is: reference == oracle, bit for bit, on the host that made the golden (tests/test_oracle_cpu.py in the
        # build container); HIP == oracle ON THIS HOST up to audited near-ties at the usual tolerance (below).  What is NOT required is that the
        # CPU oracle of another host reproduces every code of the golden: with LSTM weights x 4 and LayerScale ~ 1 the torch CPU kernels of two
        # hosts differ by ~1e-4 of a frame's norm (measured: the GPU box's oracle and the HIP path agree to 2e-6 and BOTH leave the golden at
        # the same vector, top-2 gap 6e-4 of E|x|^2 - profiles/r06_stress_goldens.txt), so that link is a count, not an audit.
        left = int((ac_o != ref_ac).any(dim=1).sum() + (sc_o != ref_sc).any(dim=1).sum())
        assert left <= max(1, ref_ac.shape[0] * ref_ac.shape[2] // 5), left
        audit_codes_bnq(taps["enc.emb"], cb_a, ac, ac_o)
        audit_codes_bnq(taps["enc.sem"], cb_s, sc, sc_o)
    else:
        assert torch.equal(ac_o, ref_ac) and torch.equal(sc_o, ref_sc)
        audit_codes_bnq(taps["enc.emb"], cb_a, ac, ref_ac)
        audit_codes_bnq(taps["enc.sem"], cb_s, sc, ref_sc)
    rec = tok.detokenize(ref_ac.to(gpu_device), ref_sc.to(gpu_device)).cpu().numpy()
    assert rec.shape == g["wav_rec"].shape
    assert float(np.sqrt(np.mean((rec - g["wav_rec"]) ** 2)) / np.sqrt(np.mean(g["wav_rec"] ** 2))) < 1e-4
