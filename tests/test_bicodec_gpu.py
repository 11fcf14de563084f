"""BiCodec.detokenize through the C-ABI against the CPU oracle (pinned to the reference's own modules) and the golden waveforms
those modules produced."""
import os

import numpy as np
import pytest
import torch

from oracle import bicodec_ref as B
from oracle import gen_golden_bicodec as G
from tests.util import rel_err
from unified_audio_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
STAGE_TOL = 5e-5


def _model(spec, sd, device):
    import unified_audio_amd as qa

    kw = {f: getattr(spec, f) for f in spec.__dataclass_fields__}
    return qa.BiCodec(qa.BiCodecSpec(**kw), device=device).load_state_dict(sd)


def _cl(t):  # oracle [B, C, T] -> library [B, T, C], flat
    return t.transpose(1, 2).contiguous().flatten()


def _run(spec, seed, batch, frames, device):
    sd = synth.bicodec_state_dict(seed, spec)
    sem, glob = synth.bicodec_tokens(seed + 100, batch, frames, spec)
    m = _model(spec, sd, device).enable_taps()
    taps = {}
    want = B.detokenize(sd, sem, glob, spec, taps)
    got = m.detokenize(sem.to(device), glob.to(device))
    torch.cuda.synchronize()
    assert got.shape == want.shape == (batch, 1, frames * spec.hop)
    report = {"z_q": rel_err(m.tap("z_q"), _cl(taps["z_q"])), "d_vector": rel_err(m.tap("d_vector"), taps["d_vector"].flatten()),
              "prenet.down": rel_err(m.tap("prenet.down"), taps["prenet.down"].flatten()),
              "prenet.backbone": rel_err(m.tap("prenet.backbone"), taps["prenet.backbone"].flatten()),
              "prenet.out": rel_err(m.tap("prenet.out"), _cl(taps["prenet.out"]))}
    n = len(spec.rates)
    for i in range(n):  # the library keeps a block's output already activated by the next consumer's Snake
        alpha = sd[f"decoder.model.{i + 2}.block.0.alpha"] if i + 1 < n else sd[f"decoder.model.{n + 1}.alpha"]
        report[f"gen.block{i}"] = rel_err(m.tap(f"gen.block{i}"), _cl(B.snake(taps[f"gen.block{i}"], alpha)))
    report["wav"] = rel_err(got.cpu(), want)
    return report, got.cpu(), want


def test_small_detokenize_stage_parity(qa_lib, gpu_device):
    spec = B.BiCodecSpec(**G.SMALL)
    report, got, want = _run(spec, 21, 3, 11, gpu_device)
    print(report)
    assert all(v < STAGE_TOL for v in report.values()), report
    assert float((got - want).abs().max()) < 1e-4


def test_odd_geometry(qa_lib, gpu_device):
    """stride-3 / k=7 and stride-2 / k=4 ConvTranspose1d (phases with 3 and 2 taps), a single frame, batch 1."""
    spec = B.BiCodecSpec(**dict(G.SMALL, rates=(3, 2), kernel_sizes=(7, 4), gen_channels=128))
    report, got, want = _run(spec, 22, 1, 1, gpu_device)
    assert all(v < STAGE_TOL for v in report.values()), report


def test_published_widths_short_backbone(qa_lib, gpu_device):
    """Every kernel shape of the published configuration (1024-wide latents, 384 / 2048 Vocos, 1536-channel generator with rates
    8, 5, 4, 2) with a 2-layer backbone, 2 segments x 8 frames."""
    spec = B.BiCodecSpec(vocos_layers=2)
    report, got, want = _run(spec, 23, 2, 8, gpu_device)
    print(report)
    assert all(v < 2 * STAGE_TOL for v in report.values()), report
    assert float((got - want).pow(2).mean().sqrt()) < 1e-3  # north_star's waveform tolerance


@pytest.mark.parametrize("name", list(G.CASES))
def test_reproduces_reference_module_goldens(qa_lib, gpu_device, name):
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    spec, sd, sem, glob = G.case_tensors(name)
    got = _model(spec, sd, gpu_device).detokenize(sem.to(gpu_device), glob.to(gpu_device)).cpu().numpy()
    assert got.shape == g["wav"].shape
    rms = float(np.sqrt(np.mean((got - g["wav"]) ** 2)))
    assert rms < 1e-3 and rms / float(np.sqrt(np.mean(g["wav"] ** 2))) < 1e-4, rms


def test_full_model_5s_segment_runs_and_is_deterministic(qa_lib, gpu_device):
    """The published configuration at the UniSE working point: 5 s segments (250 tokens -> 80 000 samples), batch 2."""
    spec = B.SPEC_BICODEC
    sd = synth.bicodec_state_dict(31, spec)
    sem, glob = synth.bicodec_tokens(32, 2, 250, spec)
    m = _model(spec, sd, gpu_device)
    a = m.detokenize(sem.to(gpu_device), glob.to(gpu_device))
    b = m.detokenize(sem.to(gpu_device), glob.to(gpu_device))
    assert a.shape == (2, 1, 80000) and torch.isfinite(a).all() and torch.equal(a, b)
    assert float(a.abs().max()) <= 1.0 and float(a.pow(2).mean().sqrt()) > 0.01
    one = m.detokenize(sem[1:].to(gpu_device), glob[1:].to(gpu_device))
    assert torch.equal(one[0], a[1])  # batch invariance
    # against the oracle on the first second of clip 0 would need the whole receptive field: compare the full clip instead
    want = B.detokenize(sd, sem[:1], glob[:1], spec)
    assert float((a[:1].cpu() - want).pow(2).mean().sqrt()) < 1e-3 and rel_err(a[:1].cpu(), want) < 2e-4


def test_argument_errors(qa_lib, gpu_device):
    import unified_audio_amd as qa

    spec = B.BiCodecSpec(**G.SMALL)
    sd = synth.bicodec_state_dict(1, spec)
    m = _model(spec, sd, gpu_device)
    sem, glob = synth.bicodec_tokens(2, 1, 3, spec)
    with pytest.raises(IndexError):
        m.detokenize(sem + spec.codebook_size, glob)
    with pytest.raises(IndexError):
        m.detokenize(sem, glob + 4096)
    with pytest.raises(qa.QuarkAudioError):
        m.detokenize(sem, glob[:, :, :3])
    del sd["decoder.model.0.bias"]
    with pytest.raises(qa.QuarkAudioError):
        _model(spec, sd, gpu_device)
