"""Structural pins of the Resample oracle (torchaudio is absent: see oracle/resample_ref.py) and of the host-side filter
design in the library (qa_resample_length; the taps themselves are checked on the GPU against this oracle)."""
import math

import pytest
import torch

from oracle import resample_ref as RR


@pytest.mark.parametrize("orig,new,T", [(48000, 16000, 48000), (48000, 16000, 3841), (44100, 16000, 4410), (16000, 16000, 100)])
def test_length_rule_and_library_agrees(qa_lib, orig, new, T):
    x = torch.randn(2, T, generator=torch.Generator().manual_seed(0))
    y = RR.resample(x, orig, new)
    assert y.shape == (2, math.ceil(new * T / orig))
    assert qa_lib.qa_resample_length(T, orig, new) == y.shape[-1]


def test_filter_is_a_unit_gain_lowpass():
    k, width, orig, new = RR.sinc_resample_kernel(48000, 16000)
    assert (orig, new, width, k.shape[-1]) == (3, 1, 19, 41)  # ceil(6 * 3 / 0.99) = 19, 2 * 19 + 3 taps
    assert abs(float(k.sum()) - 1.0) < 2e-3                   # DC gain of the decimator
    t = torch.arange(48000, dtype=torch.float64) / 48000
    for f, keep in ((1000.0, True), (6000.0, True), (12000.0, False), (20000.0, False)):  # 16 kHz Nyquist = 8 kHz
        y = RR.resample(torch.sin(2 * math.pi * f * t).float()[None], 48000, 16000)[0, 200:-200]
        rms = float(y.pow(2).mean().sqrt())
        assert (abs(rms - 2 ** -0.5) < 0.02) if keep else (rms < 0.02), (f, rms)
    ref = torch.sin(2 * math.pi * 1000.0 * torch.arange(16000, dtype=torch.float64) / 16000).float()
    y = RR.resample(torch.sin(2 * math.pi * 1000.0 * t).float()[None], 48000, 16000)[0]
    assert float((y - ref)[200:-200].abs().max()) < 5e-3      # same phase: the filter is zero-delay (symmetric about its centre)
