"""TEST INFRASTRUCTURE ONLY - CPU restatement of torchaudio.transforms.Resample(orig_freq, new_freq) with its defaults
(resampling_method="sinc_interp_hann", lowpass_filter_width=6, rolloff=0.99), the 48 kHz -> 16 kHz step in front of HuBERT in
the reference's H-Codec 2.0 tokenizer (QuarkAudio-HCodec/HCodec-2.0/audio_tokenizer.py:44,51).

torchaudio is a third-party dependency that is absent from this container and not vendored by the reference
(HCodec-2.0/requirements: torchaudio): the algorithm below restates its published `functional._get_sinc_resample_kernel` /
`_apply_sinc_resample_kernel` (kernel index arithmetic in float64, kernel stored as float32, zero padding (width, width + orig),
strided conv1d, output cut to ceil(new * T / orig)).  PARITY UNPINNED with respect to the torchaudio package itself; pinned
structurally by tests/test_resample_cpu.py (length rule, unit DC gain, pass-band sinusoids preserved, stop-band rejected).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    kernels = kernels * window * (base / orig)
    return kernels.to(torch.float32), width, orig, new  # [new, 1, 2 * width + orig]


def resample(wav: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """wav [..., T] float32 -> [..., ceil(new * T / orig)]."""
    kernel, width, orig, new = sinc_resample_kernel(orig_freq, new_freq)
    if orig == new:
        return wav
    shape = wav.shape
    x = wav.reshape(-1, shape[-1])
    n, length = x.shape
    x = F.pad(x, (width, width + orig))
    y = F.conv1d(x[:, None], kernel, stride=orig)
    y = y.transpose(1, 2).reshape(n, -1)
    target = math.ceil(new * length / orig)
    return y[..., :target].reshape(shape[:-1] + (target,))
