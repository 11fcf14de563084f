"""Minimal torchaudio.transforms used by HCodec-2.0/vq/codec_encoder.py (Spectrogram, Resample)."""
import torch
from torch import nn


class Spectrogram(nn.Module):
    def __init__(self, n_fft, hop_length, win_length=None, center=False, power=None, **_):
        super().__init__()
        self.n_fft, self.hop = n_fft, hop_length
        self.center, self.power = center, power
        self.register_buffer("window", torch.hann_window(win_length or n_fft), persistent=False)

    def forward(self, x):
        s = torch.stft(x, self.n_fft, self.hop, window=self.window, center=self.center, return_complex=True)
        return s if self.power is None else s.abs().pow(self.power)


class Resample(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, x):
        raise NotImplementedError


class MelSpectrogram(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
