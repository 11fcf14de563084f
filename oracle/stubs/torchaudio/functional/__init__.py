from . import functional  # noqa: F401


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk"):
    """torchaudio.functional.melscale_fbanks with its defaults (HTK scale, no normalisation): triangular filters [n_freqs, n_mels].
    Only the frame count of the mel reaches the arithmetic of the UniSE path (llm_sft.py:108)."""
    import math

    import torch

    assert norm is None and mel_scale == "htk"
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    hz2mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)  # noqa: E731
    m_pts = torch.linspace(hz2mel(f_min), hz2mel(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down, up = -slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.minimum(down, up), min=0.0)
