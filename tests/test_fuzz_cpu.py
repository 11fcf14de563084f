"""Hypothesis-driven checks of the host logic and the pure-host C-ABI helpers (no device needed): padding geometry against the
reference's rules, the UniSE driver's segmenting arithmetic against numpy / torch.stft, the Resample length rule."""
import ctypes as C
import math

import numpy as np
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import hcodec_ref as R
from unified_audio_amd import unise

FUZZ = settings(max_examples=150, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))


@FUZZ
@given(L=st.integers(1, 5000), k=st.integers(1, 32), s=st.integers(1, 32))
def test_sconv_geometry_fuzz(qa_lib, L, k, s):
    """qa_sconv_geometry == the (left, right + extra) of SConv1d.forward (encoder_modules/conv.py:195-211) for any k >= s."""
    if s > k:
        k, s = s, k
    t_out, left, right = C.c_int64(), C.c_int32(), C.c_int32()
    assert qa_lib.qa_sconv_geometry(L, k, s, C.byref(t_out), C.byref(left), C.byref(right)) == 0
    y = R.sconv1d(torch.zeros(1, 1, L), torch.zeros(1, 1, k), None, s)
    assert t_out.value == y.shape[-1] == math.ceil(L / s)
    pt = k - s
    assert left.value == pt - pt // 2 and right.value == pt // 2 + R._extra_padding(L, k, s, pt)


@FUZZ
@given(L=st.integers(1, 60), left=st.integers(0, 40), right=st.integers(0, 40))
def test_resolve_frame_fuzz(qa_lib, L, left, right):
    """reflect padding incl. the short-input branch of pad1d (encoder_modules/conv.py:79-96): pad a ramp, read it back"""
    x = torch.arange(1, L + 1, dtype=torch.float32).view(1, 1, L)
    ref = R._pad1d_reflect(x, left, right)[0, 0].tolist()
    got = []
    for p in range(-left, L + right):
        i = qa_lib.qa_resolve_frame(p, L, max(left, right), 1)
        got.append(0.0 if i < 0 else float(x[0, 0, i]))
    assert got == ref


@FUZZ
@given(T=st.integers(1, 400000))
def test_wrap_pad_and_mel_frames_fuzz(T):
    """model.py:176-182 (np.pad 'wrap' to a multiple of 5 s) and the frame count of stft_logmel (model.py:53-79)"""
    T = min(T, 200000)
    src = torch.arange(T, dtype=torch.float32).view(1, T)
    pad_len = math.ceil(T / unise.SEG_LEN) * unise.SEG_LEN - T
    want = np.pad(src.numpy(), [(0, 0), (0, pad_len)], "wrap")
    assert np.array_equal(unise.wrap_pad(src).numpy(), want)
    n = min(T, 20000)  # the frame count only depends on n; keep the STFT small
    assert unise.mel_frames(n) == unise.stft_logmel(torch.zeros(1, n)).shape[1]


@FUZZ
@given(T=st.integers(1, 10 ** 7), orig=st.sampled_from([8000, 16000, 22050, 24000, 44100, 48000]),
       new=st.sampled_from([8000, 16000, 22050, 24000, 44100, 48000]))
def test_resample_length_fuzz(qa_lib, T, orig, new):
    """torchaudio's target length ceil(new * T / orig) (functional.py:_apply_sinc_resample_kernel) in exact integer arithmetic"""
    g = math.gcd(orig, new)
    assert qa_lib.qa_resample_length(T, orig, new) == -(-(new // g) * T // (orig // g))
