"""C-ABI checks that need no GPU: the library loads, exports every symbol include/quarkaudio.h declares, fails loudly
without a device, and its host-side padding logic equals the reference's SConv1d / pad1d."""
import ctypes as C
import math
import os
import re

import pytest
import torch

from oracle import hcodec_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported_and_bound(qa_lib):
    from unified_audio_amd import _lib

    header = open(os.path.join(ROOT, "include", "quarkaudio.h")).read()
    declared = set(re.findall(r"\b(qa_[a-z0-9_]+)\s*\(", header))
    assert {"qa_hcodec_encode", "qa_hcodec_decode", "qa_rvq_search", "qa_lm_generate"} <= declared
    for name in sorted(declared):
        assert hasattr(qa_lib, name), f"{name} declared in quarkaudio.h but not exported"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    assert qa_lib.qa_version() == 103


def test_knob_table_is_enumerable_settable_and_documented(qa_lib):
    """csrc/knobs.h: every tuning switch is a row of one table - enumerable through the C-ABI, settable at run time, initialised from
    the environment variable of its name, and listed in INTEGRATION.md."""
    from unified_audio_amd import _lib

    rows = _lib.knobs()
    assert {"QA_SERIAL", "QA_LSTM_PERSISTENT", "QA_GEMM_CFG", "QA_LM_CHAINS"} <= set(rows)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name, (value, default, text) in rows.items():
        assert name.startswith("QA_") and text and f"`{name}`" in doc, f"{name} is not documented in INTEGRATION.md"
        if name not in os.environ:
            assert value == default, name
    old = _lib.set_knob("QA_GEMM_CFG", 2)
    assert _lib.get_knob("QA_GEMM_CFG") == 2 and _lib.set_knob("QA_GEMM_CFG", old) == 2
    with pytest.raises(_lib.QuarkAudioError):
        _lib.set_knob("QA_NO_SUCH_KNOB", 1)
    csrc = os.path.join(ROOT, "unified_audio_amd", "csrc")
    for f in os.listdir(csrc):  # the library reads the environment in ONE place (api.cpp: the knob table and QA_GEMM_SHAPES)
        if f != "api.cpp":
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f


def test_struct_layouts_match_header(qa_lib):
    from unified_audio_amd import _lib

    assert C.sizeof(_lib.qa_hcodec_spec) == 4 * (31 + 9 + 6 + 5)  # 21 + 9 + 6 + 5 (causal flags / contexts) scalar 4-byte fields, ratios[8] and sem_strides[4] inline
    assert C.sizeof(_lib.qa_tensor) == 24
    assert C.sizeof(_lib.qa_conv_args) == 7 * 8 + 9 * 8 + 9 * 4 + 4  # 9 int32 + tail padding to 8
    assert C.sizeof(_lib.qa_lm_spec) == 40


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-device error path")
def test_no_device_fails_loudly(qa_lib):
    import unified_audio_amd as qa
    from oracle import synth
    from tests.util import MINI, mini_oracle_spec

    assert qa_lib.qa_device_count() == 0
    sd = synth.hcodec10_state_dict(1, mini_oracle_spec())
    with pytest.raises(qa.QuarkAudioError):
        qa.Codec(spec=qa.HCodecSpec(**MINI)).load_state_dict(sd)


def test_codec_without_weights_raises(qa_lib):
    import unified_audio_amd as qa

    c = qa.Codec(device="cuda:0")
    with pytest.raises(qa.QuarkAudioError):
        c.encode(torch.zeros(1, 1, 640), torch.zeros(1, 768, 2))


@pytest.mark.parametrize("k,s", [(7, 1), (3, 1), (4, 2), (8, 4), (10, 5), (16, 8), (1, 1)])
def test_sconv_geometry_matches_reference_rule(qa_lib, k, s):
    """encoder_modules/conv.py:54-61,195-211 restated by oracle.hcodec_ref._extra_padding."""
    for L in list(range(1, 40)) + [639, 640, 641, 160000]:
        t_out, left, right = C.c_int64(), C.c_int32(), C.c_int32()
        assert qa_lib.qa_sconv_geometry(L, k, s, C.byref(t_out), C.byref(left), C.byref(right)) == 0
        pad_total = k - s
        extra = R._extra_padding(L, k, s, pad_total)
        assert right.value == pad_total // 2 + extra and left.value == pad_total - pad_total // 2
        assert t_out.value == (L + left.value + right.value - k) // s + 1 == math.ceil(L / s)


@pytest.mark.parametrize("L,left,right", [(10, 3, 3), (2, 4, 4), (1, 3, 5), (3, 3, 2), (5, 0, 0), (4, 8, 11)])
def test_resolve_frame_matches_reference_pad1d(qa_lib, L, left, right):
    """Reflect padding incl. the short-input branch (encoder_modules/conv.py:79-96): pad a ramp, read it back."""
    x = torch.arange(1, L + 1, dtype=torch.float32).view(1, 1, L)
    ref = R._pad1d_reflect(x, left, right)[0, 0]
    got = []
    for p in range(-left, L + right):
        i = qa_lib.qa_resolve_frame(p, L, max(left, right), 1)
        got.append(0.0 if i < 0 else float(x[0, 0, i]))
    assert got == ref.tolist()
    zero = [qa_lib.qa_resolve_frame(p, L, max(left, right), 0) for p in range(-left, L + right)]
    assert zero == [-1] * left + list(range(L)) + [-1] * right
