import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_terminal_summary(terminalreporter):
    """What the near-tie audits of integer codes actually saw (tests/util.audit_codes): the tolerances are set from this."""
    try:
        from tests.util import AUDIT_LOG, CODE_TIE_TOL
    except Exception:  # noqa: BLE001
        return
    if AUDIT_LOG:
        n = sum(a[0] for a in AUDIT_LOG)
        terminalreporter.write_line(
            f"code audits: {len(AUDIT_LOG)} audits over {n} vectors; largest flip fraction {max(a[1] for a in AUDIT_LOG):.5f}, "
            f"largest accepted near-tie gap {max(a[2] for a in AUDIT_LOG):.2e} and largest excess {max(a[3] for a in AUDIT_LOG):.2e} "
            f"(relative to E|x|^2; bar {CODE_TIE_TOL:.0e})")


@pytest.fixture(scope="session")
def qa_lib():
    """The C-ABI library; built on demand where hipcc exists (no compute is run by CPU tests)."""
    from unified_audio_amd import _lib, build

    if not os.path.exists(_lib.lib_path()):
        build.build_library()
    return _lib.load_library()


@pytest.fixture
def knob(qa_lib):
    """knob(name, value): set a tuning knob of the library (csrc/knobs.h) for the duration of one test."""
    from unified_audio_amd import _lib

    saved = {}

    def _set(name, value):
        old = _lib.set_knob(name, value)
        saved.setdefault(name, old)

    yield _set
    for name, old in saved.items():
        _lib.set_knob(name, old)


@pytest.fixture(scope="session")
def gpu_device():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("a test marked `gpu` ran on a machine without a GPU")
    return torch.device("cuda:0")
