import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def qa_lib():
    """The C-ABI library; built on demand where hipcc exists (no compute is run by CPU tests)."""
    from unified_audio_amd import _lib, build

    if not os.path.exists(_lib.lib_path()):
        build.build_library()
    return _lib.load_library()


@pytest.fixture(scope="session")
def gpu_device():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("a test marked `gpu` ran on a machine without a GPU")
    return torch.device("cuda:0")
