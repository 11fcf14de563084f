"""HIP SSL front-end (through the C-ABI) against golden vectors produced by transformers' own models."""
import dataclasses

import pytest
import torch

from tests import ssl_golden_util as GU
from tests.util import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", GU.NAMES)
def test_hip_matches_reference_generated_golden(qa_lib, gpu_device, name):
    import unified_audio_amd as qa

    kind, spec, sd, wav, mean, comp = GU.load(name)
    kw = {f: getattr(spec, f) for f in spec.__dataclass_fields__}
    for expo, want in ((0.0, mean), (0.3, comp)):
        fx = qa.SSLFeatureExtractor(qa.SSLSpec(**{**kw, "compress_exponent": expo}), device=gpu_device).load_state_dict(sd)
        got = fx(wav.to(gpu_device)).cpu()
        assert got.shape == want.shape
        if expo == 0.0:
            assert rel_err(got, want) < 5e-5
        else:
            far = mean.abs() > 1e-3
            assert float((got - want)[far].abs().max()) < 1e-3
