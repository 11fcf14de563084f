"""Oracle of the SSL front-end against golden vectors produced by transformers' own models (oracle/gen_golden_ssl.py)."""
import dataclasses

import pytest
import torch

from oracle import ssl_ref as S
from tests import ssl_golden_util as GU


@pytest.mark.parametrize("name", GU.NAMES)
def test_oracle_matches_reference_generated_golden(name):
    kind, spec, sd, wav, mean, comp = GU.load(name)
    with torch.no_grad():
        got_mean = S.extract_features(sd, wav, dataclasses.replace(spec, compress_exponent=0.0))
        got_comp = S.extract_features(sd, wav, dataclasses.replace(spec, compress_exponent=0.3))
    assert got_mean.shape == mean.shape
    assert torch.allclose(got_mean, mean, rtol=1e-4, atol=2e-5), float((got_mean - mean).abs().max())
    far = mean.abs() > 1e-3
    assert float((got_comp - comp)[far].abs().max()) < 1e-3
