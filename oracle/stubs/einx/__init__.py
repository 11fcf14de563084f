"""TEST INFRASTRUCTURE ONLY - stand-in for `einx` (absent here).  The reference's
QuarkAudio-UniSE/model/bicodec/modules/fsq/residual_fsq.py:10,131 uses exactly one call:
    get_at("q [c] d, b n q -> q b n d", codebooks, indices)      codebooks [q, c, d], indices [b, n, q]
i.e. out[q, b, n, :] = codebooks[q, indices[b, n, q], :]."""
import torch


def get_at(pattern, codebooks, indices):
    assert pattern.replace(" ", "") == "q[c]d,bnq->qbnd", pattern
    q = codebooks.shape[0]
    return torch.stack([codebooks[i][indices[..., i]] for i in range(q)], dim=0)
