"""TEST INFRASTRUCTURE ONLY - stand-in for `vector-quantize-pytorch==1.22.15` (absent from the
container, not vendored by the reference: /root/reference/QuarkAudio-HCodec/HCodec-1.0/requirements.txt:54).

Restates the eval-mode arithmetic of `ResidualVQ` that the reference calls at
HCodec-1.0/vq/codec.py:101-119 (ctor), :171-172 (forward), :183-184 (get_output_from_indices), following the
in-tree statement of the same algorithm: vq/core_vq.py:223-231 (distance / arg-max of the negative),
:394-404 (residual loop) and :406-412 (decode = sum of look-ups).

PINNED to the reference's in-tree statement: tests/test_rvq_pin_cpu.py runs this stand-in, oracle/rvq_ref.c and
oracle/hcodec_ref.rvq_search/rvq_lookup against `vq.core_vq.ResidualVectorQuantization.encode/decode` imported from
/root/reference (bit-exact indices and sums) and against tests/golden/rvq_corevq_*.npz which that module produced.
The pip package itself (1.22.15) is not reachable here and has never been diffed ("Oracle").
"""
import torch
from torch import nn


class _EuclideanCodebook(nn.Module):
    def __init__(self, codebook_size: int, dim: int):
        super().__init__()
        # upstream keeps the code vectors in a buffer `embed` of shape [heads=1, K, D]
        self.register_buffer("embed", torch.zeros(1, codebook_size, dim))


class _VectorQuantize(nn.Module):
    def __init__(self, codebook_size: int, dim: int):
        super().__init__()
        self._codebook = _EuclideanCodebook(codebook_size, dim)


def nearest_code(residual: torch.Tensor, embed: torch.Tensor) -> torch.Tensor:
    """core_vq.py:223-231: dist = -(|x|^2 - 2 x.e^T + |e|^2); index = argmax(dist) (first max wins)."""
    flat = residual.reshape(-1, residual.shape[-1])
    dist = -(flat.pow(2).sum(1, keepdim=True) - 2 * flat @ embed.t() + embed.pow(2).sum(1)[None, :])
    return dist.max(dim=-1).indices.view(residual.shape[:-1])


class ResidualVQ(nn.Module):
    def __init__(self, *, dim, codebook_size, num_quantizers, **_ignored):
        super().__init__()
        self.layers = nn.ModuleList([_VectorQuantize(codebook_size, dim) for _ in range(num_quantizers)])

    def forward(self, x):  # x [B, N, D]
        residual = x
        quantized_out = torch.zeros_like(x)
        all_indices = []
        for layer in self.layers:  # core_vq.py:394-404
            embed = layer._codebook.embed[0]
            idx = nearest_code(residual, embed)
            quantized = embed[idx]
            residual = residual - quantized
            quantized_out = quantized_out + quantized
            all_indices.append(idx)
        losses = torch.zeros(len(self.layers), dtype=x.dtype)
        return quantized_out, torch.stack(all_indices, dim=-1), losses

    def get_output_from_indices(self, indices):  # [B, N, Q] -> [B, N, D]; core_vq.py:406-412
        # upstream (residual_vq.py get_codes_from_indices): `mask = indices == -1`, the look-up runs on indices with the mask filled with 0,
        # then `all_codes.masked_fill(mask, 0.)`: a dropped code (quantize dropout) contributes a zero vector
        out = 0
        for q, layer in enumerate(self.layers):
            idx = indices[..., q]
            dropped = idx == -1
            codes = layer._codebook.embed[0][idx.masked_fill(dropped, 0)]
            out = out + codes.masked_fill(dropped[..., None], 0.0)
        return out


ResidualSimVQ = ResidualFSQ = ResidualVQ  # names imported by vq/codec.py:14, never constructed
