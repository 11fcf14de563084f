"""TEST INFRASTRUCTURE ONLY - golden vectors for the RVQ stage, produced by the reference's OWN in-tree statement of the
algorithm: /root/reference/QuarkAudio-HCodec/HCodec-1.0/vq/core_vq.py (`ResidualVectorQuantization.encode` :394-404,
`.decode` :406-412, `EuclideanCodebook.quantize` :223-231).  The module is loaded from where it lies (never copied) and
fed seeded inputs; the outputs are committed as tests/golden/rvq_corevq_*.npz so the GPU box (which has no
/root/reference) can check both the oracle and the HIP kernel against the reference's numbers.

Run in the build container:  python -m oracle.gen_golden_rvq
"""
from __future__ import annotations

import importlib.util
import os

import numpy as np
import torch

from oracle import ref_shim

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
CASES = {  # name -> (seed, n_vec, Q, K, D)
    "rvq_corevq_hcodec10": (11, 416, 4, 1024, 512),     # H-Codec 1.0 / 1.5 codebook geometry (vq/codec.py:101-119)
    "rvq_corevq_hcodec20": (12, 96, 16, 1024, 512),     # H-Codec 2.0: 16 stages (conf/large_12.5hz_config.yaml:22-29)
    "rvq_corevq_small": (13, 257, 3, 64, 128),          # ragged vector count, the mini geometry of the whole-graph tests
}


def load_core_vq():
    path = os.path.join(ref_shim.REFERENCE_ROOT, "QuarkAudio-HCodec", "HCodec-1.0", "vq", "core_vq.py")
    spec = importlib.util.spec_from_file_location("qa_ref_core_vq", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def case_inputs(seed: int, n: int, Q: int, K: int, D: int):
    """Seeded inputs (numpy PCG64): residual-scaled codebooks like SURVEY.md 8d, inputs near the first codebook's scale."""
    rng = np.random.default_rng(seed)
    cb = np.stack([rng.standard_normal((K, D)).astype(np.float32) * np.float32(0.5 ** q) for q in range(Q)])
    x = rng.standard_normal((n, D)).astype(np.float32)
    return x, cb


def reference_rvq(mod, x: np.ndarray, cb: np.ndarray):
    Q, K, D = cb.shape
    rvq = mod.ResidualVectorQuantization(num_quantizers=Q, dim=D, codebook_size=K, kmeans_init=False).eval()
    for q, layer in enumerate(rvq.layers):
        layer._codebook.embed.data.copy_(torch.from_numpy(cb[q]))
    xt = torch.from_numpy(x).t()[None]  # the reference's layout [b, d, n]
    with torch.no_grad():
        idx = rvq.encode(xt)            # [Q, b, n]
        quant = rvq.decode(idx)         # [b, d, n]
        quant_fwd, idx_fwd, _ = rvq(xt)
    assert torch.equal(idx, idx_fwd)
    return idx[:, 0].t().contiguous().numpy(), quant[0].t().contiguous().numpy(), quant_fwd[0].t().contiguous().numpy()


def main():
    mod = load_core_vq()
    os.makedirs(GOLDEN, exist_ok=True)
    for name, (seed, n, Q, K, D) in CASES.items():
        x, cb = case_inputs(seed, n, Q, K, D)
        idx, quant, quant_fwd = reference_rvq(mod, x, cb)
        np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), seed=seed, n=n, Q=Q, K=K, D=D, indices=idx.astype(np.int16),
                            quant_sample=quant[::7, ::5].copy(), quant_fwd_sample=quant_fwd[::7, ::5].copy())
        print(name, idx.shape)


if __name__ == "__main__":
    main()
