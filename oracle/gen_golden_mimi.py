"""TEST INFRASTRUCTURE ONLY - tests/golden/mimi_stream.npz: the reference's own StreamingTransformer
(HCodec-1.5/adaptive/model_blocks/mimi/transformer.py, through oracle/ref_shim.load_reference_mimi) causal with a context window,
evaluated offline and under `with model.streaming(B)` on the chunk pattern of tests/test_mimi_stream_cpu.py.

Run in the build container:   python -m oracle.gen_golden_mimi
"""
import os

import numpy as np
import torch

from . import ref_shim, synth

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mimi_stream.npz")
D, H, L, FF, CTX = 128, 4, 3, 256, 6
CHUNKS = (1, 1, 3, 2, 1, 6, 4, 1, 5)


def main(seed: int = 40):
    sd = synth.mimi_state_dict(seed, D, L, FF)
    model = ref_shim.load_reference_mimi(D, H, L, FF, True, CTX)
    model.load_state_dict({k[len("transformer."):]: v for k, v in sd.items()}, strict=True)
    x = torch.randn(2, sum(CHUNKS), D, generator=torch.Generator().manual_seed(seed + 1))
    with torch.no_grad():
        offline = model(x)
        out, at = [], 0
        with model.streaming(2):
            for c in CHUNKS:
                out.append(model(x[:, at:at + c]))
                at += c
    np.savez_compressed(OUT, seed=seed, offline=offline.numpy(), streamed=torch.cat(out, 1).numpy())
    print("mimi_stream", tuple(offline.shape), float((offline - torch.cat(out, 1)).abs().max()))


if __name__ == "__main__":
    main()
