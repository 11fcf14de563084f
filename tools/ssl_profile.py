#!/usr/bin/env python
"""Three passes of the SSL front-end (default: XLSR-53 architecture, 32 x 10 s) for rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    model = sys.argv[1] if len(sys.argv) > 1 else "1.5"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    print(bench.ssl_bench(torch.device("cuda:0"), model, B, 10.0 if model != "unise" else 5.0, reps=2))
