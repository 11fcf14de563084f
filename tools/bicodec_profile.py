#!/usr/bin/env python
"""Passes of BiCodec.detokenize (published shapes, B segments x 5 s) for rocprofv3 --kernel-trace --stats.  usage: bicodec_profile.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    print(bench.bicodec_bench(torch.device("cuda:0"), B, reps=2))
