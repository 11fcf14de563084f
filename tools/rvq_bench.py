#!/usr/bin/env python
"""The RVQ search alone (qa_rvq_search = ResidualVQ.forward) at a given shape - what bench.py's `extras.rvq_search_*` time,
without any model around it, so that rocprofv3 passes over it stay short.
usage: python tools/rvq_bench.py [n_vec=6000] [Q=16] [reps=5]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import unified_audio_amd as qa  # noqa: E402


def main():
    n_vec = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
    Q = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    dev = torch.device("cuda:0")
    print(json.dumps(bench.rvq_bench(dev, qa.load_library(), n_vec, Q, reps=reps)))


if __name__ == "__main__":
    main()
