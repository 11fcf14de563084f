#!/usr/bin/env python
"""bench.py's unise_micro_batch_bench alone (A/B sessions)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

print(json.dumps(bench.unise_micro_batch_bench(torch.device("cuda:0")), indent=1))
