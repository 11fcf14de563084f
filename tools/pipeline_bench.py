#!/usr/bin/env python
"""UniSE end to end, sequential vs pipelined driver (bench.py's unise_pipeline_bench alone).  usage: pipeline_bench.py [batches] [segments]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

r = bench.unise_pipeline_bench(torch.device("cuda:0"), int(sys.argv[1]) if len(sys.argv) > 1 else 6, int(sys.argv[2]) if len(sys.argv) > 2 else 16,
                               lm_graph=(sys.argv[3] != "0") if len(sys.argv) > 3 else True)
print(json.dumps(r))
