#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02w
timeout 900 python -m pytest tests/test_fuzz_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r02w/fuzz.log; cat gpurun_out/r02w/fuzz.log
