#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02aa
timeout 900 python -m pytest tests/test_hcodec_gpu.py tests/test_golden_gpu.py -x -q -k "20" 2>&1 | tail -8 > gpurun_out/r02aa/tests.log; cat gpurun_out/r02aa/tests.log
