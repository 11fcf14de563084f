#!/bin/bash
# round 2, session k: causal variant + mimi streaming on the GPU
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mimi_stream_gpu.py tests/test_hcodec_gpu.py tests/test_golden_gpu.py tests/test_kernels_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/r02k_tests.log
cat gpurun_out/r02k_tests.log
