#!/bin/bash
TAG=${1:-r02h}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_boundary_gpu.py tests/test_dist_gpu.py tests/test_hcodec_gpu.py tests/test_golden_gpu.py -q --timeout 300 > $O/pytest.log 2>&1
tail -40 $O/pytest.log
