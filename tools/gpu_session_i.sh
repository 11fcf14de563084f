#!/bin/bash
TAG=${1:-r02k}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_bicodec_gpu.py tests/test_unise_driver_gpu.py -q --timeout 600 > $O/pytest.log 2>&1
tail -30 $O/pytest.log | cut -c1-300
