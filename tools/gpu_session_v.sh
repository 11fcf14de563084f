#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02v
for l in 0 90000 140000; do
  echo "QA_LM_SPREAD_LDS=$l" >> gpurun_out/r02v/lm.log
  QA_LM_SPREAD_LDS=$l timeout 300 python tools/lm_bench.py 16 3 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r02v/lm.log
done
cat gpurun_out/r02v/lm.log
