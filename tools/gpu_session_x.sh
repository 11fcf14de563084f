#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02x
QA_LM_QKVATTN=1 timeout 900 python -m pytest tests/test_llm_gpu.py -x -q 2>&1 | tail -12 > gpurun_out/r02x/tests.log; cat gpurun_out/r02x/tests.log
for f in 0 1; do
  echo "QA_LM_QKVATTN=$f" >> gpurun_out/r02x/lm.log
  QA_LM_QKVATTN=$f timeout 300 python tools/lm_bench.py 16 3 2>&1 | grep -v amdgpu.ids | tail -2 >> gpurun_out/r02x/lm.log
done
cat gpurun_out/r02x/lm.log
