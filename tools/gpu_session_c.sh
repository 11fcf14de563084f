#!/bin/bash
TAG=${1:-r02e}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for v in "QA_NONE=1" "QA_LM_NT_DOWN=8" "QA_LM_NT_DOWN=16" "QA_LM_NT_O=8" "QA_LM_NT_O=16" "QA_LM_NT_QKV=16" "QA_LM_NT_QKV=4" "QA_LM_NT_GU=8" "QA_LM_MFMA16=1"; do
  echo "== $v" >> $O/lm_ab.log
  env $v timeout 120 python tools/lm_bench.py 16 3 2>&1 | tail -1 >> $O/lm_ab.log
done
cat $O/lm_ab.log
