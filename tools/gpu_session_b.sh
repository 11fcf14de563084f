#!/bin/bash
# LM iteration: correctness (LM GPU tests), timing, kernel trace.  usage: gpu_session_b.sh <tag>
TAG=${1:-r02b}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_llm_gpu.py -q --timeout 300 -x > $O/pytest_lm.log 2>&1
tail -4 $O/pytest_lm.log
for v in "QA_NONE=1" "QA_LM_GRAPH=0"; do
  echo "== $v" >> $O/lm_ab.log
  env $v timeout 120 python tools/lm_bench.py 16 3 >> $O/lm_ab.log 2>&1
done
grep -v amdgpu.ids $O/lm_ab.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/trlm -o lm -- python $R/tools/lm_bench.py 16 2 > $O/lm_trace.log 2>&1
python $R/tools/rocpd_stats.py /tmp/trlm/lm_results.db $O/lm_kernel_stats.md >> $O/lm_trace.log 2>&1
head -20 $O/lm_kernel_stats.md | cut -c1-150
