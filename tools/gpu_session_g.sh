#!/bin/bash
TAG=${1:-r02i}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1
tail -5 $O/pytest.log
for v in "QA_NONE=1" "QA_LSTM_SPLIT=0" "QA_LSTM_GRAPH=0"; do
  env $v timeout 300 python bench.py --steps 5 --warmup 2 --lean 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],2))" >> $O/ab.log
done
cat $O/ab.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tr -o bench -- python $R/bench.py --steps 3 --warmup 1 --lean > $O/trace.log 2>&1
python $R/tools/rocpd_stats.py /tmp/tr/bench_results.db $O/kernel_stats.md
head -22 $O/kernel_stats.md | cut -c1-160
