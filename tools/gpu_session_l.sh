#!/bin/bash
# round 2, session l: fused SEANet front kernel - parity, fuzz, A/B timing (QA_SEANET_FUSED=0/1), kernel trace
TAG=${1:-r02l}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hcodec_gpu.py tests/test_golden_gpu.py tests/test_fuzz_gpu.py tests/test_properties_gpu.py -x -q 2>&1 | tail -15 > $O/tests.log
cat $O/tests.log
for f in 1 0; do
  QA_SEANET_FUSED=$f timeout 600 python bench.py --lean --steps 10 --warmup 2 > $O/bench_fused$f.json 2> $O/bench_fused$f.err
  python -c "
import json; d=json.load(open('$O/bench_fused$f.json')); print('fused=$f', d['value'], d['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
QA_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/trs -o bench -- python $R/bench.py --steps 3 --warmup 1 --lean > $O/trace_serial.log 2>&1
python $R/tools/rocpd_stats.py /tmp/trs/bench_results.db $O/hcodec15_kernel_stats_serial.md
head -16 $O/hcodec15_kernel_stats_serial.md | cut -c1-150
grep -n "seanet\|conv_in\|128, 32" $O/hcodec15_kernel_stats_serial.md | cut -c1-170
