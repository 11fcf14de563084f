#!/bin/bash
TAG=${1:-r02l2}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hcodec_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | tail -5 > $O/tests.log
cat $O/tests.log
cd /tmp && export TMPDIR=/tmp
QA_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/trs -o bench -- python $R/bench.py --steps 3 --warmup 1 --lean > $O/trace_serial.log 2>&1
python $R/tools/rocpd_stats.py /tmp/trs/bench_results.db $O/hcodec15_kernel_stats_serial.md
grep -n "seanet" $O/hcodec15_kernel_stats_serial.md | cut -c1-170
