#!/bin/bash
# codec kernel trace (product mode + serial) of the lean bench
TAG=${1:-r02f}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --lean"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tr -o bench -- $CMD > $O/trace.log 2>&1
python $R/tools/rocpd_stats.py /tmp/tr/bench_results.db $O/kernel_stats.md
QA_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/trs -o bench -- $CMD > $O/trace_serial.log 2>&1
python $R/tools/rocpd_stats.py /tmp/trs/bench_results.db $O/kernel_stats_serial.md
head -45 $O/kernel_stats_serial.md | cut -c1-170
