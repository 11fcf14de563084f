#!/bin/bash
# Round-end evidence on the GPU box: rocprofv3 kernel trace of the bench command (product mode and QA_SERIAL=1), three PMC passes
# (FETCH_SIZE | WRITE_SIZE | MFMA busy), summaries into gpurun_out/$1/ (raw traces stay in /tmp).
TAG=${1:-r01g}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --lean"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/tr -o bench -- $CMD > $O/trace.log 2>&1
python $R/tools/rocpd_stats.py /tmp/tr/bench_results.db $O/kernel_stats.md
QA_SERIAL=1 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/trs -o bench -- $CMD > $O/trace_serial.log 2>&1
python $R/tools/rocpd_stats.py /tmp/trs/bench_results.db $O/kernel_stats_serial.md
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  QA_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc$i -o b -- python $R/bench.py --steps 2 --warmup 1 --lean > $O/pmc$i.log 2>&1
done
python $R/tools/pmc_summary.py $O/pmc_summary /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 > $O/pmc_summary.log 2>&1
tail -3 $O/pmc_summary.log
