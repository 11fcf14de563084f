#!/bin/bash
# RVQ search at the configs[4] shape: kernel trace + MFMA-busy counters (separate pass, --pmc with --kernel-trace only)
TAG=${1:-r02_rvq}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 6000 48000; do
  timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/tr$n -o rvq -- python $R/tools/rvq_bench.py $n 16 3 > $O/trace_$n.log 2>&1
  python $R/tools/rocpd_stats.py /tmp/tr$n/rvq_results.db $O/rvq_${n}x16_kernel_stats.md
  timeout 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc$n -o b -- python $R/tools/rvq_bench.py $n 16 3 > $O/pmc_$n.log 2>&1
  python $R/tools/pmc_summary.py $O/rvq_${n}x16_pmc_summary /tmp/pmc$n > $O/pmc_summary_$n.log 2>&1
done
tail -2 $O/trace_6000.log $O/trace_48000.log
cat $O/rvq_6000x16_pmc_summary.md $O/rvq_48000x16_pmc_summary.md
