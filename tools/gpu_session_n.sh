#!/bin/bash
TAG=${1:-r02n}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_mimi_stream_gpu.py tests/test_ssl_gpu.py tests/test_ssl_golden_gpu.py tests/test_llm_gpu.py tests/test_hcodec_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | tail -12 > $O/tests.log
cat $O/tests.log
timeout 600 python bench.py --lean --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
QA_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/trs -o bench -- python $R/bench.py --steps 3 --warmup 1 --lean > $O/trace_serial.log 2>&1
python $R/tools/rocpd_stats.py /tmp/trs/bench_results.db $O/hcodec15_kernel_stats_serial.md
grep -n "attention\|seanet" $O/hcodec15_kernel_stats_serial.md | cut -c1-150
