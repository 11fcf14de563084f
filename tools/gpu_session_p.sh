#!/bin/bash
# kernel traces of the other two codec versions (serial mode): where H-Codec 2.0 (configs[4] share) and 1.0 spend their step
TAG=${1:-r02p}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
QA_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/t20 -o b -- python $R/bench.py --model 2.0 --batch 16 --seconds 30 --steps 2 --warmup 1 --lean > $O/trace20.log 2>&1
python $R/tools/rocpd_stats.py /tmp/t20/b_results.db $O/hcodec20_kernel_stats_serial.md
head -20 $O/hcodec20_kernel_stats_serial.md | cut -c1-170
QA_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/t10 -o b -- python $R/bench.py --model 1.0 --steps 3 --warmup 1 --lean > $O/trace10.log 2>&1
python $R/tools/rocpd_stats.py /tmp/t10/b_results.db $O/hcodec10_kernel_stats_serial.md
head -16 $O/hcodec10_kernel_stats_serial.md | cut -c1-170
cd $R
timeout 600 python bench.py --model 1.0 --lean --steps 10 --warmup 2 > $O/bench10.json 2> $O/bench10.err
python -c "
import json; d=json.load(open('$O/bench10.json')); print('1.0', d['value'], d['ms_per_step'])"
