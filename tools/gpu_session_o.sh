#!/bin/bash
TAG=${1:-r02o}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -6 > $O/tests.log
cat $O/tests.log
for t in 0 1; do
  echo "== QA_GEMM_LINEAR=$t" >> $O/gemm.log
  QA_BENCH_ONLY=cal.16384,mimi,bt,convnext,dec.lstm,dec.qkv,dec.w2,enc.lstm,enc.o,small QA_GEMM_LINEAR=$t timeout 300 python tools/gemm_bench.py >> $O/gemm.log 2>&1
done
cat $O/gemm.log
for t in 0 1; do
QA_GEMM_LINEAR=$t timeout 600 python bench.py --lean --steps 10 --warmup 2 > $O/bench$t.json 2> $O/bench$t.err
python -c "
import json; d=json.load(open('$O/bench$t.json')); print('linear=$t', d['value'], d['ms_per_step'])"
done
