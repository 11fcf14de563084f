#!/bin/bash
TAG=${1:-r02m2}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for t in 0 384 100000; do
  echo "== QA_GEMM_BK16_MIN_TILES=$t" >> $O/gemm_small.log
  QA_BENCH_ONLY=small,mimi.out_proj,bt.lin2,enc.o QA_GEMM_BK16_MIN_TILES=$t timeout 300 python tools/gemm_bench.py >> $O/gemm_small.log 2>&1
done
cat $O/gemm_small.log
