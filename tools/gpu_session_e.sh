#!/bin/bash
TAG=${1:-r02g}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
QA_SERIAL=1 QA_GEMM_SHAPES=$O/gemm_shapes_serial.md timeout 300 python bench.py --steps 3 --warmup 1 --lean > $O/bench_serial.json 2> $O/bench_serial.err
QA_GEMM_SHAPES=$O/gemm_shapes.md timeout 300 python bench.py --steps 3 --warmup 1 --lean > $O/bench.json 2> $O/bench.err
cat $O/gemm_shapes_serial.md
