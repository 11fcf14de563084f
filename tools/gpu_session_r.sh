#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02r
timeout 600 python tools/split_bench.py 8 > gpurun_out/r02r/split2.log 2>&1; tail -2 gpurun_out/r02r/split2.log
QA_SPLIT_WAYS=4 timeout 600 python tools/split_bench.py 8 > gpurun_out/r02r/split4.log 2>&1; tail -2 gpurun_out/r02r/split4.log
