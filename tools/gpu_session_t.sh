#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02t
echo "default (nt weight / KV loads)" > gpurun_out/r02t/lm.log
timeout 300 python tools/lm_bench.py 16 4 >> gpurun_out/r02t/lm.log 2>&1
echo "cached loads" >> gpurun_out/r02t/lm.log
QA_LIBRARY=$GRAFT_REPO_ROOT/tools/_variants/lm_cached/libquarkaudio_hip.so timeout 300 python tools/lm_bench.py 16 4 >> gpurun_out/r02t/lm.log 2>&1
grep -v amdgpu.ids gpurun_out/r02t/lm.log
