#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02u
QA_LIBRARY=$GRAFT_REPO_ROOT/tools/_variants/lm_timing/libquarkaudio_hip.so timeout 300 python tools/lm_timing.py > gpurun_out/r02u/lm_timing.log 2>&1
grep -v amdgpu.ids gpurun_out/r02u/lm_timing.log
