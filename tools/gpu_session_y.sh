#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02y
( time timeout 1200 python tools/ubsan_host.py ) > gpurun_out/r02y/ubsan.log 2>&1; echo "rc=$?" >> gpurun_out/r02y/ubsan.log
grep -v "amdgpu.ids\|option-ignored\|^$" gpurun_out/r02y/ubsan.log | tail -25
