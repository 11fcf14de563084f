#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02s
timeout 120 ./tools/micro/dep_chain > gpurun_out/r02s/dep_chain.log 2>&1; cat gpurun_out/r02s/dep_chain.log
