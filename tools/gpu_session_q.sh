#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02q
( time timeout 900 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/r02q/smoke.log 2>&1
tail -5 gpurun_out/r02q/smoke.log
