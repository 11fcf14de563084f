#!/bin/bash
TAG=${1:-r02j}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/time.log
tail -3 $O/time.log; tail -5 $O/bench.err; cut -c1-600 $O/bench.json
