#!/bin/bash
# verification pass on the GPU box: full GPU suite, then the default bench line (no profiler)
TAG=${1:-r02v}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
( time timeout 700 python -m pytest tests -q -m gpu 2>&1 | tail -15 ) > $O/tests.log 2>&1
cat $O/tests.log
( time timeout 600 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.log
tail -3 $O/bench_time.log
python -c "
import json; d=json.load(open('$O/bench.json')); lm=d['unise_lm']
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('isolated')); print(lm['value'], lm.get('end_to_end_b16')); print(json.dumps(d.get('extras'), indent=0)[:3000])"
