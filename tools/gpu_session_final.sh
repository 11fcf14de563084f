#!/bin/bash
# round-end evidence: full GPU suite, default bench line, rocprofv3 / PMC summaries (tools/collect_profiles.sh)
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}_final; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 ) > $O/tests.log 2>&1
cat $O/tests.log
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.log
tail -3 $O/bench_time.log
python -c "
import json; d=json.load(open('$O/bench.json')); lm=d['unise_lm']
print(d['value'], d['ms_per_step'], d['roofline']['frac']); print(lm['value'], lm.get('end_to_end_b16')); print(lm.get('bicodec_detokenize')); print(d.get('extras'))"
bash tools/collect_profiles.sh $TAG > $O/collect.log 2>&1
tail -3 $O/collect.log
