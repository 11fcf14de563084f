#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02z
( time timeout 900 python bench.py > gpurun_out/r02z/bench.json 2> gpurun_out/r02z/bench.err ) 2> gpurun_out/r02z/time.log
tail -3 gpurun_out/r02z/time.log
python -c "
import json; d=json.load(open('gpurun_out/r02z/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic']); print({k:(v.get('value'), v.get('ms_per_step'), v.get('error')) for k,v in d['extras'].items()})"
