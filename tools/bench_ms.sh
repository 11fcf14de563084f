#!/bin/bash
# one line per run: value and ms_per_step of a lean bench.py pass (A/B sessions); usage: bench_ms.sh [bench.py args]
python /root/repo/bench.py --lean "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readlines()[-1])
print('value %.1f %s  ms_per_step %.3f  workload %s' % (d['value'], d['unit'], d['ms_per_step'], d['config']['workload'][:60]))"
