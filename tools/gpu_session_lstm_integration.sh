#!/bin/bash
# persistent LSTM in the product path: H-Codec 2.0 tests (default: persistent at d = 1536), every golden with the persistent kernel forced
# on at the other widths, then the 2.0 bench share with and without it
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r02_lstm_int}; mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 300 python -m pytest tests -q -m gpu -k "20" 2>&1 | tail -6 ) 2>&1 | tee $O/tests_20.log
( time QA_LSTM_PERSISTENT=1 timeout 300 python -m pytest tests/test_golden_gpu.py -q 2>&1 | tail -6 ) 2>&1 | tee $O/tests_forced.log
for mode in 0 auto; do
  if [ $mode = auto ]; then unset QA_LSTM_PERSISTENT; else export QA_LSTM_PERSISTENT=$mode; fi
  echo "== QA_LSTM_PERSISTENT=$mode" | tee -a $O/bench20.log
  timeout 200 python bench.py --model 2.0 --batch 16 --seconds 30 --lean --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" | tee -a $O/bench20.log
done
