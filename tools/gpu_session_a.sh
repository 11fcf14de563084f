#!/bin/bash
# GPU session A (round 2): layout probe, LM tests first (new fused decode), then the full GPU suite, LM A/B timings, bench.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02a; mkdir -p $O
cd $R
./tools/micro/mfma4_layout > $O/mfma4.log 2>&1
timeout 600 python -m pytest tests/test_llm_gpu.py -q --timeout 300 > $O/pytest_lm.log 2>&1
echo "lm tests rc=$?" >> $O/pytest_lm.log
QA_LM_MFMA16=1 timeout 300 python -m pytest tests/test_llm_gpu.py -q --timeout 300 -k "golden or narrow or small_lm" > $O/pytest_lm_mfma16.log 2>&1
for v in "QA_NONE=1" "QA_LM_GRAPH=0" "QA_LM_MFMA16=1" "QA_LM_UNFUSED=1"; do
  echo "== $v" >> $O/lm_ab.log
  env $v timeout 120 python tools/lm_bench.py 16 3 >> $O/lm_ab.log 2>&1
done
env timeout 120 python tools/lm_bench.py 32 2 >> $O/lm_ab.log 2>&1
env timeout 120 python tools/lm_bench.py 8 2 >> $O/lm_ab.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --deselect tests/test_llm_gpu.py > $O/pytest_all.log 2>&1
echo "all tests rc=$?" >> $O/pytest_all.log
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
QA_LSTM_GRAPH=0 timeout 300 python bench.py --steps 5 --warmup 2 --lean > $O/bench_lstm_eager.json 2> $O/bench_lstm_eager.err
QA_RVQ_LEGACY=1 timeout 300 python bench.py --steps 5 --warmup 2 --lean > $O/bench_rvq_legacy.json 2> $O/bench_rvq_legacy.err
tail -3 $O/pytest_lm.log $O/pytest_all.log; cat $O/mfma4.log; cat $O/lm_ab.log | grep -v Warn | tail -30; cut -c1-400 $O/bench.json
