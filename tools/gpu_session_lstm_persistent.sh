#!/bin/bash
# persistent LSTM recurrence vs one launch per step (tools/micro/lstm_persistent.hip), every run under its own timeout
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r02_lstm_persistent}; mkdir -p $O
cd $GRAFT_REPO_ROOT
for cfg in "1536 16 500" "1024 32 500" "768 32 500" "512 32 500"; do
  echo "== $cfg" | tee -a $O/log.txt
  timeout 60 ./tools/micro/lstm_persistent $cfg 2>&1 | tee -a $O/log.txt
  echo "rc=$?" | tee -a $O/log.txt
done
