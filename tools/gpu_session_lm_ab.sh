#!/bin/bash
# LM decode A/B on the GPU box: default build vs tools/_variants/* (tools/variants.py), generate time + token checksums, then the LM suite
TAG=${1:-r02_lm_ab}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for v in default "$@"; do
  [ "$v" = "$TAG" ] && continue
  lib=$R/tools/_variants/$v/libquarkaudio_hip.so
  [ "$v" = default ] && lib=$R/unified_audio_amd/libquarkaudio_hip.so
  echo "== $v" | tee -a $O/ab.log
  QA_LIBRARY=$lib timeout 120 python tools/lm_bench.py 16 4 2>&1 | tail -5 | tee -a $O/ab.log
done
( time timeout 400 python -m pytest tests/test_llm_gpu.py -q -x 2>&1 | tail -5 ) 2>&1 | tee $O/lm_tests.log
