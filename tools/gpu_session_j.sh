#!/bin/bash
TAG=${1:-r02m}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
python - <<PY > $O/infer.log 2>&1
import torch, sys
sys.path.insert(0, "$R")
from unified_audio_amd import audio_io, synth
audio_io.write_wav("/tmp/a.wav", synth.synth_wav(1, 1, 16000 * 7)[0], 16000)
audio_io.write_wav("/tmp/b.wav", synth.synth_wav(2, 1, 44100 * 3)[0], 44100)
PY
timeout 300 python tools/unise_infer.py --mode se --synthetic --out /tmp/out_se /tmp/a.wav /tmp/b.wav >> $O/infer.log 2>&1
timeout 300 python tools/unise_infer.py --mode ss --synthetic --out /tmp/out_ss /tmp/a.wav >> $O/infer.log 2>&1
ls -la /tmp/out_se /tmp/out_ss >> $O/infer.log 2>&1
tail -8 $O/infer.log
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/time.log
tail -3 $O/time.log
python -c "
import json; d=json.load(open('$O/bench.json')); lm=d['unise_lm']
print(d['value'], d['ms_per_step']); print(lm['value'], lm.get('end_to_end_b16')); print(lm.get('bicodec_detokenize'))"
cd /tmp && export TMPDIR=/tmp
cat > /tmp/bic.py <<PY
import sys, torch
sys.path.insert(0, "$R")
import unified_audio_amd as qa
from unified_audio_amd import synth
dev = torch.device("cuda:0")
m = qa.BiCodec(device=dev).load_state_dict(synth.bicodec_state_dict(77))
sem, glob = synth.bicodec_tokens(78, 16, 250)
for _ in range(3): w = m.detokenize(sem.to(dev), glob.to(dev))
torch.cuda.synchronize()
PY
QA_SERIAL=1 QA_GEMM_SHAPES=$O/bicodec_gemm_shapes.md timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/trb -o bic -- python /tmp/bic.py > $O/bic_trace.log 2>&1
python $R/tools/rocpd_stats.py /tmp/trb/bic_results.db $O/bicodec_kernel_stats.md
head -16 $O/bicodec_kernel_stats.md | cut -c1-150
