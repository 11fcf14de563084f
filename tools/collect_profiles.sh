#!/bin/bash
# Round evidence on the GPU box: rocprofv3 kernel trace of the bench command (product mode and QA_SERIAL=1), three PMC passes
# (FETCH_SIZE | WRITE_SIZE | MFMA busy), the LM generate trace; summaries into gpurun_out/$1/ (raw traces stay in /tmp).
# The PMC summary json is stamped with the hash of conv_gemm.hip so that bench.py refuses to quote it for another build.
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --lean"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tr -o bench -- $CMD > $O/trace.log 2>&1
python $R/tools/rocpd_stats.py /tmp/tr/bench_results.db $O/hcodec15_kernel_stats.md
QA_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/trs -o bench -- $CMD > $O/trace_serial.log 2>&1
python $R/tools/rocpd_stats.py /tmp/trs/bench_results.db $O/hcodec15_kernel_stats_serial.md
QA_SERIAL=1 QA_GEMM_SHAPES=$O/hcodec15_gemm_shapes_serial.md timeout 300 $CMD > /dev/null 2>&1
if [ -z "$SKIP_LM" ]; then
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/trlm -o lm -- python $R/tools/lm_bench.py 16 2 > $O/lm_trace.log 2>&1
python $R/tools/rocpd_stats.py /tmp/trlm/lm_results.db $O/lm_kernel_stats.md
fi
# H-Codec 1.0 at the metric's batch (32 x 10 s) and H-Codec 2.0 at the per-GPU share of configs[4] (16 x 30 s @48 kHz), every kernel alone on the device
QA_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tr10 -o b -- python $R/bench.py --steps 5 --warmup 2 --lean --model 1.0 > $O/trace10.log 2>&1
python $R/tools/rocpd_stats.py /tmp/tr10/b_results.db $O/hcodec10_kernel_stats_serial.md
QA_SERIAL=1 timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/tr20 -o b -- python $R/bench.py --steps 2 --warmup 1 --lean --model 2.0 --batch 16 --seconds 30 > $O/trace20.log 2>&1
python $R/tools/rocpd_stats.py /tmp/tr20/b_results.db $O/hcodec20_kernel_stats_serial.md
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  QA_SERIAL=1 timeout 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc$i -o b -- python $R/bench.py --steps 2 --warmup 1 --lean > $O/pmc$i.log 2>&1
done
python $R/tools/pmc_summary.py $O/hcodec15_pmc_summary /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 > $O/pmc_summary.log 2>&1
python - <<PY
import hashlib, json
p = "$O/hcodec15_pmc_summary.json"
d = json.load(open(p))
d["_kernel_source_sha256_16"] = hashlib.sha256(open("$R/unified_audio_amd/csrc/conv_gemm.hip", "rb").read()).hexdigest()[:16]
d["_source"] = "profiles/${TAG}_hcodec15_pmc_summary.md"
json.dump(d, open(p, "w"), indent=1)
PY
tail -3 $O/pmc_summary.log
