#!/bin/bash
# ONE parameterised GPU-box session (replaces the per-experiment gpu_session_*.sh scripts of rounds 1-2).
#   usage:  gpurun --timeout N -- 'bash tools/gpu_session.sh TAG STEP [STEP ...]'
# Every step writes into gpurun_out/TAG/ (merged back by gpurun); summaries worth keeping are copied to profiles/ by hand.
# Steps:
#   tests[:EXPR]      pytest -m gpu (optionally -k EXPR, or a file path when EXPR contains ".py")
#   bench[:ARGS]      python bench.py ARGS (default: the driver's default command) -> bench.json
#   lm[:B[:REPS]]     tools/lm_bench.py (decode-step time of the UniSE LM, knobs from the environment)
#   ab:KNOB=a,b[,c]:CMD   run CMD once per value of an environment knob, logs side by side (CMD: lm | bench-lean | hc10 | hc20)
#   trace:NAME:CMD    rocprofv3 --kernel-trace --stats of CMD (lm | bench-lean | bench-serial | hc10 | hc20) -> NAME_kernel_stats.md
#   pmc:NAME:CMD      three separate --pmc passes (FETCH_SIZE | WRITE_SIZE | MFMA busy) of CMD -> NAME_pmc_summary.{md,json}
#   profiles          the round's standard evidence set (tools/collect_profiles.sh TAG)
#   kab:ARGS          tools/knob_ab.py ARGS - in-process knob A/B (and --libs per-build A/B) of the codec hot path -> knob_ab.txt
#   run:CMD           any shell command (log tail kept in run.log)
#   smoke             __graft_entry__.smoke()
# r04 lesson (25 GPU-minutes lost): `ab:...:python tools/x.py 32 2` used to reach cmd_of UNQUOTED, so only the word `python` came back
# and three interactive interpreters sat on stdin until their time-outs - commands are now passed quoted, stdin is /dev/null, and
# `DRY=1 bash tools/gpu_session.sh TAG STEP...` prints what each step would run (check every new session line with it first).
TAG=${1:?tag}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
cmd_of() {
  case "$1" in
    lm) echo "python $R/tools/lm_bench.py 16 2" ;;
    lm-short) echo "python $R/tools/lm_bench.py 16 1 24" ;;
    bench-lean) echo "python $R/bench.py --steps 3 --warmup 1 --lean" ;;
    bench-serial) echo "env QA_SERIAL=1 python $R/bench.py --steps 3 --warmup 1 --lean" ;;
    hc10) echo "python $R/bench.py --steps 5 --warmup 2 --lean --model 1.0" ;;
    hc20) echo "python $R/bench.py --steps 2 --warmup 1 --lean --model 2.0 --batch 16 --seconds 30" ;;
    *) echo "$1" ;;
  esac
}
if [ -n "$DRY" ]; then  # print the command of every step instead of running it
  for step in "$@"; do
    kind=${step%%:*}; rest=${step#*:}; [ "$rest" = "$step" ] && rest=""
    case "$kind" in
      ab) IFS=: read -r kv what <<< "$rest"; echo "[$kind] for ${kv#*=} of ${kv%%=*}: $(cmd_of "$what")" ;;
      trace|pmc) IFS=: read -r name what <<< "$rest"; echo "[$kind $name] $(cmd_of "$what")" ;;
      kab) echo "[kab] python tools/knob_ab.py $rest" ;;
      run) echo "[run] $rest" ;;
      *) echo "[$kind] $rest" ;;
    esac
  done
  exit 0
fi
for step in "$@"; do
  kind=${step%%:*}; rest=${step#*:}; [ "$rest" = "$step" ] && rest=""
  echo "=== $step" | tee -a $O/session.log
  case "$kind" in
    tests)
      case "$rest" in
        "") ( time timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25 ) 2>&1 | tee -a $O/tests.log ;;
        *.py*) ( time timeout 1500 python -m pytest $rest -q -m gpu 2>&1 | tail -25 ) 2>&1 | tee -a $O/tests.log ;;
        *) ( time timeout 1500 python -m pytest tests -q -m gpu -k "$rest" 2>&1 | tail -25 ) 2>&1 | tee -a $O/tests.log ;;
      esac ;;
    bench)
      ( time timeout 900 python bench.py $rest > $O/bench.json 2> $O/bench.err ) 2>&1 | tail -3 | tee -a $O/session.log
      python - <<PY 2>&1 | tee -a $O/session.log
import json
d = json.load(open("$O/bench.json")); lm = d.get("unise_lm") or {}
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "isolated", (d["roofline"].get("isolated") or {}).get("frac"))
print("lm", lm.get("value"), lm.get("ms_per_decode_step"), (lm.get("roofline") or {}).get("frac"), "e2e", lm.get("end_to_end_b16"))
for k, v in (d.get("extras") or {}).items(): print(k, v.get("value", v.get("error")), v.get("ms_per_step"))
PY
      ;;
    lm) IFS=: read -r b reps <<< "$rest"; timeout 300 python tools/lm_bench.py ${b:-16} ${reps:-3} 2>&1 | tail -6 | tee -a $O/lm.log ;;
    ab)
      IFS=: read -r kv what <<< "$rest"; k=${kv%%=*}; vals=${kv#*=}
      for v in ${vals//,/ }; do
        echo "--- $k=$v  ($what)" | tee -a $O/ab.log
        env $k=$v timeout ${STEP_TIMEOUT:-300} $(cmd_of "$what") < /dev/null 2>&1 | tail -4 | cut -c1-600 | tee -a $O/ab.log
      done ;;
    trace)
      IFS=: read -r name what <<< "$rest"
      ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_$name -o t -- $(cmd_of "$what") < /dev/null > $O/${name}_trace.log 2>&1 )
      python tools/rocpd_stats.py /tmp/tr_$name/t_results.db $O/${name}_kernel_stats.md 2>&1 | tail -2 ;;
    pmc)
      IFS=: read -r name what <<< "$rest"; i=0
      for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
        i=$((i+1))
        ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_${name}_$i -o b -- $(cmd_of "$what") < /dev/null > $O/${name}_pmc$i.log 2>&1 )
      done
      python tools/pmc_summary.py $O/${name}_pmc_summary /tmp/pmc_${name}_1 /tmp/pmc_${name}_2 /tmp/pmc_${name}_3 2>&1 | tail -3 ;;
    kab) timeout ${STEP_TIMEOUT:-400} python tools/knob_ab.py $rest < /dev/null 2>&1 | grep -v amdgpu.ids | tee -a $O/knob_ab.txt ;;
    run) echo "$rest" >> $O/run.log; ( timeout ${STEP_TIMEOUT:-600} bash -c "$rest" < /dev/null 2>&1 | tail -${TAIL:-60} ) | tee -a $O/run.log ;;
    profiles) bash tools/collect_profiles.sh $TAG ;;
    smoke) ( time timeout 900 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8 ) 2>&1 | tee -a $O/smoke.log ;;
    *) echo "unknown step $step" ;;
  esac
done
