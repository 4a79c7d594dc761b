#!/bin/bash
# Round 4 full GPU check: smoke, parity suite, the bench line as the driver asks for it (--steps 20 --warmup 5, counters and extras
# on), rocprofv3 kernel stats + one PMC pass per counter set of the benchmarked command -> gpurun_out/r04_full_<tag> (what is to be
# judged is copied into profiles/r04).   gpurun -- bash tools/r04_full.sh [tag] [ab]
#   ab: also time the 20-step region with the pipeline drained by role launches (NHDFIT_ROLE_DRAIN=1, tuning build) and in one launch
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-a}
AB=${2:-}
OUT=$ROOT/gpurun_out/r04_full_$TAG
mkdir -p $OUT
cd $ROOT
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
SECONDS=0
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? seconds=$SECONDS" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error|Error|assert" $OUT/pytest_gpu.log | tail -6
SECONDS=0; timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err; echo "bench rc=$? seconds=$SECONDS" | tee -a $OUT/bench_driver_form.err
python - <<'PY' $OUT/bench_driver_form.json
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "cold", d["cold_start"]["ms_per_step"], "steady", d["steady_state"]["ms_per_step_median"], "repeats", d["repeats"])
r=d["roofline"]; print("frac", r["frac"], "kernel_ms", r["kernel_ms"], "unit_fracs", r.get("unit_fracs"), "lds", r.get("lds"))
print("mode_b", {k:v for k,v in d["mode_b"].items() if k!="parity"}, d["mode_b"]["parity"]["identical"])
print("single", d["single_find"]["ms_per_call_median"], "e2e", d["end_to_end"]["ms_per_call"], "score_only", d["score_only"])
for o in d["other_configs"]: print({k:(v if not isinstance(v,dict) else v.get("identical")) for k,v in o.items()})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("python_restatement"))
PY
ST=$OUT/stats; rm -rf $ST; mkdir -p $ST
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ST -o s -- python $ROOT/bench.py --steps 400 --warmup 400 --no-cpu-baseline --no-pmc --no-extras > $ST/run.log 2>&1)
find $ST -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -8 $OUT/kernel_stats.csv | cut -c1-200
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES" "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  D=$OUT/pmc_$name; rm -rf $D; mkdir -p $D
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $D -o p -- python $ROOT/bench.py --steps 60 --warmup 20 --no-settle --no-cpu-baseline --no-pmc --no-extras > $D/run.log 2>&1)
done
python tools/summarize_profile.py $OUT > $OUT/pmc_summary.txt 2>&1 || true
tail -30 $OUT/pmc_summary.txt
# the per-counter directories hold every dispatch: keep the summaries only (gpurun_out is capped)
rm -rf $OUT/pmc_*/ $ST
if [ -n "$AB" ]; then
  TL=$ROOT/nhd_amd/libnhdfit_tuning.so
  B="--no-pmc --no-extras --no-cpu-baseline --steps 20 --warmup 5"
  {
  for k in 1 2 3; do
    for drain in role fused; do
      if [ $drain = role ]; then export NHDFIT_ROLE_DRAIN=1; else unset NHDFIT_ROLE_DRAIN; fi
      echo -n "== drain=$drain: "
      NHDFIT_LIBRARY=$TL timeout 300 python bench.py $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('ms_per_step', round(d['ms_per_step'],5), 'cold', round(d['cold_start']['ms_per_step'],5), 'repeats', d['repeats']['ms_per_step_min'], d['repeats']['ms_per_step_median'], 'steady', round(d['steady_state']['ms_per_step_median'],5))"
    done
  done
  } 2>&1 | tee $OUT/drain_ab.log
  timeout 200 python tools/time_single_find.py | tee $OUT/single_find_latency.json | tail -3
fi
