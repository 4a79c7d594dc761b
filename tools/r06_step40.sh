#!/bin/bash
# Round 6, last full GPU call on the shipped tree (after the LDS-room fix of k_decide): smoke, the whole GPU suite, the driver-form bench line, the full profile (kernel stats
# + PMC passes, summaries only).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_final5
mkdir -p $OUT
cd $ROOT
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2 | tee $OUT/smoke.log
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? seconds=$SECONDS" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|^FAILED|^ERROR|^E  " $OUT/pytest_gpu.log | tail -8
SECONDS=0
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err; echo "bench rc=$? seconds=$SECONDS"
python - <<PY $OUT/bench_driver_form.json
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("driver-form %.3f us  repeats med %.3f  steady %.3f  value %.3e" % (d["ms_per_step"]*1e3, d["repeats"]["ms_per_step_median"]*1e3, d["steady_state"]["ms_per_step_median"]*1e3, d["value"]))
print("roofline", json.dumps({k: v for k, v in d["roofline"].items() if not isinstance(v, (dict, str))}))
print("mode_b", d["mode_b"]["decisions_per_s"], "e2e", d["end_to_end"]["ms_per_call"], "single", d["single_find"]["ms_per_call_median"], "score_only", d.get("score_only", {}).get("ms_per_step"))
print("sched_loop", json.dumps(d.get("sched_loop"))[:700])
for o in d.get("other_configs", []): print({k: o[k] for k in ("config","nodes","pods","ms_per_step","find_ms_per_call","mode_b_decisions_per_s") if k in o})
PY
bash tools/gpu_profile.sh r06_final5 > $OUT/profile.log 2>&1
cp $ROOT/gpurun_out/prof_r06_final5/summary.txt $OUT/profile_summary.txt 2>/dev/null
cp $ROOT/gpurun_out/prof_r06_final5/kernel_stats.csv $OUT/rocprof_kernel_stats.csv 2>/dev/null
head -14 $OUT/profile_summary.txt | cut -c1-200
du -sh $ROOT/gpurun_out
