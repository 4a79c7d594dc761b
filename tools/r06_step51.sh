#!/bin/bash
# Round 6, GPU call: the whole GPU suite on the tree whose records carry the C row (per-test time limit), then the driver-form bench line.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step51
mkdir -p $OUT
cd $ROOT
timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -2 | tee $OUT/smoke.log
SECONDS=0
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider -x --timeout=300 --durations=5 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? seconds=$SECONDS $(grep -E 'passed|failed' $OUT/pytest_gpu.log | tail -1)"
grep -E "^FAILED|^ERROR|Timeout" $OUT/pytest_gpu.log | head -5
SECONDS=0
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err; echo "bench rc=$? seconds=$SECONDS"
python - <<PY $OUT/bench_driver_form.json
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("driver-form %.3f us  repeats med %.3f  steady %.3f  value %.3e" % (d["ms_per_step"]*1e3, d["repeats"]["ms_per_step_median"]*1e3, d["steady_state"]["ms_per_step_median"]*1e3, d["value"]))
print("roofline", json.dumps({k: v for k, v in d["roofline"].items() if not isinstance(v, (dict, str))}))
print("mode_b", d["mode_b"]["decisions_per_s"], d["mode_b"].get("parity"), "e2e", d["end_to_end"]["ms_per_call"], "single", d["single_find"]["ms_per_call_median"])
for o in d.get("other_configs", []): print({k: o[k] for k in ("config","nodes","pods","ms_per_step","find_ms_per_call","mode_b_decisions_per_s") if k in o})
PY
