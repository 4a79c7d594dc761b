#!/bin/bash
# Round 6, GPU call: smoke and the driver-form bench line on the tree as shipped (bench.py's CPU legs size their teams by the usable cores).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step53
mkdir -p $OUT
cd $ROOT
timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -2 | tee $OUT/smoke.log
SECONDS=0
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err; echo "bench rc=$? seconds=$SECONDS"
python - <<PY $OUT/bench_driver_form.json
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("driver-form %.3f us  repeats med %.3f  steady %.3f  value %.3e" % (d["ms_per_step"]*1e3, d["repeats"]["ms_per_step_median"]*1e3, d["steady_state"]["ms_per_step_median"]*1e3, d["value"]))
print("roofline", json.dumps({k: v for k, v in d["roofline"].items() if not isinstance(v, (dict, str))})[:600])
print("cpu_baseline", json.dumps({k: v for k, v in d["cpu_baseline"].items() if not isinstance(v, (dict, str))}), d["cpu_baseline"].get("all_host_cores"))
print("mode_b", d["mode_b"]["decisions_per_s"], d["mode_b"]["parity"]["identical"])
PY
