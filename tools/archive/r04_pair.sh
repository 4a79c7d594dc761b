#!/bin/bash
# Round 4: the pair form of the fit role's sweep (fit_core.h "pair rows") - parity first, then the step time with the tables
# off / C only / C + XX (tuning build: NHDFIT_PAIR=0|1|unset), then the driver-form bench line of the ship build.
#   gpurun -- bash tools/r04_pair.sh [tag]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-a}
OUT=$ROOT/gpurun_out/r04_pair_$TAG
mkdir -p $OUT
cd $ROOT
SECONDS=0
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$? seconds=$SECONDS" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error|Error|assert" $OUT/pytest_gpu.log | tail -8
TL=$ROOT/nhd_amd/libnhdfit_tuning.so
B="--no-pmc --no-extras --no-cpu-baseline --steps 1000 --warmup 200"
{
for pair in 0 1 2; do
  for shape in "--config 4 --nodes-per-gpu 65536 --pods 4096" "--config 5 --nodes-per-gpu 32768 --pods 16384" "--config 2 --nodes-per-gpu 4096 --pods 256"; do
    echo "== NHDFIT_PAIR=$pair $shape"
    NHDFIT_LIBRARY=$TL NHDFIT_PAIR=$pair timeout 300 python bench.py $B $shape 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['steady_state']
print('ms_per_step', round(d['ms_per_step'],5), 'cold', round(d['cold_start']['ms_per_step'],5), 'steady min/med/max', round(s['ms_per_step_min'],5), round(s['ms_per_step_median'],5), round(s['ms_per_step_max'],5), 'kernel_ms', round(d['roofline']['kernel_ms'],5), 'lds', d['config']['lds_bytes_per_block'], 'placed', d['placed_pods'])"
  done
done
} 2>&1 | tee $OUT/pair_ab.log
echo "seconds=$SECONDS"
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err
echo "bench rc=$? seconds=$SECONDS"
python - <<'PY' $OUT/bench_driver_form.json
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "cold", d["cold_start"], "steady", d["steady_state"]["ms_per_step_median"])
r=d["roofline"]; print("frac", r["frac"], "lds", r.get("lds"), "hbm", r.get("hbm_counter"), "issue", r.get("issue"))
print("mode_b", {k:v for k,v in d["mode_b"].items() if k!="parity"}, d["mode_b"]["parity"]["identical"])
print("single", d["single_find"]["ms_per_call_median"], "e2e", d["end_to_end"]["ms_per_call"])
for o in d["other_configs"]: print({k:(v if not isinstance(v,dict) else v.get("identical")) for k,v in o.items()})
PY
