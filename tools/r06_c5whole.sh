#!/bin/bash
# Round 6, first GPU call: BASELINE config 5's WHOLE cluster (262 144 nodes) on one device - the new parity tests, the bench line
# of that shape with its kernel stats, and the default (driver-form) line of the tree for reference.
#   gpurun --timeout 1500 -- bash tools/r06_c5whole.sh [tag]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-a}
OUT=$ROOT/gpurun_out/r06_c5whole_$TAG
mkdir -p $OUT
cd $ROOT
nproc; free -g | head -2
SECONDS=0
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x --durations=8 -k "262144 or config5_whole" > $OUT/pytest_c5whole.log 2>&1
echo "pytest rc=$? seconds=$SECONDS" | tee -a $OUT/pytest_c5whole.log
grep -E "passed|failed|^FAILED|^ERROR|^E  |s call" $OUT/pytest_c5whole.log | tail -20
SECONDS=0
timeout 600 python bench.py --gpus 1 --config 5 --total-nodes 262144 --pods 16384 --steps 100 --warmup 20 --cpu-sample-pods 256 --no-extras > $OUT/bench_c5whole.json 2> $OUT/bench_c5whole.err
echo "bench c5 rc=$? seconds=$SECONDS" | tee -a $OUT/bench_c5whole.err; tail -3 $OUT/bench_c5whole.err
python - <<'PY' $OUT/bench_c5whole.json
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value", d["value"], "ms_per_step", d["ms_per_step"], "steady", (d.get("steady_state") or {}).get("ms_per_step_median"), "placed", d["placed_pods"])
    r=d["roofline"]; print({k:v for k,v in r.items() if not isinstance(v,(str,dict))})
    print("mode_b", {k:v for k,v in d.get("mode_b",{}).items() if k!="parity"}, (d.get("mode_b",{}).get("parity") or {}).get("identical"))
    print("e2e", d.get("end_to_end"))
    print("cpu", {k:v for k,v in d.get("cpu_baseline",{}).items() if k!="reference"})
except Exception as e:
    print("no line:", e)
PY
SECONDS=0
timeout 800 python bench.py --gpus 1 --config 5 --total-nodes 262144 --pods 16384 --steps 100 --warmup 20 --no-cpu-baseline --no-pmc > $OUT/bench_c5whole_extras.json 2> $OUT/bench_c5whole_extras.err
echo "bench c5 extras rc=$? seconds=$SECONDS" | tee -a $OUT/bench_c5whole_extras.err; tail -3 $OUT/bench_c5whole_extras.err
python - <<'PY' $OUT/bench_c5whole_extras.json
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("mode_b", {k:v for k,v in d.get("mode_b",{}).items() if k!="parity"}, (d.get("mode_b",{}).get("parity") or {}))
    print("e2e", d.get("end_to_end")); print("single", d.get("single_find")); print("score_only", d.get("score_only"))
except Exception as e:
    print("no line:", e)
PY
ST=$OUT/stats; rm -rf $ST; mkdir -p $ST
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ST -o s -- python $ROOT/bench.py --gpus 1 --config 5 --total-nodes 262144 --pods 16384 --steps 100 --warmup 20 --no-cpu-baseline --no-pmc --no-extras > $ST/run.log 2>&1)
find $ST -name "*kernel_stats.csv" -exec cp {} $OUT/c5whole_kernel_stats.csv \;
head -8 $OUT/c5whole_kernel_stats.csv | cut -c1-200
rm -rf $ST
SECONDS=0; timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err; echo "bench rc=$? seconds=$SECONDS" | tee -a $OUT/bench_driver_form.err
python - <<'PY' $OUT/bench_driver_form.json
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "cold", d["cold_start"]["ms_per_step"], "steady", d["steady_state"]["ms_per_step_median"], "repeats", d["repeats"])
print("mode_b", {k:v for k,v in d["mode_b"].items() if k!="parity"}, d["mode_b"]["parity"]["identical"])
print("single", d["single_find"]["ms_per_call_median"], "e2e", d["end_to_end"]["ms_per_call"], "score_only", d["score_only"]["ms_per_step"])
for o in d["other_configs"]: print({k:(v if not isinstance(v,dict) else v.get("identical")) for k,v in o.items()})
PY
