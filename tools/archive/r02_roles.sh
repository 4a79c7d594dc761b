#!/bin/bash
# per-role stand-alone kernel times (NHDFIT_ROLE_KERNELS=1), role windows inside the fused launch, and the fused bench
set -u
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/roles_$TAG
mkdir -p $OUT
cd $ROOT
B="python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-pmc --no-extras"
timeout 90 $B > $OUT/bench.json 2> $OUT/bench.err
python -c "import json; j=json.load(open('$OUT/bench.json')); print('fused', round(j['value']/1e12,3), 'T evals/s', round(j['ms_per_step']*1e3,1), 'us/step kernel', round(j['roofline']['kernel_ms']*1e3,1))" || tail -5 $OUT/bench.err
NHDFIT_ROLE_TIMES=100 timeout 60 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pmc --no-extras 2>&1 >/dev/null | grep nhdfit
cd /tmp && export TMPDIR=/tmp
NHDFIT_ROLE_KERNELS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-pmc --no-extras > $OUT/stats.log 2>&1
python - <<PY
import csv,glob
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_role" in r["Name"]: print("role", r["Name"].split("k_role<512, ")[1][0], "avg us", round(float(r["AverageNs"])/1e3,1), "calls", r["Calls"])
PY
