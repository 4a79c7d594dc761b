#!/bin/bash
# per-role stand-alone kernel times (NHDFIT_ROLE_KERNELS=1) + the fused bench
set -u
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/roles_$TAG
mkdir -p $OUT
cd $ROOT
timeout 90 python bench.py --steps 400 --warmup 20 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python -c "import json; j=json.load(open('$OUT/bench.json')); print('fused', round(j['value']/1e12,3), 'T evals/s', round(j['ms_per_step']*1e3,1), 'us/step kernel', round(j['roofline']['kernel_ms']*1e3,1))" || tail -5 $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
NHDFIT_ROLE_KERNELS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline > $OUT/stats.log 2>&1
for f in $(find $OUT/stats -name "*kernel_stats.csv"); do cut -c1-120 $f | head -12; done
