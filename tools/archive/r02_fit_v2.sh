#!/bin/bash
# GPU call: parity of the v2 fit path + tuning sweep (chunks per wavefront, block size)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_fit_v2
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
B="python bench.py --steps 400 --warmup 20 --no-cpu-baseline"
run() { tag=$1; shift; env "$@" timeout 90 $B > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; python - <<PY
import json
try:
    j=json.load(open('$OUT/bench_$tag.json')); print('$tag', round(j['value']/1e12,3), 'T evals/s', round(j['ms_per_step']*1e3,1), 'us/step kernel', round(j['roofline']['kernel_ms']*1e3,1))
except Exception as e: print('$tag FAILED', e, open('$OUT/bench_$tag.err').read()[-400:])
PY
}
run default X=1
run b256 NHDFIT_BLOCK=256
run cpw16 NHDFIT_CPW=16,12,8,4
run cpw4 NHDFIT_CPW=4,3,2,1
run b256cpw16 NHDFIT_BLOCK=256 NHDFIT_CPW=16,12,8,4
run nomap X=1
NHDFIT_ROLE_TIMES=100 timeout 60 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 >/dev/null | grep nhdfit > $OUT/roles.log
NHDFIT_SPLIT=1 NHDFIT_ROLE_TIMES=100 timeout 60 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 >/dev/null | grep nhdfit > $OUT/roles_split.log
cat $OUT/roles.log; echo; cat $OUT/roles_split.log
timeout 120 python tools/exp_outputs.py > $OUT/outputs_cost.json 2>&1; cat $OUT/outputs_cost.json
