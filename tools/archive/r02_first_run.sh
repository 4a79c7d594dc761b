#!/bin/bash
# Round-2 first GPU call: VALU issue calibration + the variants round 1 left unmeasured.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02_first
mkdir -p $OUT
cd $ROOT
timeout 120 tools/valu_calib > $OUT/valu_calib.jsonl 2> $OUT/valu_calib.err
B="python bench.py --steps 400 --warmup 20 --no-cpu-baseline"
timeout 90 $B > $OUT/bench_default.json 2> $OUT/bench_default.err
NHDFIT_SET_STATES=1 timeout 60 $B > $OUT/bench_set_states.json 2> $OUT/bench_set_states.err
NHDFIT_NODE_RECORDS=1 timeout 60 $B > $OUT/bench_node_records.json 2> $OUT/bench_node_records.err
NHDFIT_NODE_RECORDS=1 NHDFIT_SET_STATES=1 timeout 60 $B > $OUT/bench_both.json 2> $OUT/bench_both.err
NHDFIT_ROLE_TIMES=100 timeout 60 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 >/dev/null | grep nhdfit > $OUT/roles_default.log
NHDFIT_NODE_RECORDS=1 NHDFIT_SET_STATES=1 NHDFIT_ROLE_TIMES=100 timeout 60 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 >/dev/null | grep nhdfit > $OUT/roles_both.log
timeout 120 python tools/exp_outputs.py > $OUT/outputs_cost.json 2>&1
NHDFIT_SET_STATES=1 NHDFIT_NODE_RECORDS=1 timeout 400 python -m pytest tests -m gpu -x -q > $OUT/tests_both.log 2>&1
for f in bench_default bench_set_states bench_node_records bench_both; do
  python -c "import json,sys; j=json.load(open('$OUT/$f.json')); print('$f', round(j['value']/1e12,3), 'T evals/s', round(j['ms_per_step']*1e3,1), 'us/step', 'kernel', round(j['roofline']['kernel_ms']*1e3,1))"
done
cat $OUT/valu_calib.jsonl
cat $OUT/roles_default.log $OUT/roles_both.log; tail -3 $OUT/tests_both.log; cat $OUT/outputs_cost.json | tail -5
