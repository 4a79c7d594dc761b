#!/bin/bash
# First GPU call of the next round (run through gpurun, ~3 GPU-minutes): times what round 1 left prepared but
# unmeasured, with the role clocks that show where a step's time goes.
#   tools/next_round_first_run.sh            -> gpurun_out/next_round/*.json|log
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/next_round
mkdir -p $OUT
cd $ROOT
B="python bench.py --steps 400 --warmup 20 --no-cpu-baseline"
timeout 60 $B > $OUT/bench_default.json 2> $OUT/bench_default.err
NHDFIT_SET_STATES=1 timeout 60 $B > $OUT/bench_set_states.json 2> $OUT/bench_set_states.err       # state machine for G = 3 shapes
NHDFIT_NODE_RECORDS=1 timeout 60 $B > $OUT/bench_node_records.json 2> $OUT/bench_node_records.err   # precomputed node records in the fit role
NHDFIT_NODE_RECORDS=1 NHDFIT_SET_STATES=1 timeout 60 $B > $OUT/bench_both.json 2> $OUT/bench_both.err
NHDFIT_NO_CHOOSE_TABLE=1 timeout 60 $B > $OUT/bench_no_choose_table.json 2> $OUT/bench_no_choose_table.err
NHDFIT_ROLE_TIMES=100 timeout 60 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 >/dev/null | grep nhdfit > $OUT/roles_default.log
NHDFIT_SET_STATES=1 NHDFIT_ROLE_TIMES=100 timeout 60 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 >/dev/null | grep nhdfit > $OUT/roles_set_states.log
timeout 120 python tools/exp_outputs.py > $OUT/outputs_cost.json 2>&1                                # bitmap / mapping roles on and off
NHDFIT_SET_STATES=1 NHDFIT_NODE_RECORDS=1 timeout 300 python -m pytest tests -m gpu -x -q > $OUT/tests_set_states.log 2>&1  # full parity with both opt-ins
for f in bench_default bench_set_states bench_node_records bench_both bench_no_choose_table; do
  python -c "import json,sys; j=json.load(open('$OUT/$f.json')); print('$f', round(j['value']/1e12,3), 'T evals/s', round(j['ms_per_step']*1e3,1), 'us/step')"
done
cat $OUT/roles_default.log $OUT/roles_set_states.log; tail -1 $OUT/tests_set_states.log
