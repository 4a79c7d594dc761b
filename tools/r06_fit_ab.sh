#!/bin/bash
# Round 6: A/B of fit-role variants built as candidate libraries (nhd_amd/libnhdfit_cand_<name>.so, selected through NHDFIT_LIBRARY):
# a parity subset, then the steady-state and the driver-form step of bench.py per variant; the bench legs run twice (second pass in
# reverse order) so that drift on the box shows.
#   gpurun --timeout 1800 -- bash tools/r06_fit_ab.sh "head base swp1 swp3 hoist swp3hoist" [tag]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
VARS=${1:-"head base"}
TAG=${2:-a}
OUT=$ROOT/gpurun_out/r06_fit_ab_$TAG
mkdir -p $OUT
cd $ROOT
for v in $VARS; do
  export NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_cand_$v.so
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "test_baseline_configs_vs_c_oracle and not 262144 and not 16384-pods or test_pipelined_steps_match_single_finds or test_single_launch_batch_find_equals_the_staged_path and c4-ragged" > $OUT/parity_$v.log 2>&1
  echo "$v parity rc=$? $(tail -1 $OUT/parity_$v.log)"
done
run_bench() {
  v=$1; pass=$2
  export NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_cand_$v.so
  timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc > $OUT/bench_${v}_$pass.json 2> $OUT/bench_${v}_$pass.err
  python - <<PY $OUT/bench_${v}_$pass.json $v $pass
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "pass", sys.argv[3], "driver-form %.3f us  repeats med %.3f  steady %.3f (min %.3f)  kernel_ms %.4f  cold %.3f" % (d["ms_per_step"]*1e3, d["repeats"]["ms_per_step_median"]*1e3, d["steady_state"]["ms_per_step_median"]*1e3, d["steady_state"]["ms_per_step_min"]*1e3, d["roofline"]["kernel_ms"], d["cold_start"]["ms_per_step"]*1e3))
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
}
for v in $VARS; do run_bench $v 1; done
for v in $(echo $VARS | tr ' ' '\n' | tac | tr '\n' ' '); do run_bench $v 2; done
