#!/bin/bash
# Round 6, GPU call 5: the pair table derived once per tile by the digest role (NHDFIT_PAIR_DIGEST=0/1 on the tuning build: same binary),
# parity subset + driver-form / steady-state bench on the ship build, the fit blocks' phases again.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step4
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "test_baseline_configs_vs_c_oracle and not 262144 or test_pipelined_steps or test_full_size_config4 or test_mode_b_at_baseline_sizes and not 262144 or test_pipelined_steps_after_uploads or test_commits_and_deltas_between or test_more_node_classes" > $OUT/parity.log 2>&1
echo "parity rc=$? $(grep -E 'passed|failed' $OUT/parity.log | tail -1)"
grep -E "^FAILED|^E  " $OUT/parity.log | head
bench_line() {
  python - <<PY $1 "$2"
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "driver-form %.3f us  repeats med %.3f  steady %.3f (min %.3f)  kernel_ms %.4f  cold %.3f" % (d["ms_per_step"]*1e3, d["repeats"]["ms_per_step_median"]*1e3, d["steady_state"]["ms_per_step_median"]*1e3, d["steady_state"]["ms_per_step_min"]*1e3, d["roofline"]["kernel_ms"], d["cold_start"]["ms_per_step"]*1e3))
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
}
for pass in 1 2; do
  for v in 0 1; do
    NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so NHDFIT_PAIR_DIGEST=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc > $OUT/bench_tuning_pd${v}_$pass.json 2>/dev/null
    bench_line $OUT/bench_tuning_pd${v}_$pass.json "tuning build, pair table in digest = $v, pass $pass:"
  done
done
for pass in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc > $OUT/bench_ship_$pass.json 2>/dev/null
  bench_line $OUT/bench_ship_$pass.json "ship build pass $pass:"
done
for v in 0 1; do
  for step in 40 400; do
    NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so NHDFIT_PAIR_DIGEST=$v NHDFIT_ROLE_TIMES=$step timeout 200 python bench.py --steps 500 --warmup 20 --no-settle --no-extras --no-cpu-baseline --no-pmc 2>&1 >/dev/null | grep "nhdfit" | sed "s/^/pd=$v /" >> $OUT/role_windows.log
  done
done
grep -E "fit blocks|role digest|role fit" $OUT/role_windows.log
for cfgs in "5 32768 2048" "5 32768 16384" "3 16384 1024" "2 4096 256"; do
  set -- $cfgs
  timeout 300 python bench.py --config $1 --nodes-per-gpu $2 --pods $3 --steps 300 --warmup 50 --no-extras --no-cpu-baseline --no-pmc > $OUT/bench_c$1_$2_$3.json 2>/dev/null
  bench_line $OUT/bench_c$1_$2_$3.json "ship build config $1 $2 x $3:"
done
