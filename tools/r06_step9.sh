#!/bin/bash
# Round 6, GPU call 10: non-temporal verdict stores A/B (candidate libraries), the polled commit (parity + scheduler-loop leg).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step9
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "commit or attach or scheduler_loop or pending_list or mutators or golden_vectors or deltas" > $OUT/parity_commit.log 2>&1
echo "parity (commit paths, ship) rc=$? $(grep -E 'passed|failed' $OUT/parity_commit.log | tail -1)"; grep -E "^FAILED|^E  " $OUT/parity_commit.log | head
NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_cand_nt.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "test_baseline_configs_vs_c_oracle and not 262144 or test_pipelined_steps_match or test_full_size_config4 or test_mode_b_at_baseline_sizes and c4" > $OUT/parity_nt.log 2>&1
echo "parity (nt) rc=$? $(grep -E 'passed|failed' $OUT/parity_nt.log | tail -1)"
for pass in 1 2 3; do
  for v in plain nt; do
    NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_cand_$v.so timeout 200 python tools/time_driver_form.py 20 60 | sed "s/^/$v /" | tee -a $OUT/driver_form_nt_ab.log | cut -c1-330
  done
done
timeout 300 python -c "
import json, bench
print(json.dumps(bench.sched_loop(4, 0)))" | tee $OUT/sched_loop_bench_leg.json
timeout 300 python tools/time_sched_loop.py 2>/dev/null | tee $OUT/sched_loop_with_standin_bookkeeping.json
