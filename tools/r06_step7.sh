#!/bin/bash
# Round 6, GPU call 8: the driver-form region over 60 repeats per variant (tools/time_driver_form.py): pipes 2 / 3, the drain after the
# register-resident candidate masks; parity subset on the ship build (mapping paths + mode B with the lean decision engine).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step7
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "test_golden_vectors or test_random_clusters or test_baseline_configs_vs_c_oracle and not 262144 or test_mode_b_at_baseline_sizes and not 262144 or test_mode_b_heterogeneous or test_config_text or test_single_launch_find_equals" > $OUT/parity.log 2>&1
echo "parity rc=$? $(grep -E 'passed|failed' $OUT/parity.log | tail -1)"; grep -E "^FAILED|^E  " $OUT/parity.log | head
for pass in 1 2; do
  timeout 200 python tools/time_driver_form.py 20 60 | tee -a $OUT/driver_form_ab.log | cut -c1-400
  for pipes in 2 3; do
    NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so NHDFIT_PIPES=$pipes timeout 200 python tools/time_driver_form.py 20 60 | tee -a $OUT/driver_form_ab.log | cut -c1-400
  done
done
NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so NHDFIT_DRAIN_PROF=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc 2>&1 >/dev/null | grep "drain of" | head -4 | tee $OUT/drain_phases_after.log
