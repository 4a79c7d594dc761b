#!/bin/bash
# Round 6, GPU call 9: the staging copy with four loads in flight; the pair table derived once per tile again, now copied the same way
# (NHDFIT_PAIR_DIGEST=1, tuning build); 60 driver-form regions per variant + role windows.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step8
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "test_baseline_configs_vs_c_oracle and not 262144 or test_pipelined_steps_match or test_full_size_config4" > $OUT/parity.log 2>&1
echo "parity (ship) rc=$? $(grep -E 'passed|failed' $OUT/parity.log | tail -1)"
NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so NHDFIT_PAIR_DIGEST=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "test_baseline_configs_vs_c_oracle and not 262144 or test_pipelined_steps_match or test_full_size_config4 or test_pipelined_steps_after_uploads or test_commits_and_deltas_between" > $OUT/parity_pd1.log 2>&1
echo "parity (tuning, pair table in digest) rc=$? $(grep -E 'passed|failed' $OUT/parity_pd1.log | tail -1)"
for pass in 1 2; do
  timeout 200 python tools/time_driver_form.py 20 60 | tee -a $OUT/driver_form_ab.log | cut -c1-330
  for pd in 0 1; do
    for pipes in 2 3; do
      NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so NHDFIT_PAIR_DIGEST=$pd NHDFIT_PIPES=$pipes timeout 200 python tools/time_driver_form.py 20 60 | sed "s/^/pd=$pd /" | tee -a $OUT/driver_form_ab.log | cut -c1-330
    done
  done
done
for pd in 0 1; do
  NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so NHDFIT_PAIR_DIGEST=$pd NHDFIT_ROLE_TIMES=400 timeout 200 python bench.py --steps 500 --warmup 20 --no-settle --no-extras --no-cpu-baseline --no-pmc 2>&1 >/dev/null | grep "fit blocks\|role fit\|role digest" | sed "s/^/pd=$pd /" | tee -a $OUT/role_windows.log
done
