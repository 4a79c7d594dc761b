#!/bin/bash
# Round 6, GPU call: the step's floor - fit blocks that sweep ONE chunk per wavefront (NHDFIT_FIT_SKIP=64, tuning build; results are wrong
# with it) against the step as shipped, at two, three and four launches in flight, and with the fit role's blocks per tile halved.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step42
mkdir -p $OUT
cd $ROOT
for pipes in 2 3 4; do
  for skip in 0 64 80; do
    NHDFIT_PIPES=$pipes NHDFIT_FIT_SKIP=$skip NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so timeout 200 python tools/time_driver_form.py 1000 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipes $pipes skip $skip: median %.2f us per step (min %.2f max %.2f)' % (d['us_per_step_median'], d['min'], d['max']))" | tee -a $OUT/step_floor.log
  done
done
NHDFIT_ROLE_TIMES=1 NHDFIT_FIT_SKIP=64 NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so timeout 200 python tools/time_driver_form.py 200 3 2>&1 | grep -i "nhdfit\]" | tail -12 | tee -a $OUT/step_floor.log
NHDFIT_ROLE_TIMES=1 NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so timeout 200 python tools/time_driver_form.py 200 3 2>&1 | grep -i "nhdfit\]" | tail -12 | tee -a $OUT/step_floor.log
