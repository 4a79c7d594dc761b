#!/bin/bash
# Round 6, GPU call: three-way A/B on one box - orig (the tree before), nocrow (optional pointers tested as bits of one scalar), ship (that + the C row's
# address from the record): steady state and driver form.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step44
mkdir -p $OUT
cd $ROOT
for pass in 1 2 3; do
  for v in orig nocrow ship; do
    lib=$ROOT/nhd_amd/libnhdfit_cand_$v.so; [ $v = ship ] && lib=$ROOT/nhd_amd/libnhdfit.so
    NHDFIT_LIBRARY=$lib timeout 200 python tools/time_driver_form.py 1000 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v steady: median %.2f us per step (min %.2f max %.2f)' % (d['us_per_step_median'], d['min'], d['max']))" | tee -a $OUT/crow_ab3.log
    NHDFIT_LIBRARY=$lib timeout 200 python tools/time_driver_form.py 20 60 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v driver form: median %.2f us per step (min %.2f max %.2f)' % (d['us_per_step_median'], d['min'], d['max']))" | tee -a $OUT/crow_ab3.log
  done
done
