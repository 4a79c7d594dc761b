#!/bin/bash
# Round 6, GPU call: the pipelined sweep reads its C row's address from the record (no further instantiation of the loop this time): parity, then A/B
# against the tree before (orig), steady state and driver form, one box.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step47
mkdir -p $OUT
cd $ROOT
for pass in 1 2 3; do
  for v in orig ship; do
    lib=$ROOT/nhd_amd/libnhdfit_cand_$v.so; [ $v = ship ] && lib=$ROOT/nhd_amd/libnhdfit.so
    NHDFIT_LIBRARY=$lib timeout 200 python tools/time_driver_form.py 1000 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v steady: median %.2f us per step (min %.2f max %.2f)' % (d['us_per_step_median'], d['min'], d['max']))" | tee -a $OUT/crow_ab.log
    NHDFIT_LIBRARY=$lib timeout 200 python tools/time_driver_form.py 20 60 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v driver form: median %.2f us per step (min %.2f max %.2f)' % (d['us_per_step_median'], d['min'], d['max']))" | tee -a $OUT/crow_ab.log
  done
done
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "not mode_b and not soak" > $OUT/parity.log 2>&1
echo "parity rc=$? $(grep -E 'passed|failed' $OUT/parity.log | tail -1)"
