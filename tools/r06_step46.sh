#!/bin/bash
# Round 6, GPU call: how sensitive is the step to vector instructions in the pipelined chunk loop?  0 / 8 / 16 / 32 extra v_and_b32 per chunk on a side
# chain (-DNHDFIT_FIT_PAD=n, a probe that is never defined in a shipped build), steady state, one box.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step46
mkdir -p $OUT
cd $ROOT
for pass in 1 2; do
  for n in 0 8 16 32; do
    NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_cand_pad$n.so timeout 200 python tools/time_driver_form.py 1000 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pad $n steady: median %.2f us per step (min %.2f max %.2f)' % (d['us_per_step_median'], d['min'], d['max']))" | tee -a $OUT/valu_pad_sensitivity.log
  done
done
