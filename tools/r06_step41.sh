#!/bin/bash
# Round 6, GPU call: what the parts of the fit role's chunk loop are worth (tuning build, NHDFIT_FIT_SKIP ablations; results are WRONG with them):
# 0 as shipped, 1 no table sweep (no row fetches, no combining), 2 no winner tracking, 4 constant record (no record loads), 8 no predicate rows, 16 no staging, 64 one chunk per wavefront, 128 no pair table; sums combine.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step41
mkdir -p $OUT
cd $ROOT
for pass in 1; do
  for skip in 0 1 4 5 16 128 144 149 64 213; do
    NHDFIT_FIT_SKIP=$skip NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so timeout 200 python tools/time_driver_form.py 1000 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip $skip', json.dumps(d)[:300])" | tee -a $OUT/fit_skip_ablation.log
  done
done
