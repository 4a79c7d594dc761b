#!/bin/bash
# Round 6, GPU call: the three-group tiles (row width 8: twice a narrow tile's cost per chunk) cut into 16 or 32 blocks instead of 8
# (NHDFIT_XCD_K8, tuning build) - a launch lasts as long as its slowest blocks.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step23
mkdir -p $OUT
cd $ROOT
export NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so
for pass in 1 2; do
  for k in 0 2 4; do
    NHDFIT_XCD_K8=$k timeout 200 python tools/time_driver_form.py 20 60 | sed "s/^/xcd_k8=$k /" | tee -a $OUT/driver_form_xcd_k8.log | cut -c1-250
  done
done
