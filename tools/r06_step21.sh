#!/bin/bash
# Round 6, GPU call: four fit blocks per tile instead of eight for the row-width classes in NHDFIT_FIT_HALF's mask (tuning build) - half as
# many stagings and pair-table derivations, twice the chunks per wavefront - now that the sweep itself is shorter.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step21
mkdir -p $OUT
cd $ROOT
export NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so
for pass in 1 2; do
  for mask in 0 1 2 3 4 7; do
    NHDFIT_FIT_HALF=$mask timeout 200 python tools/time_driver_form.py 20 60 | sed "s/^/fit_half=$mask /" | tee -a $OUT/driver_form_fit_half.log | cut -c1-250
  done
done
