#!/bin/bash
# Round 6, GPU call 19: digest blocks per tile (NHDFIT_WC_PARTS, tuning build) with the shorter launches of this round - every side-role
# block holds a full block slot (the launch's LDS size, eight wavefronts) for the length of its latency chain.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step18
mkdir -p $OUT
cd $ROOT
export NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so
for pass in 1 2; do
  for parts in 1 2 3 4; do
    NHDFIT_WC_PARTS=$parts timeout 200 python tools/time_driver_form.py 20 60 | sed "s/^/wc_parts=$parts /" | tee -a $OUT/driver_form_wc_parts.log | cut -c1-260
  done
done
