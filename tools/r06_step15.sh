#!/bin/bash
# Round 6, GPU call 16: four 512-thread blocks per CU? k_step compiled for 8 wavefronts per SIMD (<= 64 VGPRs, a few spilled) with the
# pair tables kept within a quarter of the CU's LDS - A/B against the shipped form (6 wavefronts, a third).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step15
mkdir -p $OUT
cd $ROOT
for pass in 1 2; do
  for v in base w8 w8s1 w8l3; do
    NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_cand_$v.so timeout 200 python tools/time_driver_form.py 20 60 | sed "s/^/$v /" | tee -a $OUT/driver_form_occupancy_ab.log | cut -c1-260
  done
done
