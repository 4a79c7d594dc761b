#!/bin/bash
# Round 6, GPU call 12: the pod-at-a-time loop call by call; two against three launches in flight with the non-temporal stores.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step11
mkdir -p $OUT
cd $ROOT
timeout 300 python tools/time_pod_loop.py 16384 1024 | tee $OUT/pod_loop_16384.json
timeout 300 python tools/time_pod_loop.py 65536 512 | tee $OUT/pod_loop_65536.json
for pass in 1 2; do
  for pipes in 2 3; do
    NHDFIT_PIPES=$pipes NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so timeout 200 python tools/time_driver_form.py 20 60 | sed "s/^/pipes=$pipes /" | tee -a $OUT/driver_form_pipes.log | cut -c1-330
  done
done
