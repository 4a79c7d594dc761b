#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_quick2
mkdir -p $OUT
cd $ROOT
TL=$ROOT/nhd_amd/libnhdfit_tuning.so
{
for k in 0 1 2 4 7; do
echo "== c2 skip $k"; NHDFIT_LIBRARY=$TL NHDFIT_SEQ_SKIP=$k NHDFIT_SEQ_PROF=1 timeout 300 python tools/time_mode_b.py 4096 256 2 2>&1 | grep "driver" | cut -c1-330
done
} 2>&1 | tee $OUT/modeb.log
