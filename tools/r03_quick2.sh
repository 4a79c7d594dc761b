#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_quick2
mkdir -p $OUT
cd $ROOT
TL=$ROOT/nhd_amd/libnhdfit_tuning.so
{
echo "== c4"; NHDFIT_LIBRARY=$TL NHDFIT_SEQ_PROF=1 timeout 300 python tools/time_mode_b.py 65536 4096 4 2>&1 | tail -4 | cut -c1-330
echo "== c2"; NHDFIT_LIBRARY=$TL NHDFIT_SEQ_PROF=1 timeout 300 python tools/time_mode_b.py 4096 256 2 2>&1 | tail -4 | cut -c1-330
} 2>&1 | tee $OUT/modeb.log
