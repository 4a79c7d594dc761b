#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_quick2
mkdir -p $OUT
cd $ROOT
TL=$ROOT/nhd_amd/libnhdfit_tuning.so
{
for d in 0 1 2 4; do
echo "== c4 hint distance $d"; NHDFIT_LIBRARY=$TL NHDFIT_HINT_DISTANCE=$d NHDFIT_SEQ_PROF=1 timeout 300 python tools/time_mode_b.py 65536 4096 4 2>&1 | tail -3 | cut -c1-330
done
for w in 6 24; do
echo "== c4 workers $w"; NHDFIT_LIBRARY=$TL NHDFIT_SEQ_WORKERS=$w NHDFIT_SEQ_PROF=1 timeout 300 python tools/time_mode_b.py 65536 4096 4 2>&1 | tail -3 | cut -c1-330
done
echo "== c5 16384 ship"; timeout 300 python tools/time_mode_b.py 32768 16384 5 2>&1 | tail -1 | cut -c1-250
echo "== c5 16384 d=0"; NHDFIT_LIBRARY=$TL NHDFIT_HINT_DISTANCE=0 NHDFIT_SEQ_PROF=1 timeout 300 python tools/time_mode_b.py 32768 16384 5 2>&1 | tail -3 | cut -c1-330
echo "== c2 d=0"; NHDFIT_LIBRARY=$TL NHDFIT_HINT_DISTANCE=0 NHDFIT_SEQ_PROF=1 timeout 300 python tools/time_mode_b.py 4096 256 2 2>&1 | tail -3 | cut -c1-330
} 2>&1 | tee $OUT/modeb.log
