#!/bin/bash
# Round 4: mode B, worker-block sweep on the tuning build.  gpurun -- bash tools/r04_modeb_workers.sh [tag]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_modeb_${1:-w}
mkdir -p $OUT; cd $ROOT
TL=$ROOT/nhd_amd/libnhdfit_tuning.so
{
for w in 8 16 32; do
  for shape in "65536 4096 4" "32768 16384 5"; do
    echo "== $shape workers $w"; NHDFIT_LIBRARY=$TL NHDFIT_SEQ_WORKERS=$w timeout 300 python tools/time_mode_b.py $shape 2>&1 | tail -1 | cut -c1-200
  done
done
} 2>&1 | tee $OUT/workers.log
