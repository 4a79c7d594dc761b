#!/bin/bash
# Round 4: timing of mode-B build variants (nhd_amd/libnhdfit_tuning_*.so), no parity run.  gpurun -- bash tools/r04_modeb_variants.sh [tag]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-v}
OUT=$ROOT/gpurun_out/r04_modeb_$TAG
mkdir -p $OUT
cd $ROOT
{
for shape in "4096 256 2" "65536 4096 4" "16384 1024 3"; do
  for lib in $ROOT/nhd_amd/libnhdfit_tuning.so $ROOT/nhd_amd/libnhdfit_tuning_*.so; do
    [ -f "$lib" ] || continue
    echo "== $shape $(basename $lib)"; NHDFIT_LIBRARY=$lib NHDFIT_SEQ_PROF=1 timeout 300 python tools/time_mode_b.py $shape 2>&1 | tail -4
  done
done
} 2>&1 | tee $OUT/modeb_variants.log
