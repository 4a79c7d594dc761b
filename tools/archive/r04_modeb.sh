#!/bin/bash
# Round 4: mode B as speculate + retire (seq2_kernel.h) - parity first, then decisions/s and the sequencer's phase log.
#   gpurun -- bash tools/r04_modeb.sh [tag] [quick]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-a}
QUICK=${2:-}
OUT=$ROOT/gpurun_out/r04_modeb_$TAG
mkdir -p $OUT
cd $ROOT
SECONDS=0
if [ -z "$QUICK" ]; then
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mode_b or commit or sched or pending or heterogeneous" > $OUT/pytest_modeb.log 2>&1
else
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mode_b_at_baseline or mode_b_sequential or heterogeneous" > $OUT/pytest_modeb.log 2>&1
fi
echo "pytest rc=$? seconds=$SECONDS" | tee -a $OUT/pytest_modeb.log
grep -E "passed|failed|error|Error|assert" $OUT/pytest_modeb.log | tail -8
TL=$ROOT/nhd_amd/libnhdfit_tuning.so
{
for shape in "65536 4096 4" "4096 256 2" "16384 1024 3" "32768 2048 5" "32768 16384 5"; do
  echo "== $shape ship"; timeout 300 python tools/time_mode_b.py $shape 2>&1 | tail -1
  echo "== $shape phases (tuning build)"; NHDFIT_LIBRARY=$TL NHDFIT_SEQ_PROF=1 timeout 300 python tools/time_mode_b.py $shape 2>&1 | tail -4
  for extra in $ROOT/nhd_amd/libnhdfit_tuning_*.so; do
    [ -f "$extra" ] || continue
    echo "== $shape phases ($(basename $extra))"; NHDFIT_LIBRARY=$extra NHDFIT_SEQ_PROF=1 timeout 300 python tools/time_mode_b.py $shape 2>&1 | tail -4
  done
done
} 2>&1 | tee $OUT/modeb_times.log
