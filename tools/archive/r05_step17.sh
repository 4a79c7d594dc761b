#!/bin/bash
# Round 5, GPU call 17: mode B with relaxed polls in the workers / the publication wait (parity subset + rates); batch find: pieces per tile.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_step17
mkdir -p $OUT
cd $ROOT
SECONDS=0
timeout 700 python -m pytest tests -m gpu -x -q -k "mode_b or schedule or seq or commit or decide or pending or heterogeneous or rank_to_rank" > $OUT/pytest.log 2>&1
echo "pytest rc=$? seconds=$SECONDS"; tail -4 $OUT/pytest.log | cut -c1-300
for shape in "4096 256 2" "65536 4096 4" "16384 1024 3" "32768 2048 5"; do
  for rep in 1 2; do timeout 120 python tools/time_mode_b.py $shape 2>/dev/null | tee -a $OUT/mode_b.log; done
done
echo "mode b seconds=$SECONDS"
for k in 1 2 4; do
NHDFIT_XCD_K=$k NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so timeout 100 python tools/time_batch_find.py "4:65536:4096,3:0:0" 2>/dev/null | tee -a $OUT/batch_find_k.log
done
