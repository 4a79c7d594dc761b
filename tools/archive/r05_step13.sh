#!/bin/bash
# Round 5, GPU call 13: the single-launch find of a whole batch (k_findn) - parity against the staged path, then the call's wall time.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_step13
mkdir -p $OUT
cd $ROOT
SECONDS=0
timeout 500 python -m pytest tests -m gpu -x -q -k "single_launch or pipelined or edge_cases or golden_vectors" > $OUT/pytest_findn.log 2>&1
echo "pytest rc=$? seconds=$SECONDS"; tail -25 $OUT/pytest_findn.log | cut -c1-400
timeout 200 python tools/time_batch_find.py "4:65536:4096,2:0:0,3:0:0,5:32768:2048,5:32768:16384" > $OUT/batch_find.json 2> $OUT/batch_find.err
echo "batch rc=$? seconds=$SECONDS"
cat $OUT/batch_find.json; tail -5 $OUT/batch_find.err
NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so NHDFIT_FIND_PROF=1 timeout 100 python tools/time_batch_find.py "4:65536:4096,2:0:0" > $OUT/batch_find_tuning.json 2> $OUT/batch_find_phases.log
grep "P=4096" $OUT/batch_find_phases.log | tail -4; grep "P=256" $OUT/batch_find_phases.log | tail -3
