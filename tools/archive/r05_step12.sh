#!/bin/bash
# Round 5, GPU call 12: k_big_map with lane = tuple (parity subset, kernel trace of the general path for requests), and where a
# whole-batch nhdfit_find spends its time (NHDFIT_FIND_PROF, tuning build).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_step12
mkdir -p $OUT
cd $ROOT
SECONDS=0
timeout 400 python -m pytest tests -m gpu -x -q -k "big or wide or sharing" > $OUT/pytest_big.log 2>&1
echo "pytest rc=$? seconds=$SECONDS"; tail -3 $OUT/pytest_big.log
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o big -- python $ROOT/tools/time_big_find.py 4:65536 > $OUT/time_big_find.json 2> $OUT/time_big_find.err
echo "big rc=$? seconds=$SECONDS"
cat $OUT/time_big_find.json
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/big_kernel_stats.csv && head -6 $OUT/big_kernel_stats.csv
rm -rf $OUT/prof
cd $ROOT
NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so NHDFIT_FIND_PROF=1 timeout 200 python tools/time_batch_find.py > $OUT/batch_find.json 2> $OUT/batch_find_phases.log
echo "batch rc=$? seconds=$SECONDS"
cat $OUT/batch_find.json
tail -40 $OUT/batch_find_phases.log
