#!/bin/bash
# Round 5, GPU call 15: leaner staging + host-side phases of the single-launch batch find; k_big_eval back to lane = node without the memo.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_step15
mkdir -p $OUT
cd $ROOT
SECONDS=0
timeout 600 python -m pytest tests -m gpu -x -q -k "single_launch or pipelined or big or wide or general_path or sharing" > $OUT/pytest.log 2>&1
echo "pytest rc=$? seconds=$SECONDS"; tail -5 $OUT/pytest.log | cut -c1-400
timeout 200 python tools/time_batch_find.py "4:65536:4096,2:0:0,3:0:0,5:32768:2048" > $OUT/batch_find.json 2> $OUT/batch_find.err
echo "batch rc=$? seconds=$SECONDS"; cat $OUT/batch_find.json
NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so NHDFIT_FIND_PROF=1 timeout 100 python tools/time_batch_find.py "4:65536:4096,2:0:0" > $OUT/batch_find_tuning.json 2> $OUT/batch_find_phases.log
grep "P=4096" $OUT/batch_find_phases.log | tail -7; grep "P=256" $OUT/batch_find_phases.log | tail -7
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o big -- python $ROOT/tools/time_big_find.py 4:65536 > $OUT/time_big_find.json 2> $OUT/time_big_find.err
echo "big rc=$? seconds=$SECONDS"
cat $OUT/time_big_find.json | cut -c1-900
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/big_kernel_stats.csv && head -4 $OUT/big_kernel_stats.csv
rm -rf $OUT/prof
