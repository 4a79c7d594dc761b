#!/bin/bash
# Round 5, GPU call 14: hand-offs of the single-launch finds by the guide's recipe (relaxed polls, one acquire, one lane's release), and
# k_big_eval with lane = assignment: parity subsets, then the calls' wall times.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_step14
mkdir -p $OUT
cd $ROOT
SECONDS=0
timeout 600 python -m pytest tests -m gpu -x -q -k "single_launch or pipelined or edge_cases or golden_vectors or big or wide or sharing or general_path" > $OUT/pytest.log 2>&1
echo "pytest rc=$? seconds=$SECONDS"; tail -25 $OUT/pytest.log | cut -c1-400
timeout 200 python tools/time_batch_find.py "4:65536:4096,2:0:0,3:0:0,5:32768:2048,5:32768:16384" > $OUT/batch_find.json 2> $OUT/batch_find.err
echo "batch rc=$? seconds=$SECONDS"
cat $OUT/batch_find.json; tail -3 $OUT/batch_find.err
timeout 200 python tools/time_single_find.py > $OUT/single_find.json 2> $OUT/single_find.err
echo "single rc=$? seconds=$SECONDS"; cat $OUT/single_find.json | cut -c1-1500
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o big -- python $ROOT/tools/time_big_find.py 4:65536 > $OUT/time_big_find.json 2> $OUT/time_big_find.err
echo "big rc=$? seconds=$SECONDS"
cat $OUT/time_big_find.json | cut -c1-1200
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/big_kernel_stats.csv && head -4 $OUT/big_kernel_stats.csv
rm -rf $OUT/prof
