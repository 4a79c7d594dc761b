#!/bin/bash
# Round 4: kernel trace of the general path for requests (k_big_eval / k_big_map / k_big_commit) on config 4's 65 536 nodes and the config-5 shard.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_big_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SECONDS=0
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o big -- python $ROOT/tools/time_big_find.py > $OUT/time_big_find.json 2> $OUT/time_big_find.err
echo "rc=$? seconds=$SECONDS"
cat $OUT/time_big_find.json
find $OUT/prof -name "*kernel_stats*" | head -3
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/big_kernel_stats.csv && head -8 $OUT/big_kernel_stats.csv
rm -rf $OUT/prof
