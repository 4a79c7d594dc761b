#!/bin/bash
# Round 4: the general path for requests after the odometer walk + per-NUMA memo of the NIC stage - its GPU tests and its kernel trace again.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_big2
mkdir -p $OUT
cd $ROOT
SECONDS=0
timeout 90 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "big_pods or general_path_for_requests or wide_nodes_at_scale or reference_generated_mixed" > $OUT/pytest_big.log 2>&1
echo "pytest rc=$? seconds=$SECONDS" | tee -a $OUT/pytest_big.log
grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_big.log | tail -6
cd /tmp && export TMPDIR=/tmp
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o big -- python $ROOT/tools/time_big_find.py > $OUT/time_big_find.json 2> $OUT/time_big_find.err
echo "rc=$? seconds=$SECONDS"
cat $OUT/time_big_find.json
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/big_kernel_stats.csv && head -5 $OUT/big_kernel_stats.csv
rm -rf $OUT/prof
