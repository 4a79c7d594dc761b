#!/bin/bash
# Round 6, GPU call: where does the mode-B test at the config-5 shard x 16 384 pods stand still on the tree whose records carry the C row?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step49
mkdir -p $OUT
cd $ROOT
timeout 150 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -o faulthandler_timeout=60 -k "mode_b_at_baseline_sizes and c5-shard-x-16384" > $OUT/hang.log 2>&1
echo "rc=$?"; tail -40 $OUT/hang.log | cut -c1-200
