#!/bin/bash
# Round 6, GPU call: the mode-B soak with its new regimes (small clusters that fill up, nine nodes in ten under maintenance, a dry run
# that is undone before the batch is decided for good).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step35
mkdir -p $OUT
cd $ROOT
timeout 1500 python tools/soak_mode_b_gpu.py ${1:-200} ${2:-120} > $OUT/soak_mode_b_gpu.log 2>&1
echo "soak rc=$? $(tail -1 $OUT/soak_mode_b_gpu.log)"
