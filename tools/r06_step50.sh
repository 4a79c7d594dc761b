#!/bin/bash
# Round 6, GPU call: the whole GPU suite on the shipped tree (per-test time limit: a test that stands still is named, not waited for).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step50
mkdir -p $OUT
cd $ROOT
SECONDS=0
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider -x --timeout=240 --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? seconds=$SECONDS $(grep -E 'passed|failed' $OUT/pytest_gpu.log | tail -1)"
tail -25 $OUT/pytest_gpu.log | cut -c1-200
