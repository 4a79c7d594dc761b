#!/bin/bash
# Round 6, GPU call: the decision engine's dynamic LDS is admitted against the room the code object leaves (the soak's first seed was
# refused by hipFuncSetAttribute): the regression test, then the mode-B soak it came from.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step34
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "edge_of_the_lds or mode_b_sparse" > $OUT/parity_lds_edge.log 2>&1
echo "regression test rc=$? $(grep -E 'passed|failed' $OUT/parity_lds_edge.log | tail -1)"
timeout 1500 python tools/soak_mode_b_gpu.py ${1:-120} > $OUT/soak_mode_b_gpu.log 2>&1
echo "soak rc=$? $(tail -1 $OUT/soak_mode_b_gpu.log)"
