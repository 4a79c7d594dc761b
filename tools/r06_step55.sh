#!/bin/bash
# Round 6, last GPU call: soaks on the tree as committed, seed ranges no earlier call drew.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step55
mkdir -p $OUT
cd $ROOT
run() { name=$1; shift; timeout 420 "$@" > $OUT/$name.log 2>&1; echo "$name rc=$? $(tail -1 $OUT/$name.log | cut -c1-200)"; }
run soak_gpu_1500_seeds_from_70000 python tools/soak_gpu.py 1500 70000
run soak_mode_b_150_seeds_from_5000 python tools/soak_mode_b_gpu.py 150 5000
