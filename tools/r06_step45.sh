#!/bin/bash
# Round 6, GPU call: the device soaks over a third set of seed ranges (finds + commit paths, the edges of the record formats, mode B's decision engine).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step45
mkdir -p $OUT
cd $ROOT
run() { name=$1; shift; timeout 1200 "$@" > $OUT/$name.log 2>&1; echo "$name rc=$? $(tail -1 $OUT/$name.log | cut -c1-200)"; }
run soak_gpu_3000_seeds_from_30000 python tools/soak_gpu.py 3000 30000
run soak_extreme_device_1500_seeds_from_5000 python tools/soak_extreme.py 1500 5000 --device
run soak_mode_b_300_seeds_from_2000 python tools/soak_mode_b_gpu.py 300 2000
run soak_big_device_300_seeds python tools/soak_big.py 300 --device
