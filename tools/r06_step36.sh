#!/bin/bash
# Round 6, GPU call: the device soaks once more over seed ranges no earlier call drew (finds + commit paths, the same under ENABLE_SHARING,
# the edges of the record formats, big pods, delta streams, mode B's decision engine).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step36
mkdir -p $OUT
cd $ROOT
run() { name=$1; shift; timeout 1200 "$@" > $OUT/$name.log 2>&1; echo "$name rc=$? $(tail -1 $OUT/$name.log | cut -c1-200)"; }
run soak_gpu_3000_seeds_from_10000 python tools/soak_gpu.py 3000 10000
run soak_gpu_sharing_600_seeds_from_20000 python tools/soak_gpu.py 600 20000 --sharing
run soak_extreme_device_1000_seeds_from_1000 python tools/soak_extreme.py 1000 1000 --device
run soak_big_device_200_seeds python tools/soak_big.py 200 --device
run soak_deltas_600_streams python tools/soak_deltas_gpu.py 600
run soak_mode_b_400_seeds_from_320 python tools/soak_mode_b_gpu.py 400 320
