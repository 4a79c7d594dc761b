#!/bin/bash
# Round 6, GPU call: the whole GPU suite and soaks on the tree whose records carry the C row (NodeRec::flags).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step48
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? $(grep -E 'passed|failed' $OUT/pytest_gpu.log | tail -1)"
run() { name=$1; shift; timeout 900 "$@" > $OUT/$name.log 2>&1; echo "$name rc=$? $(tail -1 $OUT/$name.log | cut -c1-200)"; }
run soak_gpu_800_seeds_from_50000 python tools/soak_gpu.py 800 50000
run soak_extreme_device_400_seeds_from_9000 python tools/soak_extreme.py 400 9000 --device
run soak_mode_b_100_seeds_from_3000 python tools/soak_mode_b_gpu.py 100 3000
run soak_deltas_200_streams python tools/soak_deltas_gpu.py 200
