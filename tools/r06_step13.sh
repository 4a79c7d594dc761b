#!/bin/bash
# Round 6, GPU call 14: soak of the commit paths on the device (batched and pod by pod against the oracle), the sharing fixtures with
# several RX / TX cores per group, the wire path.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step13
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "sharing or wire or configs" > $OUT/parity_sharing_wire.log 2>&1
echo "parity (sharing, wire) rc=$? $(grep -E 'passed|failed' $OUT/parity_sharing_wire.log | tail -1)"; grep -E "^FAILED|^E  " $OUT/parity_sharing_wire.log | head
timeout 900 python tools/soak_gpu.py 400 1000 2>&1 | tail -5 | tee $OUT/soak_gpu.log
