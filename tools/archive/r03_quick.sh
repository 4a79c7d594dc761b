#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_quick
mkdir -p $OUT
cd $ROOT
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mode_b or commit or sched or pending or delta or heterogeneous" > $OUT/pytest_modeb.log 2>&1
echo "pytest rc=$? seconds=$SECONDS" | tee -a $OUT/pytest_modeb.log
grep -E "passed|failed|error|Error|assert" $OUT/pytest_modeb.log | tail -8
TL=$ROOT/nhd_amd/libnhdfit_tuning.so
{
echo "== c4"; NHDFIT_LIBRARY=$TL NHDFIT_SEQ_PROF=1 timeout 300 python tools/time_mode_b.py 65536 4096 4 2>&1 | tail -3
echo "== c2"; NHDFIT_LIBRARY=$TL NHDFIT_SEQ_PROF=1 timeout 300 python tools/time_mode_b.py 4096 256 2 2>&1 | tail -3
echo "== c5 2048"; NHDFIT_LIBRARY=$TL NHDFIT_SEQ_PROF=1 timeout 300 python tools/time_mode_b.py 32768 2048 5 2>&1 | tail -3
echo "== c4 ship"; timeout 300 python tools/time_mode_b.py 65536 4096 4 2>&1 | tail -1
echo "== c5 16384 ship"; timeout 300 python tools/time_mode_b.py 32768 16384 5 2>&1 | tail -1
} 2>&1 | tee $OUT/modeb.log
