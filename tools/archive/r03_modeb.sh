#!/bin/bash
# Round 3: the decision engine form of mode B (seq2_kernel.h) - parity first, then decisions/s against the one-block kernel.
#   gpurun -- bash tools/r03_modeb.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_modeb
mkdir -p $OUT
cd $ROOT
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mode_b or commit or sched or pending or delta or heterogeneous" > $OUT/pytest_modeb.log 2>&1
echo "pytest rc=$? seconds=$SECONDS" | tee -a $OUT/pytest_modeb.log
grep -E "passed|failed|error|Error|assert" $OUT/pytest_modeb.log | tail -8
TL=$ROOT/nhd_amd/libnhdfit_tuning.so
{
echo "== c4 65536x4096 decision engine"; timeout 300 python tools/time_mode_b.py 65536 4096 4 2>&1 | tail -1
echo "== c4 65536x4096 general (tuning build)"; NHDFIT_LIBRARY=$TL NHDFIT_SEQ_GENERAL=1 timeout 300 python tools/time_mode_b.py 65536 4096 4 2>&1 | tail -1
echo "== c4 65536x4096 4 worker blocks (tuning)"; NHDFIT_LIBRARY=$TL NHDFIT_SEQ_WORKERS=4 timeout 300 python tools/time_mode_b.py 65536 4096 4 2>&1 | tail -1
echo "== c4 65536x4096 32 worker blocks (tuning)"; NHDFIT_LIBRARY=$TL NHDFIT_SEQ_WORKERS=32 timeout 300 python tools/time_mode_b.py 65536 4096 4 2>&1 | tail -1
echo "== c5 32768x2048 decision engine"; timeout 300 python tools/time_mode_b.py 32768 2048 5 2>&1 | tail -1
echo "== c5 32768x16384 decision engine"; timeout 300 python tools/time_mode_b.py 32768 16384 5 2>&1 | tail -1
echo "== c3 16384x1024 decision engine"; timeout 300 python tools/time_mode_b.py 16384 1024 3 2>&1 | tail -1
echo "== c2 4096x256 decision engine"; timeout 300 python tools/time_mode_b.py 4096 256 2 2>&1 | tail -1
} 2>&1 | tee $OUT/modeb_times.log
ST=$OUT/stats; rm -rf $ST; mkdir -p $ST
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ST -o s -- python $ROOT/tools/time_mode_b.py 65536 4096 4 > $ST/run.log 2>&1)
find $ST -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_modeb.csv \;
head -12 $OUT/kernel_stats_modeb.csv | cut -c1-200
timeout 300 python tools/exp_two_pipes.py 2 2>&1 | tail -1 | tee $OUT/two_pipes.log
timeout 300 python tools/exp_two_pipes.py 3 2>&1 | tail -1 | tee -a $OUT/two_pipes.log
