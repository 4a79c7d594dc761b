#!/bin/bash
# Round 3: two pipes natively + decision engine with fresher hints.   gpurun -- bash tools/r03_pipes.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_pipes
mkdir -p $OUT
cd $ROOT
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not baseline_sizes and not full_size" > $OUT/pytest.log 2>&1
echo "pytest rc=$? seconds=$SECONDS" | tee -a $OUT/pytest.log
grep -E "passed|failed|error|Error|assert" $OUT/pytest.log | tail -8
B="python bench.py --no-cpu-baseline --no-pmc --no-extras"
line() { python -c "import sys,json; o=json.loads(sys.stdin.read()); r=o['roofline']; print(o['value']/1e12, o['ms_per_step'], r['kernel_ms'], r['concurrency'], r['frac'], r['achieved_from_wall'], o['placed_pods'], o.get('repeats'))"; }
TL=$ROOT/nhd_amd/libnhdfit_tuning.so
{
echo "== ship (two pipes)"; timeout 300 $B 2>&1 | tail -1 | line
echo "== ship (two pipes)"; timeout 300 $B 2>&1 | tail -1 | line
echo "== tuning, one pipe"; NHDFIT_LIBRARY=$TL NHDFIT_ONE_PIPE=1 timeout 300 $B 2>&1 | tail -1 | line
echo "== c5 shard x 16384 pods"; timeout 300 $B --config 5 --nodes-per-gpu 32768 --pods 16384 2>&1 | tail -1 | line
} 2>&1 | tee $OUT/pipes.log
{
echo "== c4 decision engine"; NHDFIT_LIBRARY=$TL NHDFIT_SEQ_PROF=1 timeout 300 python tools/time_mode_b.py 65536 4096 4 2>&1 | tail -2
echo "== c2"; NHDFIT_LIBRARY=$TL NHDFIT_SEQ_PROF=1 timeout 300 python tools/time_mode_b.py 4096 256 2 2>&1 | tail -2
echo "== c5 shard 2048"; NHDFIT_LIBRARY=$TL NHDFIT_SEQ_PROF=1 timeout 300 python tools/time_mode_b.py 32768 2048 5 2>&1 | tail -2
echo "== c5 shard 16384"; timeout 300 python tools/time_mode_b.py 32768 16384 5 2>&1 | tail -1
} 2>&1 | tee $OUT/modeb.log
timeout 300 python tools/time_findnode.py 2>/dev/null | cut -c1-900 | tee $OUT/findnode.log
