#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_sweep
mkdir -p $OUT
cd $ROOT
TL=$ROOT/nhd_amd/libnhdfit_tuning.so
B="python bench.py --no-cpu-baseline --no-pmc --no-extras --steps 600 --warmup 1500"
line() { python -c "import sys,json; o=json.loads(sys.stdin.read()); r=o['roofline']; print(o['value']/1e12, o['ms_per_step'], r['kernel_ms'], o['placed_pods'], o['repeats']['ms_per_step_median'])"; }
{
echo "== ship (new defaults)"; timeout 300 $B 2>&1 | tail -1 | line
echo "== ship (new defaults)"; timeout 300 $B 2>&1 | tail -1 | line
echo "== tuning new defaults"; NHDFIT_LIBRARY=$TL timeout 300 $B 2>&1 | tail -1 | line
echo "== tuning WC_PARTS=3"; NHDFIT_LIBRARY=$TL NHDFIT_WC_PARTS=3 timeout 300 $B 2>&1 | tail -1 | line
echo "== tuning WC_PARTS=1"; NHDFIT_LIBRARY=$TL NHDFIT_WC_PARTS=1 timeout 300 $B 2>&1 | tail -1 | line
echo "== tuning FIT_BLOCKS=384"; NHDFIT_LIBRARY=$TL NHDFIT_FIT_BLOCKS=384 timeout 300 $B 2>&1 | tail -1 | line
echo "== ship c5 shard x 16384"; timeout 300 $B --config 5 --nodes-per-gpu 32768 --pods 16384 2>&1 | tail -1 | line
echo "== ship c3"; timeout 300 $B --config 3 --nodes-per-gpu 16384 --pods 1024 2>&1 | tail -1 | line
} 2>&1 | tee $OUT/sweep2.log
