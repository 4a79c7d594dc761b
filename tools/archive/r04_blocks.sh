#!/bin/bash
# Round 4: fit blocks per launch (tuning build, NHDFIT_FIT_BLOCKS: plain contiguous ranges, not XCD-aligned) against the ship build.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_blocks_${1:-a}
mkdir -p $OUT
cd $ROOT
B="--no-pmc --no-extras --no-cpu-baseline --steps 1000 --warmup 200"
TL=$ROOT/nhd_amd/libnhdfit_tuning.so
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['steady_state']
print('ms_per_step', round(d['ms_per_step'],5), 'steady min/med/max', round(s['ms_per_step_min'],5), round(s['ms_per_step_median'],5), round(s['ms_per_step_max'],5), 'kernel_ms', round(d['roofline']['kernel_ms'],5), 'placed', d['placed_pods'])"; }
{
echo "== ship c4"; timeout 300 python bench.py $B 2>/dev/null | line
for fb in 256 320 384 512 768; do echo "== NHDFIT_FIT_BLOCKS=$fb c4"; NHDFIT_LIBRARY=$TL NHDFIT_FIT_BLOCKS=$fb timeout 300 python bench.py $B 2>/dev/null | line; done
for k in 2; do echo "== NHDFIT_XCD_K=$k c4"; NHDFIT_LIBRARY=$TL NHDFIT_XCD_K=$k timeout 300 python bench.py $B 2>/dev/null | line; done
for fb in 256 384; do echo "== NHDFIT_FIT_BLOCKS=$fb c5 x 16384"; NHDFIT_LIBRARY=$TL NHDFIT_FIT_BLOCKS=$fb timeout 300 python bench.py $B --config 5 --nodes-per-gpu 32768 --pods 16384 2>/dev/null | line; done
for fb in 128 256; do echo "== NHDFIT_FIT_BLOCKS=$fb c5 x 2048"; NHDFIT_LIBRARY=$TL NHDFIT_FIT_BLOCKS=$fb timeout 300 python bench.py $B --config 5 --nodes-per-gpu 32768 --pods 2048 2>/dev/null | line; done
} 2>&1 | tee $OUT/times.log
