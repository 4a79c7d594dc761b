#!/bin/bash
# Round 4: three pipes as the default for problems that fill the chip, set-state tables in LDS for the drain / single-launch tail:
# parity subset, step times per shape (ship = 3 pipes, tuning build NHDFIT_PIPES=2), the driver's 20-step region three times.
#   gpurun -- bash tools/r04_pipes.sh [tag]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-a}
OUT=$ROOT/gpurun_out/r04_pipes_$TAG
mkdir -p $OUT
cd $ROOT
SECONDS=0
timeout 900 python -m pytest tests -m gpu -x -q -k "golden or random_clusters or edge_cases or pipelined_steps or single_launch or commits_and_deltas_between or single_rank_rccl or scheduler_loop or pending_list" > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$? seconds=$SECONDS" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error|Error|assert" $OUT/pytest_gpu.log | tail -6
B="--no-pmc --no-extras --no-cpu-baseline --steps 1000 --warmup 200"
TL=$ROOT/nhd_amd/libnhdfit_tuning.so
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['steady_state']
print('ms_per_step', round(d['ms_per_step'],5), 'cold', round(d['cold_start']['ms_per_step'],5), 'steady min/med/max', round(s['ms_per_step_min'],5), round(s['ms_per_step_median'],5), round(s['ms_per_step_max'],5), 'kernel_ms', round(d['roofline']['kernel_ms'],5), 'pipes', d['roofline'].get('concurrency'), 'placed', d['placed_pods'])"; }
{
for shape in "--config 4 --nodes-per-gpu 65536 --pods 4096" "--config 3 --nodes-per-gpu 16384 --pods 1024" "--config 5 --nodes-per-gpu 32768 --pods 2048" "--config 5 --nodes-per-gpu 32768 --pods 16384"; do
  echo "== ship (3 pipes) $shape"; timeout 300 python bench.py $B $shape 2>/dev/null | line
  echo "== tuning build NHDFIT_PIPES=2 $shape"; NHDFIT_LIBRARY=$TL NHDFIT_PIPES=2 timeout 300 python bench.py $B $shape 2>/dev/null | line
done
for k in 1 2 3; do echo "== driver form, ship"; timeout 300 python bench.py --no-pmc --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | line; done
for k in 1 2; do echo "== driver form, tuning build NHDFIT_PIPES=2"; NHDFIT_LIBRARY=$TL NHDFIT_PIPES=2 timeout 300 python bench.py --no-pmc --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | line; done
} 2>&1 | tee $OUT/times.log
echo "seconds=$SECONDS"
