#!/bin/bash
# Round 4, last GPU call: XX off by default, nhdfit_sync without the event read-back - parity subset, smoke, the driver's 20-step
# form five times, steady state of config 4 and the config-5 shard.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_last_${1:-a}
mkdir -p $OUT
cd $ROOT
SECONDS=0
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -x -q -k "golden or random_clusters or baseline_configs or edge_cases or pipelined_steps or single_launch or full_size_config4 or commits_and_deltas_between" > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$? seconds=$SECONDS" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error|Error|assert" $OUT/pytest_gpu.log | tail -4
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['steady_state']
print('ms_per_step', round(d['ms_per_step'],5), 'cold', round(d['cold_start']['ms_per_step'],5), 'repeats med', round(d['repeats']['ms_per_step_median'],5), 'steady min/med/max', round(s['ms_per_step_min'],5), round(s['ms_per_step_median'],5), round(s['ms_per_step_max'],5), 'kernel_ms', round(d['roofline']['kernel_ms'],5), 'lds', d['config']['lds_bytes_per_block'], 'placed', d['placed_pods'])"; }
{
for k in 1 2 3 4 5; do echo "== driver form"; timeout 300 python bench.py --no-pmc --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | line; done
echo "== steady c4"; timeout 300 python bench.py --no-pmc --no-extras --no-cpu-baseline --steps 1000 --warmup 200 2>/dev/null | line
echo "== steady c5 shard"; timeout 300 python bench.py --no-pmc --no-extras --no-cpu-baseline --steps 1000 --warmup 200 --config 5 --nodes-per-gpu 32768 --pods 2048 2>/dev/null | line
echo "== steady c5 shard x 16384"; timeout 300 python bench.py --no-pmc --no-extras --no-cpu-baseline --steps 1000 --warmup 200 --config 5 --nodes-per-gpu 32768 --pods 16384 2>/dev/null | line
} 2>&1 | tee $OUT/times.log
echo "seconds=$SECONDS"
