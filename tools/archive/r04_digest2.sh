#!/bin/bash
# Round 4: digest by pool type + CPU rows per row width + signatures in use only; four pipes - whole GPU suite, then step times
# (ship build), then the same with the kernel arguments in device memory (HIP_FORCE_DEV_KERNARG=1: experiment).
#   gpurun -- bash tools/r04_digest2.sh [tag]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-a}
OUT=$ROOT/gpurun_out/r04_digest2_$TAG
mkdir -p $OUT
cd $ROOT
SECONDS=0
if [ "${2:-}" = "quick" ]; then SEL="-k golden_or_random_clusters_or_baseline_configs_or_edge_cases_or_pipelined_steps_or_mode_b_at_baseline"; SEL=$(echo $SEL | sed "s/_or_/ or /g"); timeout 900 python -m pytest tests -m gpu -x -q -k "golden or random_clusters or baseline_configs or edge_cases or pipelined_steps or mode_b_at_baseline" > $OUT/pytest_gpu.log 2>&1; else timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; fi
echo "pytest rc=$? seconds=$SECONDS" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error|Error|assert" $OUT/pytest_gpu.log | tail -6
B="--no-pmc --no-extras --no-cpu-baseline --steps 1000 --warmup 200"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['steady_state']
print('ms_per_step', round(d['ms_per_step'],5), 'steady min/med/max', round(s['ms_per_step_min'],5), round(s['ms_per_step_median'],5), round(s['ms_per_step_max'],5), 'kernel_ms', round(d['roofline']['kernel_ms'],5), 'placed', d['placed_pods'])"; }
{
for shape in "--config 5 --nodes-per-gpu 32768 --pods 16384" "--config 5 --nodes-per-gpu 32768 --pods 2048" "--config 4 --nodes-per-gpu 65536 --pods 4096" "--config 2 --nodes-per-gpu 4096 --pods 256"; do
  echo "== ship $shape"; timeout 300 python bench.py $B $shape 2>/dev/null | line
done
echo "== single find"; timeout 200 python tools/time_single_find.py | tail -1
echo "== mode B c4"; timeout 300 python tools/time_mode_b.py 65536 4096 4 2>&1 | tail -1
TL=$ROOT/nhd_amd/libnhdfit_tuning.so
for fb in 512 1024 2048; do echo "== tuning build NHDFIT_FIT_BLOCKS=$fb c5 x 16384"; NHDFIT_LIBRARY=$TL NHDFIT_FIT_BLOCKS=$fb timeout 300 python bench.py $B --config 5 --nodes-per-gpu 32768 --pods 16384 2>/dev/null | line; done
echo "== tuning build NHDFIT_ALL_SIGS=1 c5 x 16384"; NHDFIT_LIBRARY=$TL NHDFIT_ALL_SIGS=1 timeout 300 python bench.py $B --config 5 --nodes-per-gpu 32768 --pods 16384 2>/dev/null | line
} 2>&1 | tee $OUT/times.log
echo "seconds=$SECONDS"
