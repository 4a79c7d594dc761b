#!/bin/bash
# Round 4: the digest's signature walk (one record load + v_readlane; then signatures by pool type) and four pipes for small problems
# - parity of the digest- and pipeline-sensitive tests,
# then the step time of the config-5 shard against all 16 384 pods and of config 4, and the role windows inside one launch.
#   gpurun -- bash tools/r04_digest.sh [tag]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-a}
OUT=$ROOT/gpurun_out/r04_digest_$TAG
mkdir -p $OUT
cd $ROOT
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or random_clusters or baseline_configs or node_classes or single_launch_find_equals or edge_cases or pipelined_steps or commits_and_deltas_between or full_size_config4 or scheduler_loop or wide_nodes_at_scale" > $OUT/pytest_digest.log 2>&1
echo "pytest rc=$? seconds=$SECONDS" | tee -a $OUT/pytest_digest.log
grep -E "passed|failed|error|Error|assert" $OUT/pytest_digest.log | tail -6
B="--no-pmc --no-extras --no-cpu-baseline --steps 1000 --warmup 200"
TL=$ROOT/nhd_amd/libnhdfit_tuning.so
{
for shape in "--config 5 --nodes-per-gpu 32768 --pods 16384" "--config 5 --nodes-per-gpu 32768 --pods 2048" "--config 4 --nodes-per-gpu 65536 --pods 4096" "--config 2 --nodes-per-gpu 4096 --pods 256" "--config 3 --nodes-per-gpu 16384 --pods 1024"; do
  echo "== ship $shape"
  timeout 300 python bench.py $B $shape 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['steady_state']
print('ms_per_step', round(d['ms_per_step'],5), 'steady min/med/max', round(s['ms_per_step_min'],5), round(s['ms_per_step_median'],5), round(s['ms_per_step_max'],5), 'kernel_ms', round(d['roofline']['kernel_ms'],5), 'digest_kernel_ms', d['roofline'].get('digest_kernel_ms'), 'placed', d['placed_pods'])"
  echo "== role windows (tuning build) $shape"
  NHDFIT_LIBRARY=$TL NHDFIT_ROLE_TIMES=150 timeout 300 python bench.py --no-pmc --no-extras --no-cpu-baseline --no-settle --steps 300 --warmup 20 $shape 2>&1 >/dev/null | grep "role" | head -12
done
} 2>&1 | tee $OUT/digest_times.log
echo "seconds=$SECONDS"
