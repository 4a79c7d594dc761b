#!/bin/bash
# Round 6, GPU call 13: the wavefront form of nhdfit_commit's kernel, no stream query in front of it - parity of every commit path, the
# pod-at-a-time loop call by call, the scheduler-loop leg.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step12
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "commit or attach or scheduler_loop or pending_list or mutators or golden_vectors or deltas or sharing or big_pods or wide" > $OUT/parity_commit.log 2>&1
echo "parity (commit paths) rc=$? $(grep -E 'passed|failed' $OUT/parity_commit.log | tail -1)"; grep -E "^FAILED|^E  " $OUT/parity_commit.log | head
timeout 300 python tools/time_pod_loop.py 16384 1024 | tee $OUT/pod_loop_16384.json
timeout 300 python tools/time_pod_loop.py 65536 512 | tee $OUT/pod_loop_65536.json
timeout 300 python -c "
import json, bench
print(json.dumps(bench.sched_loop(4, 0)))" | tee $OUT/sched_loop_bench_leg.json
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o f -- python $ROOT/tools/time_pod_loop.py 16384 512 > /dev/null 2> $OUT/prof_err.log
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/pod_loop_kernel_stats.csv && head -5 $OUT/pod_loop_kernel_stats.csv | cut -c1-200
rm -rf $OUT/prof
