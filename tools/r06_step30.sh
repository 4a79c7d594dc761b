#!/bin/bash
# Round 6, GPU call: instruction-cache behaviour of mode B's decision engine (k_decide: 87 KB of code, one 64 KB cache per two CUs).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step30
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc -o p -- python $ROOT/tools/time_mode_b.py 65536 4096 4 > $OUT/run.log 2>&1
python $ROOT/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
find $OUT \( -name "*kernel_trace.csv" -o -name "*counter_collection.csv" -o -name "*agent_info.csv" \) -delete
grep -A 9 "k_decide" $OUT/summary.txt | head -12 | cut -c1-120
