#!/bin/bash
# Profile bench.py on the GPU box (run through gpurun).  Usage: tools/gpu_profile.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/: kernel stats (csv) and separate PMC passes, as the MI355X guide prescribes
# (counters never combined with sys/hip/hsa tracing).
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline $*"   # >= 100 steps: the few digest-only / flush launches of k_step must not dilute the per-launch means
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $BENCH > $OUT/stats.log 2>&1
i=0
# QUICK=1: only the instruction-mix pass and the two HBM-traffic passes
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  if [ -n "${QUICK:-}" ] && [ $i -ne 1 ] && [ $i -ne 3 ] && [ $i -ne 4 ]; then continue; fi
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- $BENCH > $OUT/pmc$i.log 2>&1
done
# the summaries stay, the per-dispatch tables go: gpurun copies back at most 64 MiB
python $ROOT/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $OUT/kernel_stats.csv
find $OUT \( -name "*kernel_trace.csv" -o -name "*counter_collection.csv" -o -name "*agent_info.csv" \) -delete
du -sh $OUT
