#!/bin/bash
# rocprofv3 passes for the current kernel.  Usage: tools/r02_prof.sh <tag> [QUICK]
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline"
# (a) fused step kernel as benchmarked
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $BENCH > $OUT/stats.log 2>&1
# (b) side roles and fit role as separate launches
NHDFIT_SPLIT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/split -o s -- $BENCH > $OUT/split.log 2>&1
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
  i=$((i+1))
  NHDFIT_SPLIT=${SPLITPMC:-} timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- $BENCH > $OUT/pmc$i.log 2>&1
done
python $ROOT/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
for f in $(find $OUT/split -name "*kernel_stats.csv"); do echo "== split"; cat $f; done >> $OUT/summary.txt
cat $OUT/summary.txt | head -120
