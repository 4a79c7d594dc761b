#!/bin/bash
# Round 6, GPU call: kernel stats and counter passes of the shipped tree (records carry the C row), summaries only.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step52
mkdir -p $OUT
cd $ROOT
bash tools/gpu_profile.sh r06_final6 > $OUT/profile.log 2>&1
cp $ROOT/gpurun_out/prof_r06_final6/summary.txt $OUT/profile_summary.txt 2>/dev/null
cp $ROOT/gpurun_out/prof_r06_final6/kernel_stats.csv $OUT/rocprof_kernel_stats.csv 2>/dev/null
head -6 $OUT/profile_summary.txt | cut -c1-200
