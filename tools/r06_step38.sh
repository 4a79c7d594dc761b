#!/bin/bash
# Round 6, GPU call: where mode B's fetcher spends its time (tuning build: list entries / windows, the ring, issue + park), config 4 and the config-5 shard x 16 384 pods.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_step38
mkdir -p $OUT
cd $ROOT
for s in "65536 4096 4" "32768 16384 5" "16384 1024 3"; do
  NHDFIT_SEQ_PROF=1 NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so timeout 300 python tools/time_mode_b.py $s 2>&1 | grep -E "k_decide|decisions_per_s" | tail -7 >> $OUT/mode_b_fetcher_phases.log
done
cat $OUT/mode_b_fetcher_phases.log | cut -c1-300
