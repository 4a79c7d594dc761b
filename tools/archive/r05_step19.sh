#!/bin/bash
# Round 5, GPU call 19: staging in four pieces (transfer beside the gather), no queries of idle side streams.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_step19
mkdir -p $OUT
cd $ROOT
SECONDS=0
timeout 900 python -m pytest tests -m gpu -x -q -k "single_launch or pipelined or baseline_configs or random_clusters or golden_vectors or attach or deltas or commits_and" > $OUT/pytest.log 2>&1
echo "pytest rc=$? seconds=$SECONDS"; tail -3 $OUT/pytest.log | cut -c1-300
timeout 200 python tools/time_batch_find.py "4:65536:4096,2:0:0,3:0:0,5:32768:2048,5:32768:16384" > $OUT/batch_find.json 2> $OUT/batch_find.err
echo "batch rc=$? seconds=$SECONDS"; cat $OUT/batch_find.json
NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so NHDFIT_FIND_PROF=1 timeout 100 python tools/time_batch_find.py "4:65536:4096,2:0:0" > $OUT/batch_find_tuning.json 2> $OUT/batch_find_phases.log
grep "P=4096" $OUT/batch_find_phases.log | tail -11; grep "P=256" $OUT/batch_find_phases.log | tail -11
timeout 100 python tools/time_single_find.py "4:65536" > $OUT/single.json 2>/dev/null; cat $OUT/single.json
