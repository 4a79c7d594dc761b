#!/bin/bash
# Round 5, GPU call 16: batch find - no needless stream wait for small shapes, digest cut four ways, staging phases.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_step16
mkdir -p $OUT
cd $ROOT
SECONDS=0
timeout 600 python -m pytest tests -m gpu -x -q -k "single_launch or pipelined" > $OUT/pytest.log 2>&1
echo "pytest rc=$? seconds=$SECONDS"; tail -3 $OUT/pytest.log | cut -c1-300
timeout 200 python tools/time_batch_find.py "4:65536:4096,2:0:0,3:0:0,5:32768:2048" > $OUT/batch_find.json 2> $OUT/batch_find.err
echo "batch rc=$? seconds=$SECONDS"; cat $OUT/batch_find.json
for wc in 2 4; do
NHDFIT_FIND_WC_PARTS=$wc NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_tuning.so NHDFIT_FIND_PROF=1 timeout 100 python tools/time_batch_find.py "4:65536:4096,2:0:0" > $OUT/batch_find_tuning_$wc.json 2> $OUT/batch_find_phases_$wc.log
echo "== wc_parts $wc"; cat $OUT/batch_find_tuning_$wc.json
grep "P=4096" $OUT/batch_find_phases_$wc.log | tail -12; grep "P=256" $OUT/batch_find_phases_$wc.log | tail -12
done
