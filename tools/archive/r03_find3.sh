#!/bin/bash
# Round 3, single-launch find with the one-staging mapping tail: parity, latency, phases
out=gpurun_out/r03_find3; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=$PWD/nhd_amd/libnhdfit_tuning.so
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "single_launch" > $out/pytest_new.log 2>&1; echo "parity rc=$?"; tail -2 $out/pytest_new.log
echo "== ship" ; timeout 200 python tools/time_single_find.py > $out/ship.json 2> $out/ship.err; cat $out/ship.json
echo "== phases (tuning build)"; NHDFIT_LIBRARY=$T NHDFIT_ROLE_TIMES=0 timeout 200 python tools/time_single_find.py 4:65536,5:32768 3 > $out/phases.log 2>&1; grep -v "^\[{" $out/phases.log | head -24; echo ...; grep -v "^\[{" $out/phases.log | tail -12
for nb in 16 64; do echo "== NHDFIT_FIND_BLOCKS=$nb"; NHDFIT_LIBRARY=$T NHDFIT_FIND_BLOCKS=$nb timeout 200 python tools/time_single_find.py 4:65536,3:16384 100 2>/dev/null | tee $out/blocks_$nb.json; done
timeout 300 python tools/time_findnode.py > $out/findnode_latency.json 2> $out/findnode.err; echo "findnode rc=$?"; cat $out/findnode_latency.json
