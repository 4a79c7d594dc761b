#!/bin/bash
# Round 3, single-launch find: where the time goes at 65 536 nodes (device-clock phases, fit block counts)
out=gpurun_out/r03_find2; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=$PWD/nhd_amd/libnhdfit_tuning.so
echo "== ship" ; timeout 200 python tools/time_single_find.py > $out/ship.json 2> $out/ship.err; cat $out/ship.json
echo "== phases (tuning build, default blocks)"; NHDFIT_LIBRARY=$T NHDFIT_ROLE_TIMES=0 timeout 200 python tools/time_single_find.py 4:65536,3:16384 6 2>&1 | grep -v "^\[{" | tail -60 > $out/phases.log; tail -44 $out/phases.log
for nb in 32 64 128 512; do echo "== NHDFIT_FIND_BLOCKS=$nb"; NHDFIT_LIBRARY=$T NHDFIT_FIND_BLOCKS=$nb timeout 200 python tools/time_single_find.py 4:65536,5:32768 100 2>/dev/null | tee $out/blocks_$nb.json; done
echo "== phases at 64 blocks"; NHDFIT_LIBRARY=$T NHDFIT_FIND_BLOCKS=64 NHDFIT_ROLE_TIMES=0 timeout 200 python tools/time_single_find.py 4:65536 6 2>&1 | grep -v "^\[{" | tail -16
