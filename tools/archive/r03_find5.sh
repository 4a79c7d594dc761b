#!/bin/bash
# Round 3: re-check of the single-launch find after its clean-up, smoke, CPU-row digest blocks of the single-launch find
out=gpurun_out/r03_find5; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=$PWD/nhd_amd/libnhdfit_tuning.so
timeout 300 python __graft_entry__.py --smoke > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
timeout 600 python -m pytest tests -m gpu -x -q -k "single_launch or create_node_classes or matcher or golden or edge or sched or dropin" > $out/pytest_new.log 2>&1; echo "tests rc=$?"; tail -2 $out/pytest_new.log
echo "== ship"; timeout 200 python tools/time_single_find.py 3:16384,4:65536,5:32768 200 2>/dev/null | tee $out/ship.json
for w in 2 8 16; do echo "== NHDFIT_FIND_WC_PARTS=$w"; NHDFIT_LIBRARY=$T NHDFIT_FIND_WC_PARTS=$w timeout 200 python tools/time_single_find.py 3:16384,4:65536,5:32768 200 2>/dev/null | tee $out/wc_$w.json; done
