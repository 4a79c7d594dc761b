#!/bin/bash
# Round 3, last GPU minutes: the table-free launch for a lone pod (k_find1) as the default of a ship-flag build
# (libnhdfit_lone.so = libnhdfit.so + -DNHDFIT_LONE_POD_DEFAULT): the whole GPU suite through it, then its latency and phases.
out=gpurun_out/r03_lone; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
L=$PWD/nhd_amd/libnhdfit_lone.so; T=$PWD/nhd_amd/libnhdfit_tuning.so
echo "== latency (lone default)"; NHDFIT_LIBRARY=$L timeout 120 python tools/time_single_find.py 2>$out/lat.err | tee $out/single_find_lone.json
echo "== phases"; NHDFIT_LIBRARY=$T NHDFIT_LONE_POD=1 NHDFIT_ROLE_TIMES=0 timeout 120 python tools/time_single_find.py 4:65536,5:32768 2 > $out/phases.log 2>&1; grep "nhdfit" $out/phases.log | grep -v "^\[{" | head -8; grep "nhdfit" $out/phases.log | tail -4
SECONDS=0; NHDFIT_LIBRARY=$L timeout 400 python -m pytest tests -m gpu -x -q > $out/pytest_lone.log 2>&1; echo "pytest(lone) rc=$? seconds=$SECONDS"; grep -E "passed|failed|error" $out/pytest_lone.log | tail -3; grep -E "^E " $out/pytest_lone.log | head -5
