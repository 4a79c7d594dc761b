#!/bin/bash
# config-5 shard after the pruned NIC-choice walk: parity of every winner's mapping, then the role windows inside the fused launch
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "baseline_configs and 5-32768" 2>&1 | tail -3
NHDFIT_ROLE_TIMES=30 timeout 300 python bench.py --config 5 --nodes-per-gpu 32768 --pods 2048 --steps 60 --no-cpu-baseline --no-pmc --no-extras 2>&1 | grep -E "nhdfit\]|ms_per_step" | cut -c1-400
