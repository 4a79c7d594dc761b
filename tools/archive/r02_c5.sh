#!/bin/bash
# config-5 shard: parity of every winner's mapping, the role windows inside the fused launch, and the default (c4) step next to it
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "baseline_configs" 2>&1 | tail -2
NHDFIT_ROLE_TIMES=30 timeout 300 python bench.py --config 5 --nodes-per-gpu 32768 --pods 2048 --steps 60 --no-cpu-baseline --no-pmc --no-extras 2>&1 | grep -E "nhdfit\]|ms_per_step" | cut -c1-330
NHDFIT_ROLE_TIMES=30 timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-extras 2>&1 | grep -E "nhdfit\]|ms_per_step" | cut -c1-330
