#!/bin/bash
# where does the config-5 step go?  role windows inside the fused launch
NHDFIT_ROLE_TIMES=30 timeout 300 python bench.py --config 5 --nodes-per-gpu 32768 --pods 2048 --steps 60 --no-cpu-baseline --no-pmc --no-extras 2>&1 | grep -E "nhdfit\]|ms_per_step" | cut -c1-260
