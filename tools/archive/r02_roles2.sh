#!/bin/bash
# role windows inside the fused launch (first start / last end per role, us from the launch's first block)
for s in 60 61; do
NHDFIT_ROLE_TIMES=$s timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-extras 2>&1 | grep -i "role\|nhdfit\]" | grep -v metric | head -8
done
