#!/bin/bash
# does device-resident kernarg memory shorten the step?  (HIP_FORCE_DEV_KERNARG, read by the HIP runtime at initialisation)
for v in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$v"
  HIP_FORCE_DEV_KERNARG=$v timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-extras 2>&1 | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value']/1e12, o['ms_per_step'], o['roofline']['kernel_ms'], o['roofline']['frac'])"
done
HIP_FORCE_DEV_KERNARG=1 timeout 300 python tools/time_findnode.py 2>/dev/null | cut -c1-700
HIP_FORCE_DEV_KERNARG=1 timeout 300 python tools/time_mode_b.py 65536 4096 4 2>&1 | tail -1 | cut -c1-200
