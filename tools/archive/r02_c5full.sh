#!/bin/bash
# the config-5 shard as one GPU of the 8-GPU run has it: 32 768 nodes x all 16 384 pods
timeout 600 python bench.py --config 5 --nodes-per-gpu 32768 --pods 16384 --steps 60 --no-cpu-baseline --no-pmc --no-extras 2>&1 | tail -1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('c5 32768x16384', o['value']/1e12, o['ms_per_step'], o['placed_pods'], o['roofline']['frac'])"
