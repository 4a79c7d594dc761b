#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or random_clusters or edge or pipelined or attach or baseline_configs or rccl or group" 2>&1 | tail -2
bash tools/r02_findprof.sh 2>&1 | tail -12
timeout 300 python tools/time_findnode.py 2>/dev/null | cut -c1-1500
timeout 300 python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import sys, json
o = json.loads(sys.stdin.read()); print(o['value']/1e12, o['ms_per_step'], o['end_to_end'])"
