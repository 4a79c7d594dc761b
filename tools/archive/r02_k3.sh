#!/bin/bash
# K3 timings: release + FindNode with deltas vs re-pack; the bench line's new legs
timeout 600 python tools/time_release.py > gpurun_out/release.json 2> gpurun_out/release.err; tail -2 gpurun_out/release.err; cat gpurun_out/release.json
timeout 600 python bench.py --no-cpu-baseline --no-pmc 2> gpurun_out/bench_k3.err | python -c "
import sys, json
o = json.loads(sys.stdin.read())
print(json.dumps({k: o[k] for k in ('value', 'ms_per_step', 'deltas', 'score_only', 'mode_b', 'end_to_end')}))
print(json.dumps(o['other_configs']))"
tail -3 gpurun_out/bench_k3.err
