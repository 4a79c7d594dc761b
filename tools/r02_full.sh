#!/bin/bash
# full GPU check of the round: parity suite, smoke, default bench, scheduler-loop and mode-B timings
mkdir -p gpurun_out/full
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/full/pytest.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/full/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/full/smoke.log
timeout 900 python bench.py > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err; echo "bench rc=$?" >> gpurun_out/full/bench.err
timeout 600 python tools/time_sched_loop.py > gpurun_out/full/sched_loop.json 2> gpurun_out/full/sched_loop.err
timeout 300 python tools/time_mode_b.py 65536 4096 4 2>&1 | tail -1 > gpurun_out/full/mode_b_c4.json
timeout 300 python tools/time_mode_b.py 65536 4096 5 2>&1 | tail -1 > gpurun_out/full/mode_b_c5.json
timeout 300 python tools/time_findnode.py > gpurun_out/full/findnode.json 2> gpurun_out/full/findnode.err
tail -3 gpurun_out/full/pytest.log; tail -2 gpurun_out/full/smoke.log; tail -c 1500 gpurun_out/full/bench.json; tail -2 gpurun_out/full/bench.err
