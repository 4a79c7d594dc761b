#!/bin/bash
# Round 3: uploads between pipelined steps (new node classes), single-launch find, FindNode and scheduler-loop rates
out=gpurun_out/r03_find4; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "single_launch or create_node_classes or pipeline" > $out/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -3 $out/pytest_new.log
timeout 300 python tools/time_findnode.py > $out/findnode_latency.json 2> $out/findnode.err; echo "findnode rc=$?"; cat $out/findnode_latency.json
timeout 400 python tools/time_sched_loop.py > $out/sched_loop.json 2> $out/sched.err; echo "sched rc=$?"; cat $out/sched_loop.json
