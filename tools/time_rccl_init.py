#!/usr/bin/env python3
"""How long does the single-rank RCCL communicator take to come up (with / without the env hints bench.py sets)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "hints":
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_IB_DISABLE", "1")
from nhd_amd.engine import Engine
t0 = time.perf_counter(); e = Engine(0); t1 = time.perf_counter()
uid = e.unique_id(); t2 = time.perf_counter()
e.comm_init(1, 0, uid); t3 = time.perf_counter()
print({"mode": sys.argv[1:] or ["plain"], "create_s": round(t1 - t0, 2), "unique_id_s": round(t2 - t1, 2), "comm_init_s": round(t3 - t2, 2)})
