#!/bin/bash
NHDFIT_SEQ_PROF=1 timeout 300 python tools/time_mode_b.py 65536 4096 4 2>&1 | tail -4
