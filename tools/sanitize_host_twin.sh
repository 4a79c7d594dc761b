#!/bin/bash
# CPU only: the kernels' shared arithmetic (fit_core.h, winner_map.h, seq_core.h, commit_core.h, wide_core.h, set_states.h,
# dict_stream.h - what tests/harness compiles for the host) built with AddressSanitizer + UndefinedBehaviorSanitizer and run
# under the host-twin tests and the edge-of-format soak: an out-of-range table index, a shift by the operand's width or a
# signed overflow in those headers is the same defect on the device, where nothing would report it.
#   tools/sanitize_host_twin.sh [soak seeds]
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
SAN=/tmp/_host_harness_san.so
g++ -O1 -g -std=c++17 -ffp-contract=off -shared -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
    tests/harness/host_harness.cpp -o $SAN
export NHD_HOST_HARNESS_SO=$SAN
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libstdc++.so.6)"
export ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1
python tools/soak_extreme.py "${1:-200}" 700 2>&1 | grep -v "^mode B\|is not mirrored" | tail -n 1
python -m pytest -q -x -m "not gpu" -p no:cacheprovider tests/test_core_vs_oracle.py tests/test_mode_b_core.py tests/test_commit_core.py \
    tests/test_delta_core.py tests/test_wide_core.py tests/test_big_core.py tests/test_lone_pod_core.py tests/test_pyset_emulation.py \
    tests/test_nic_choice_pruning.py tests/test_matcher_host_logic.py tests/test_group_engine.py tests/test_format_edges.py 2>&1 | tail -n 2
python tools/fuzz_wire_sanitized.py 3000 | tail -n 1
