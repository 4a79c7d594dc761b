#!/bin/bash
# CPU only (hipcc cross-compiles): ISA of the two wavefront routines on mode B's chain - commit_node_wave (seq2_kernel.h) and
# map_on_state_wave (seq_kernel.h) - as stand-alone kernels (tools/probe_wave.hip): static instruction counts by class, LDS reads,
# full waits (s_waitcnt lgkmcnt(0) right behind a ds_read = one LDS round trip on the chain), branches.
#   tools/probe_wave_isa.sh [out dir]
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/probe_wave}
mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S "$ROOT/tools/probe_wave.hip" -I"$ROOT/include" -o "$OUT/probe.s" 2>/dev/null
for k in probe_commitEP probe_summary probe_mapE probe_lone_mapE probe_lone_map_shipped; do
    awk -v k="$k" '$0 ~ "^_ZN.*" k ".*: ; @" {f=1} f {print} f && /^\.Lfunc_end/ {f=0}' "$OUT/probe.s" > "$OUT/$k.s"
    total=$(grep -cE '^\s+(v_|s_|ds_|global_|buffer_|flat_|scratch_)' "$OUT/$k.s" || true)
    echo "== $k: $total instructions (static)"
    grep -oE '^\s+[a-z_0-9]+' "$OUT/$k.s" | sed 's/^\s*//' | awk '{split($1,a,"_"); c[a[1]"_"a[2]]++} END{for(k in c) print c[k], k}' | sort -rn | head -12 | tr '\n' ';'; echo
    echo "   ds_read: $(grep -c 'ds_read' "$OUT/$k.s")  ds_write: $(grep -c 'ds_write' "$OUT/$k.s")  full lgkm waits: $(grep -c 's_waitcnt lgkmcnt(0)' "$OUT/$k.s")  branches: $(grep -cE 's_cbranch|s_branch' "$OUT/$k.s")  calls: $(grep -c 's_swappc' "$OUT/$k.s")"
    grep -E "\.(num_vgpr|numbered_sgpr|private_seg_size)," "$OUT/probe.s" | grep "$k" | sed 's/.*\.\(num_vgpr\|numbered_sgpr\|private_seg_size\), /   \1 /' | tr '\n' ' '; echo
done
