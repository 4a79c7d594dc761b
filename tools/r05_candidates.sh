#!/bin/bash
# Round 5, first measurement of the emulation-checked candidates of round 4's CPU-only stretch:
#   NHDFIT_CAND_COMMIT_V2  - nhd_amd/csrc/seq2_commit_v2.h: the wavefront commit with the request read once (k_decide's speculators / workers)
#   NHDFIT_CAND_MAP_V2     - nhd_amd/csrc/seq_map_v2.h: a candidate's verification with the NIC walk's uniform operands read once (k_decide)
#   NHDFIT_CAND_FIND1_WAVE - nhd_amd/csrc/find1_wave_map.h: k_find1's mapping tail on one wavefront with the lanes working together
# All are compiled out of libnhdfit.so (the flags are not in nhd_amd/build.py; with them undefined the library is bit-identical to
# the one the last GPU calls of round 4 ran - checked by sha256 when they were wired in).
#   tools/r05_candidates.sh build     here, on CPU (hipcc cross-compiles): nhd_amd/libnhdfit_cand_{commit,chain,find1}.so - they travel with gpurun
#   gpurun --timeout 700 -- 'bash tools/r05_candidates.sh run'      on the GPU box (~5 GPU-minutes): parity subsets + timings per library
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wall -Wno-unused-function"
build_one() {    # name, flags...
    local name=$1; shift
    /opt/rocm/bin/hipcc $FLAGS "$@" nhd_amd/csrc/nhdfit.hip nhd_amd/csrc/wire_digest.cpp -o nhd_amd/libnhdfit_cand_$name.so -ldl && echo "built nhd_amd/libnhdfit_cand_$name.so"
}
if [ "${1:-}" = build ]; then
    build_one commit -DNHDFIT_CAND_COMMIT_V2
    build_one chain -DNHDFIT_CAND_COMMIT_V2 -DNHDFIT_CAND_MAP_V2          # the whole chain of a GPU-less pod: verification + commit
    build_one find1 -DNHDFIT_CAND_FIND1_WAVE
    exit 0
fi
OUT=$ROOT/gpurun_out/r05_candidates
mkdir -p "$OUT"
# --- mode B's chain: every mode-B parity test per candidate library, then decisions/s per shape with the shipped library beside it
for lib in commit chain; do
    export NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_cand_$lib.so
    timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "mode_b or schedule or seq or commit or decide" > $OUT/${lib}_parity.log 2>&1
    echo "$lib candidate parity rc=$?"; tail -n 2 $OUT/${lib}_parity.log
done
for shape in "65536 4096 4" "4096 256 2" "16384 1024 3"; do
    for lib in commit chain; do
        echo "== $shape $lib"; NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_cand_$lib.so timeout 100 python tools/time_mode_b.py $shape 2>&1 | tail -n 1
    done
    echo "== $shape shipped"; NHDFIT_LIBRARY= timeout 100 python tools/time_mode_b.py $shape 2>&1 | tail -n 1
done | tee $OUT/chain_mode_b.log
# --- the one-pod launch's mapping tail: the single-launch / lone-pod parity tests, then the per-call latency, shipped library beside it
export NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_cand_find1.so
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "single or lone or find or golden or edge" > $OUT/find1_parity.log 2>&1
echo "find1 candidate parity rc=$?"; tail -n 2 $OUT/find1_parity.log
timeout 150 python tools/time_single_find.py > $OUT/find1_latency_candidate.json 2>&1; tail -n 3 $OUT/find1_latency_candidate.json
NHDFIT_LIBRARY= timeout 150 python tools/time_single_find.py > $OUT/find1_latency_shipped.json 2>&1; tail -n 3 $OUT/find1_latency_shipped.json
