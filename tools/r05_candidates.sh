#!/bin/bash
# Round 5, first measurement of the two emulation-checked candidates of round 4's CPU-only stretch:
#   NHDFIT_CAND_COMMIT_V2  - nhd_amd/csrc/seq2_commit_v2.h: the wavefront commit with the request read once (k_decide's speculators / workers)
#   NHDFIT_CAND_FIND1_WAVE - nhd_amd/csrc/find1_wave_map.h: k_find1's mapping tail on one wavefront with the lanes working together
# Both are compiled out of libnhdfit.so (the flags are not in nhd_amd/build.py; with them undefined the library is bit-identical to
# the one the last GPU calls of round 4 ran - checked by sha256 when they were wired in).
#   tools/r05_candidates.sh build     here, on CPU (hipcc cross-compiles): nhd_amd/libnhdfit_cand_{commit,find1}.so - they travel with gpurun
#   gpurun --timeout 600 -- 'bash tools/r05_candidates.sh run'      on the GPU box (~4 GPU-minutes): parity subsets + timings per library
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wall -Wno-unused-function"
if [ "${1:-}" = build ]; then
    for v in commit:-DNHDFIT_CAND_COMMIT_V2 find1:-DNHDFIT_CAND_FIND1_WAVE; do
        /opt/rocm/bin/hipcc $FLAGS ${v#*:} nhd_amd/csrc/nhdfit.hip nhd_amd/csrc/wire_digest.cpp -o nhd_amd/libnhdfit_cand_${v%%:*}.so -ldl && echo "built nhd_amd/libnhdfit_cand_${v%%:*}.so"
    done
    exit 0
fi
OUT=$ROOT/gpurun_out/r05_candidates
mkdir -p "$OUT"
# --- the commit candidate: every mode-B parity test, then decisions/s per shape with the phase sums of the shipped library beside it
export NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_cand_commit.so
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "mode_b or schedule or seq or commit or decide" > $OUT/commit_parity.log 2>&1
echo "commit candidate parity rc=$?"; tail -n 2 $OUT/commit_parity.log
for shape in "65536 4096 4" "4096 256 2" "16384 1024 3"; do
    echo "== $shape candidate"; timeout 100 python tools/time_mode_b.py $shape 2>&1 | tail -n 1
    echo "== $shape shipped";   NHDFIT_LIBRARY= timeout 100 python tools/time_mode_b.py $shape 2>&1 | tail -n 1
done | tee $OUT/commit_mode_b.log
# --- the one-pod launch's mapping tail: the single-launch / lone-pod parity tests, then the per-call latency, shipped library beside it
export NHDFIT_LIBRARY=$ROOT/nhd_amd/libnhdfit_cand_find1.so
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "single or lone or find or golden or edge" > $OUT/find1_parity.log 2>&1
echo "find1 candidate parity rc=$?"; tail -n 2 $OUT/find1_parity.log
timeout 150 python tools/time_single_find.py > $OUT/find1_latency_candidate.json 2>&1; tail -n 3 $OUT/find1_latency_candidate.json
NHDFIT_LIBRARY= timeout 150 python tools/time_single_find.py > $OUT/find1_latency_shipped.json 2>&1; tail -n 3 $OUT/find1_latency_shipped.json
