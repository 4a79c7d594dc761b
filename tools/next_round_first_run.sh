#!/bin/bash
# First GPU call of round 4 (run through gpurun, ~8 GPU-minutes): the round-3 end state re-measured on a fresh box.
#   tools/next_round_first_run.sh            -> gpurun_out/next_round/*
#
# Where round 3 stopped (DESIGN.md sections 4 and 6):
#   * mode A: 16.0-16.1 us per step of 65 536 nodes x 4 096 pods (two pipes on two streams, 8 fit blocks per tile, 2 CPU-row
#     digest blocks); counters per launch: HBM 0.31, LDS 0.43 (43 % of it bank conflicts), VALU 0.34 of peak -> latency / LDS.
#     Candidates: the pair table C[smt][free cores 0][free cores 1] for the two-group tiles (one row fetch instead of four),
#     NHDFIT_FIT_BLOCKS=384 (measured -1.5 % in the tuning build), a replicated / skewed WC table against the conflicts.
#   * mode B: 396-445 k decisions/s (decision engine, two driver wavefronts); ~8.5 us per committed GPU-less pod on a driver
#     (mapping ~3, commit ~4.2 on the lanes).  Candidates: the commit split over two wavefronts (core batches / signature keys),
#     the queue entry carrying the patch state (no coherent reload by the patcher).
#     Measured on the oracle's decisions (tools/mode_b_conflicts.py, profiles/r03/mode_b_conflict_structure.log): at c4 only 71 of the 396
#     GPU-less pods that land on nodes with GPUs meet a node an earlier one took (median 93 pods earlier) - that 4.2 ms chain can be
#     walked by several wavefronts speculatively with a per-node claim (atomicMin of the pod position) and a re-verify for the displaced.
#   * single calls: nhdfit_find for one pod is ONE launch without tables (k_find1): 27 / 30 / 37 us at 4 096 / 16 384 / 65 536
#     nodes, 35 on the c5 shard; 2..64 pods: k_find (digest -> fit -> mapping in one launch), 61-86 us.  Open: k_find1's block
#     count was never swept (NHDFIT_FIND_BLOCKS in the tuning build), its mapping tail is 8-17 us on one lane.
#   * limits the product still degrades on: > 2 NUMA nodes, > 64 physical cores per socket, > 4 groups.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/r03_full.sh
mkdir -p gpurun_out/next_round && cp -r gpurun_out/r03_full/* gpurun_out/next_round/
TL=$ROOT/nhd_amd/libnhdfit_tuning.so
NHDFIT_LIBRARY=$TL NHDFIT_SEQ_PROF=1 timeout 300 python tools/time_mode_b.py 65536 4096 4 2>&1 | tail -3 | tee gpurun_out/next_round/mode_b_phases.log
timeout 200 python tools/time_single_find.py | tee gpurun_out/next_round/single_find_latency.json
NHDFIT_LIBRARY=$TL NHDFIT_ROLE_TIMES=0 timeout 200 python tools/time_single_find.py 4:65536,5:32768 3 2>&1 | grep -v '^\[{' | tail -16 | tee gpurun_out/next_round/single_find_phases.log
