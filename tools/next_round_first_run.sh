#!/bin/bash
# First GPU call of round 3 (run through gpurun, ~4 GPU-minutes): the round-2 end state re-measured on a fresh box, then
# the block-slot accounting the last experiments of round 2 pointed at.
#   tools/next_round_first_run.sh            -> gpurun_out/next_round/*
#
# Where round 2 stopped: the step kernel is the fit role (22 us alone) plus ~1 us; inside the loop the
# CU's LDS pipe is the limit, around it ~10 us of launch / staging / tail.  A CU holds three 512-thread blocks = 768 slots;
# the grid has 568 (fit) + 8 (choose) + 32 + 32 + 320 (digest) = 960 blocks, so ~190 digest blocks queue behind the first
# finishers.  Freeing the choose role's 120 slots gave -3.5 %.  Candidates, cheapest first:
#   1. shapes / finish on 16 blocks each (256 pods per block)                                    [host constant]
#   2. digest part 0 and the CPU-row parts merged differently (NHDFIT_WC_PARTS=3)               [env, measured 4 > 2 > 1]
#   3. fit items: 8 blocks for the widest tiles too (NHDFIT_XCD_K=1: 512 fit blocks, measured 26.4 vs 26.0 us BEFORE the
#      choose change - re-measure now that 120 slots are free)
#   4. CPU pair table C[smt][free cores 0][free cores 1] for the two-group tiles (34 KB: -43 % LDS bytes for 42 % of the work)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/next_round
mkdir -p $OUT
cd $ROOT
bash tools/r02_full.sh > $OUT/full.log 2>&1; tail -8 $OUT/full.log
B="python bench.py --no-cpu-baseline --no-pmc --no-extras"
line() { python -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value']/1e12, o['ms_per_step'], o['roofline']['kernel_ms'], o['placed_pods'])"; }
for v in "" "NHDFIT_XCD_K=1" "NHDFIT_WC_PARTS=3" "NHDFIT_CHOOSE_LANES=0"; do
  echo "== $v"
  env $v timeout 300 $B 2>&1 | tail -1 | line
  env $v timeout 300 $B 2>&1 | tail -1 | line
done | tee $OUT/slots.log
NHDFIT_ROLE_TIMES=30 timeout 300 $B 2>&1 | grep "nhdfit\]" | tee $OUT/roles.log
