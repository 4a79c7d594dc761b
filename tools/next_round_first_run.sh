#!/bin/bash
# First GPU call of round 5 (run through gpurun, ~9 GPU-minutes): the round-4 end state re-measured on a fresh box.
#   tools/next_round_first_run.sh            -> gpurun_out/next_round/*
#
# Where round 4 stopped (DESIGN.md sections 4 and 6, profiles/r04/README.md):
#   * mode A: 15.5-15.7 us per step of 65 536 nodes x 4 096 pods in steady state (two pipes, pair rows C / XX in LDS for the one-
#     and two-group tiles), 18.6-19.2 us per step in the driver's 20-step form (57 us of each short region are its end: the last
#     launches' tail, the drain launch k_map_tiles, the waits), 20.7 as the process's first GPU work.  Counters per launch's worth:
#     HBM 0.37, LDS 0.37-0.43 (47 % of it bank conflicts), VALU 0.37-0.43 -> latency.  The host's enqueue costs ~11 us per step:
#     configs 2 and 3 sit on it (11.5 / 14.7 us), config 4 is 4 us above it - a hipGraph over a cycle of steps (busy_from read from
#     a device cell instead of the argument block) is the next lever once the GPU side drops further.
#     Untried against the LDS bank conflicts: order each fit block's nodes by (class pair, free-core pair) in k_xrecords (the lane <->
#     node assignment inside a block is free: verdict words are stored by node index) so that a wavefront's row fetches hit a few
#     addresses (broadcasts) instead of sixteen bank groups at random.
#     Candidates inside the fit role: the three-group tiles still sweep six rows (D = 31: C does not fit; XX with dense per-NUMA class
#     ranks would be 6.4 KB), two tiles per block (record fetch + address arithmetic shared).
#   * digest: signatures by pool type, R rows only for signatures in use, CPU rows per tile width: config 5 shard 21.7 -> 10.9 us
#     (three pipes there), x 16 384 pods 58.7 -> 53 us: 2 300 fit blocks lead the grid, the 768 digest blocks start 40 us in and run
#     50 us - a tile's signature block is still ONE block.
#   * mode B: 0.81-0.85 M decisions/s at config 4 (speculate + retire), 0.13 M at config 2 (every pod GPU-less, piling onto the same
#     nodes: verify ~3 us -> commit ~4 us is a real chain; next: the commit's summary - counts, signature ids - first, the core picks on a
#     second wavefront).
#     What the chain executes can be read without a GPU: tools/probe_wave_isa.sh (commit_node_wave: 1 385 instructions static, 40 LDS
#     round trips with a full wait - the request's byte fields one ds_read_u8 at a time, 5-6 per processing group; map_on_state_wave:
#     1 601 / 37) and checked without one: tests/harness/wave_emul.cpp runs the routines' SOURCE TEXT on 64 emulated lanes against the
#     scalar forms (tests/test_wave_commit_emulation.py) - rewrite there first (request fields from a lane-distributed register through
#     v_readlane instead of LDS bytes - nhd_amd/csrc/seq2_commit_v2.h is that for the commit, emulation-checked, wired in behind -DNHDFIT_CAND_COMMIT_V2 (off: the shipped binary is bit-identical): tools/r05_candidates.sh build, then gpurun ... run; the summary of a commit - free-core / GPU counts, hugepages, NIC classes - published before the
#     core picks so that the next pod's verification starts ~3.7 us earlier), then measure with NHDFIT_SEQ_PROF=1.
#   * single calls (candidate, emulation-checked, behind -DNHDFIT_CAND_FIND1_WAVE: nhd_amd/csrc/find1_wave_map.h - k_find1's mapping tail through
#     map_on_state_wave; measured by the same tools/r05_candidates.sh run):
#   * single calls: nhdfit_find one pod 26 / 28 / 35 / 33 us (4 096 / 16 384 / 65 536 nodes / c5 shard); HIP_FORCE_DEV_KERNARG=1 changes
#     nothing; nhdfit_find with 4 096 pods 0.20 ms (host copies of 512 KB of requests on both sides of three launches).
#   * pods beyond the table pass (5..8 processing groups, hugepage requests > 1 022 GiB): nhdfit_big_req through the general path
#     (big_kernel.h), 0.16-1.9 ms per pod at 65 536 nodes, up to 25 ms on the config-5 shard (odometer walk, totals pre-filter,
#     NIC stage remembered per NUMA node and group set: k_big_eval mean 1.0 ms; what is left is k_big_map, one thread per winner,
#     2.2 ms - its NIC searches over every assignment and the set model could spread over a wavefront if it ever matters).
#     tools/time_big_find.py + tools/r04_big_prof.sh re-measure it (30 s of box time).
#   * tests/test_kernel_resources.py guards the step kernels' scratch / VGPR budget on every CPU run - look at it first when a step
#     time jumps.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/r04_full.sh next
mkdir -p gpurun_out/next_round && cp -r gpurun_out/r04_full_next/* gpurun_out/next_round/
TL=$ROOT/nhd_amd/libnhdfit_tuning.so
NHDFIT_LIBRARY=$TL NHDFIT_SEQ_PROF=1 timeout 300 python tools/time_mode_b.py 65536 4096 4 2>&1 | tail -3 | tee gpurun_out/next_round/mode_b_phases.log
timeout 200 python tools/time_single_find.py | tee gpurun_out/next_round/single_find_latency.json
timeout 100 python tools/time_big_find.py | tee gpurun_out/next_round/big_find_latency.json
