#!/usr/bin/env python3
"""Benchmark of the node filter-and-score hot path (contract: see the task statement / DESIGN.md section 4).

    python bench.py --gpus 1 --steps 200 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One step = one pass of the whole path over one batch of pending pods with the cluster mirror and the
request records already resident in HBM: request digest -> fit+score over every (pod, node) pair of
this rank's node shard -> [RCCL all-reduce(max) of the packed scores] -> winner mapping.  The library issues
ONE kernel launch per step (k_step): its grid carries the fit role of this step together with the digest of
the next step and the mapping roles of the previous three (software pipeline, nhd_amd/csrc/nhdfit.hip).

Workload (config.workload): BASELINE.json's 64k-node case - config 4's cluster (CPU+GPU+NIC, PCI
locality for half the pods) with 65 536 nodes PER GPU and 4 096 pending pods; with N GPUs the node axis
is sharded (weak scaling: N x 65 536 nodes in total), pods are replicated, one all-reduce picks the winners.
value = pod x node evaluations per second over the whole job.

BASELINE.json's own multi-GPU shapes (strong scaling: the cluster is fixed, the shards shrink):
    ... bench.py --gpus 4 --config 4 --total-nodes 65536   --pods 4096      (config 4: 16 384 nodes per GPU)
    ... bench.py --gpus 8 --config 5 --total-nodes 262144  --pods 16384     (config 5: 32 768 nodes per GPU)
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X spec (guides/MI355X_MICROARCH.md, "Chip-level parameters"); ~6300 achievable
VALU_NS_PER_WAVE_INST = 1.2  # measured full-rate wave64 VALU issue per SIMD (tools/valu_calib.hip, profiles/r02/valu_calib.jsonl)
SIMDS = 1024
CUS = 256
CLOCK_HZ = 2.4e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--config", type=int, default=4, help="BASELINE config whose cluster/pod mix is generated")
    ap.add_argument("--nodes-per-gpu", type=int, default=65536)
    ap.add_argument("--total-nodes", type=int, default=0, help="strong scaling: fixed cluster size, shards of total/N nodes")
    ap.add_argument("--pods", type=int, default=4096)
    ap.add_argument("--cpu-sample-pods", type=int, default=1024, help="pods timed on the CPU port (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 counter passes (HBM traffic of the step kernel)")
    ap.add_argument("--no-extras", action="store_true", help="skip the mode-B and end-to-end legs after the timed region")
    ap.add_argument("--settle-ms", type=float, default=30.0, help="how long the device is kept busy with the benchmarked step in front of the headline region")
    ap.add_argument("--settle-regions", type=int, default=0, help="experiment: untimed W + K regions (each closed by a sync) at the end of the settling phase")
    ap.add_argument("--no-settle", action="store_true", help="time the W + K steps as the first GPU work of the process only (no clock settling in front of the headline region)")
    args = ap.parse_args()

    # stdout carries exactly one line: the result.  Native libraries are chatty on it (gloo's "[Gloo] Rank 0 is
    # connected ...", librccl's version banner at communicator creation), so file descriptor 1 is pointed at
    # stderr for the duration of the run and the JSON line goes to a private duplicate of the real stdout.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    strong = args.total_nodes > 0
    if strong:
        args.nodes_per_gpu = (args.total_nodes + world - 1) // world

    dist = None
    if world > 1:
        import torch.distributed as dist          # control plane only (rendezvous, barrier, max-reduce of the time)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # (a rank that fails inside a rank-to-rank exchange must not leave the others waiting for the default half hour)
        dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))

    from nhd_amd import pack
    from nhd_amd.engine import Engine, winner_index
    from workload import planes, refmodel, synth

    n_total = args.total_nodes if strong else args.nodes_per_gpu * world
    lo, hi = rank * args.nodes_per_gpu, min(n_total, (rank + 1) * args.nodes_per_gpu)
    spec_all = synth.make_cluster(args.config, n_nodes=n_total)
    spec = spec_all.shard(lo, hi)
    pods, pod_groups = synth.make_pods(args.config, n_pods=args.pods)
    tops = [refmodel.make_topology(s) for s in pods]

    pk = pack.Packer()
    table = planes.planes_from_spec(pk, spec)
    reqs = pk.digest_many(tops, pod_groups)
    pk.close_signatures()
    eng = Engine(local_rank)
    eng.set_dictionary(pk)
    eng.upload(table, global_base=lo)
    if world > 1:
        uid = [eng.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        eng.comm_init(world, rank, uid[0])
    eng.stage(reqs)
    now = spec.clock_now

    def barrier():
        if dist is not None:
            dist.barrier()

    def timed_region():
        """W untimed warm-up steps, then exactly K steps between barrier + sync on both sides; seconds for the K steps."""
        for _ in range(args.warmup):
            eng.enqueue(now)
        eng.sync()
        eng.reset_stats()
        barrier()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.enqueue(now)
        eng.sync()
        barrier()
        return time.perf_counter() - t0

    # The region is run twice and both are reported.  `cold_start`: as the process's very first GPU work - with the driver's
    # `--steps 20 --warmup 5` that is 25 steps (0.5 ms) on a device whose clocks have not ramped up yet (the same 20 steps
    # took 21.4 us each in BENCH_r03 against 16.0 us a few thousand steps later in the same process).
    # `value`: the same W + K steps after the device has been kept busy with this very step for >= 30 ms (`settle`) - the
    # state a scheduler that runs all day is in.  Nothing is skipped or cached in either: every step is digest + fit +
    # mapping with the verdict matrix written.
    cold = None
    settled = 0
    if not os.environ.get("NHD_BENCH_INNER") and not args.no_settle:
        cold = timed_region()
        settled = settle(eng, now, settle_ms=args.settle_ms, fixed_steps=2500 if world > 1 else 0)
        for _ in range(args.settle_regions):
            for _ in range(args.warmup + args.steps):
                eng.enqueue(now)
            eng.sync()
            settled += args.warmup + args.steps
    dt = timed_region()
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    st = eng.stats()
    # the timed region again, five times (same K steps, same barriers): spread of the headline figure
    rep = []
    for _ in range(5 if not os.environ.get("NHD_BENCH_INNER") else 0):
        barrier()
        eng.sync()
        r0 = time.perf_counter()
        for _ in range(args.steps):
            eng.enqueue(now)
        eng.sync()
        barrier()
        rep.append((time.perf_counter() - r0) * 1e3 / args.steps)
    steady = steady_state(eng, now, barrier, args.pods, n_total, fixed_steps=2500 if world > 1 else 0) if not os.environ.get("NHD_BENCH_INNER") else None
    score, _, maps = eng.fetch(want_bitmap=False, want_map=True)
    evals = float(args.pods) * n_total * args.steps
    ms_per_step = dt * 1e3 / args.steps
    fit_ms = st.fit_ms_total / max(1, st.launches)
    # the pipelined form keeps `pipes` step launches in flight (two pipelines on two streams): a launch's HIP-event duration
    # then overlaps its neighbour's, and the rate at which k_step moves bytes is pipes x (bytes per launch) / duration
    pipes = max(1, int(getattr(st, "pipes", 1) or 1))
    achieved = pipes * st.bytes_last / (fit_ms * 1e-3) / 1e9 if fit_ms > 0 else 0.0
    inner = bool(os.environ.get("NHD_BENCH_INNER"))

    # HBM traffic of the step kernel from hardware counters: short rocprofv3 passes of this very script (FETCH_SIZE,
    # WRITE_SIZE, SQ_INSTS_VALU; one counter set per pass, never combined with tracing), rank 0 at N = 1.  Falls back
    # to the committed profile of the same workload, and says which of the two it is.
    counters = None
    if rank == 0 and world == 1 and not args.no_pmc and not inner:
        counters = measure_counters(args)
    traffic, traffic_source = None, None
    if counters and counters.get("hbm_bytes_per_launch"):
        traffic = counters["hbm_bytes_per_launch"]
        traffic_source = "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of this script (2 x FETCH_SIZE + WRITE_SIZE, mean over the k_step launches)"
    else:
        for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")), reverse=True):
            with open(tpath) as f:
                tj = json.load(f)
            w = tj["workload"]
            if (w["config"], w["nodes_per_gpu"], w["pods"]) == (args.config, args.nodes_per_gpu, args.pods):
                traffic = tj["hbm_bytes_per_launch"]
                traffic_source = "from " + os.path.relpath(tpath, ROOT) + " (committed profile of the same command, not this run)"
                break
    valu = counters.get("valu_insts_per_launch") if counters else None

    out = {
        "metric": "pod-placement filter-and-score throughput (pod x node fit-and-score evaluations/s; placement decisions/s under commit semantics in mode_b)",
        "value": evals / dt, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "u64 bitmaps + int32 table look-ups (f64 NIC arithmetic in the request digest)", "data": "synthetic",
        "snapshot_decisions_per_s": args.pods * args.steps / dt,
        "repeats": None if not rep else {"n": len(rep), "ms_per_step_min": min(rep), "ms_per_step_median": sorted(rep)[len(rep) // 2],
                                         "ms_per_step_max": max(rep), "note": "the timed region repeated after the headline measurement (rank-local clock)"},
        "steady_state": steady,
        "protocol": {"order": "cold_start region (W warm-up + K timed steps as the first GPU work), settle, headline region (the same W + K steps)" if cold is not None
                              else "W warm-up + K timed steps as the first GPU work of the process",
                     "settle_steps": settled,
                     "note": "settle = the benchmarked step itself enqueued for >= 30 ms so that the device clocks are where a long-running scheduler "
                             "keeps them; both regions run every role of the step and write the verdict matrix"},
        "cold_start": None if cold is None else {"ms_per_step": cold * 1e3 / args.steps, "evals_per_s": evals / cold, "steps": args.steps, "warmup": args.warmup,
                                                  "note": "rank-local clock; the first %d steps this process ran on the GPU" % (args.steps + args.warmup)},
        "placed_pods": int(np.count_nonzero(score)),
        "config": {"workload": f"BASELINE config {args.config} cluster: {args.nodes_per_gpu} nodes/GPU x {args.pods} pods, "
                               f"CPU+GPU+NIC predicate, PCI locality for ~half the pods, node axis sharded over {world} GPU(s)",
                   "nodes_total": n_total, "nodes_per_gpu": args.nodes_per_gpu, "pods": args.pods,
                   "parallelism": f"node-shard x{world}, RCCL all-reduce(max) of {args.pods} u64 scores" if world > 1 else "single GPU",
                   "nic_signatures": st.nsig, "lds_bytes_per_block": st.lds_bytes},
        "roofline": roofline(st, fit_ms, achieved, traffic, traffic_source, valu, counters, pipes, ms_per_step),
    }

    if rank == 0 and world == 1 and not args.no_extras and not inner:
        out["end_to_end"] = end_to_end(eng, reqs, now, args.pods, n_total)
        out["single_find"] = single_find(eng, reqs, now, n_total)
        out["mode_b"] = mode_b(eng, pk, reqs, now, args.pods, parity=(spec, tops, pod_groups))
        out["other_configs"] = other_configs(args, local_rank)
        out["deltas"] = delta_rate(eng, table)
        out["big_pod_find"] = big_pod_find(eng, pk, pods, pod_groups, tops, reqs, now, n_total)
        out["sched_loop"] = sched_loop(args.config, local_rank)
        out["score_only"] = score_only(eng, reqs, now, args.pods, n_total)     # last: it changes the context's outputs
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not inner:
        out["cpu_baseline"] = cpu_baseline(spec, tops, pod_groups, args.cpu_sample_pods, score, lo, winner_index, args)
    if world > 1 and not strong and not args.no_extras:
        # BASELINE.json's own multi-GPU shapes ride along (strong scaling: the cluster is fixed, the shards shrink), so that one
        # run per N yields both curves: config 4's 65 536 nodes x 4 096 pods at every N, config 5's 262 144 x 16 384 at N = 8
        legs = [(4, 65536, 4096)] + ([(5, 262144, 16384)] if world == 8 else [])
        out["strong_scaling"] = [strong_leg(cfg, total, P, world, rank, local_rank, dist, min(args.steps, 400), min(args.warmup, 400))
                                 for cfg, total, P in legs]
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        eng.comm_destroy()
        dist.destroy_process_group()


def settle(eng, now, settle_ms=30.0, min_steps=1000, fixed_steps=0):
    """Keep the device busy with the benchmarked step for at least `settle_ms` (and `min_steps` steps); returns the steps run.
    Sharded runs pass `fixed_steps`: every step carries an all-reduce, so all ranks must enqueue the same number of them - a
    wall-clock loop would let the ranks disagree and the collective hang."""
    done = 0
    if fixed_steps:
        for _ in range(fixed_steps):
            eng.enqueue(now)
        done = fixed_steps
    else:
        t_end = time.perf_counter() + settle_ms * 1e-3
        while time.perf_counter() < t_end or done < min_steps:
            for _ in range(100):
                eng.enqueue(now)
            done += 100
    eng.sync()
    return done


def steady_state(eng, now, barrier, P, n_total, steps=1000, settle_ms=30.0, repeats=5, fixed_steps=0):
    """The same step at steady clocks, whatever --steps / --warmup the caller chose: the first ~1 000 steps after start-up run
    ~25 % slower (clock ramp), so a `--steps 20 --warmup 5` run times cold steps.  Here: enqueue for at
    least `settle_ms` of wall time, then time `steps` steps `repeats` times (same barriers as the headline region).  Never the
    headline `value` - that stays what --steps / --warmup define."""
    settled = settle(eng, now, settle_ms, 1000, fixed_steps)
    ts = []
    for _ in range(repeats):
        barrier()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.enqueue(now)
        eng.sync()
        barrier()
        ts.append((time.perf_counter() - t0) * 1e3 / steps)
    ts.sort()
    med = ts[len(ts) // 2]
    return {"steps": steps, "repeats": repeats, "settle_steps": settled, "ms_per_step_min": ts[0], "ms_per_step_median": med, "ms_per_step_max": ts[-1],
            "evals_per_s_median": float(P) * n_total / (med * 1e-3), "snapshot_decisions_per_s_median": P / (med * 1e-3),
            "note": "rank-local clock; >= %d settle steps (>= %.0f ms of enqueues) before the first timed repeat" % (settled, settle_ms)}


def strong_leg(cfg, total_nodes, P, world, rank, local_rank, dist, steps, warmup):
    """One of BASELINE.json's multi-GPU configurations: `total_nodes` sharded contiguously over the ranks, pods replicated,
    one all-reduce(max) of the packed scores per step.  Same protocol as the headline: warm-up, barrier, K steps, barrier,
    MAX of the ranks' times."""
    import torch
    from nhd_amd import pack
    from nhd_amd.engine import Engine
    from workload import planes, refmodel, synth
    per = (total_nodes + world - 1) // world
    lo, hi = rank * per, min(total_nodes, (rank + 1) * per)
    spec = synth.make_cluster(cfg, n_nodes=total_nodes).shard(lo, hi)
    pods, groups = synth.make_pods(cfg, n_pods=P)
    tops = [refmodel.make_topology(s) for s in pods]
    pk = pack.Packer()
    table = planes.planes_from_spec(pk, spec)
    reqs = pk.digest_many(tops, groups)
    pk.close_signatures()                 # (as in the headline run: every NIC state a commit can produce has its signature - the mode-B leg below)
    eng = Engine(local_rank)
    eng.set_dictionary(pk)
    eng.upload(table, global_base=lo)
    uid = [eng.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    eng.comm_init(world, rank, uid[0])
    eng.stage(reqs)
    for _ in range(warmup):
        eng.enqueue(spec.clock_now)
    eng.sync()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.enqueue(spec.clock_now)
    eng.sync()
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    score, _, _ = eng.fetch(want_bitmap=False, want_map=True)
    dist.barrier()
    eng.comm_destroy()
    out = {"config": cfg, "nodes_total": total_nodes, "nodes_per_gpu": per, "pods": P, "n_gpus": world, "scaling": "strong",
           "ms_per_step": dt * 1e3 / steps, "evals_per_s": float(P) * total_nodes * steps / dt,
           "snapshot_decisions_per_s": P * steps / dt, "placed_pods": int(np.count_nonzero(score))}
    if cfg == 4:
        # BASELINE's own unit at N > 1: placement decisions/s under the scheduler's commit semantics over the sharded cluster
        out["mode_b"] = sharded_mode_b(eng, pk, table, reqs, spec, lo, hi, cfg, total_nodes, tops, groups, dist, rank, world)
    eng.close()
    return out


def sharded_mode_b(eng, pk, table, reqs, spec, lo, hi, cfg, total_nodes, tops, groups, dist, rank, world):
    """Mode B with one process per GPU (nhd_amd.sharding.schedule_batch_sharded: the shards' sequential passes pipelined over pod
    slices in a lock-step ring, pods travelling rank to rank over the context's RCCL communicator - nhdfit_comm_sendrecv -, one
    all-reduce of the results): the commits
    stay in the shards' mirrors (apply), every timed call starts from freshly uploaded shards, time = MAX over the ranks.  Rank 0
    then decides the batch again with the independent oracle over the WHOLE cluster and compares node, mapping and physical ids of
    every pod.  Nothing here may take the run down: an error is reported in the leg's place."""
    import torch
    from nhd_amd import sharding
    from workload import synth
    P = len(reqs)
    now = spec.clock_now
    per = (total_nodes + world - 1) // world
    chunk = max(512, P // 4)              # pod slices of the pipeline down the ranks: few and long (a slice costs every rank two device passes)
    err = None
    ts, res = [], None
    try:
        bits = np.zeros(((hi - lo + 63) // 64) * 64, np.uint8)
        bits[:hi - lo] = (np.asarray(spec.n_gpus) == 0)
        nogpu = np.packbits(bits, bitorder="little").view(np.uint64).copy()
        transport = sharding.RcclTransport(eng)               # the pods travel rank to rank over the context's own communicator (ncclSend / ncclRecv behind the C-ABI)
        for _ in range(3):
            eng.upload(table, global_base=lo)                 # the shard as the snapshot has it (the previous call's commits are gone)
            dist.barrier()
            t0 = time.perf_counter()
            res = sharding.schedule_batch_sharded(eng, reqs, now, pk, nogpu, transport, apply=True, chunk=chunk)
            dist.barrier()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ts.append(float(t.item()))
    except (Exception, SystemExit) as e:                     # (incl. SystemExit: never leave the other ranks waiting)
        err = f"{type(e).__name__}: {e}"[:300]
    flag = torch.tensor([1 if err else 0], dtype=torch.int32)
    try:
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    except (Exception, SystemExit) as e:
        err = err or f"{type(e).__name__}: {e}"[:300]
        flag[0] = 1
    if int(flag.item()):
        return {"error": err or "another rank failed", "call": "nhd_amd.sharding.schedule_batch_sharded"}
    node, maps, places, status = res
    t = min(ts[1:]) if len(ts) > 1 else ts[0]
    out = {"call": "nhd_amd.sharding.schedule_batch_sharded (one process per GPU: nhdfit_schedule_batch per shard and pod slice, pods rank to rank over RCCL "
                   "(nhdfit_comm_sendrecv), one all-reduce of the results; commits left in the shards' mirrors)",
           "decisions_per_s": P / t, "ms_per_batch": t * 1e3, "placed": int((node >= 0).sum()), "n_gpus": world, "pods_per_slice": chunk,
           "commits_that_would_raise": int((status == 1).sum()),
           "placed_per_shard": [int(((node >= r * per) & (node < (r + 1) * per)).sum()) for r in range(world)],
           "note": "first-fit hands a pod to the first shard that still has room for it, so the commit chain stays serial whatever the number of "
                   "shards: sharding adds capacity and snapshot throughput (evals/s above), not mode-B rate - compare with mode_b of the N = 1 line"}
    if rank == 0:
        try:
            out["parity"] = mode_b_parity(synth.make_cluster(cfg, n_nodes=total_nodes), tops, groups, now, reqs, node, maps, places, status)
        except (Exception, SystemExit) as e:
            out["parity"] = {"identical": False, "error": f"{type(e).__name__}: {e}"[:300]}
    dist.barrier()
    return out


def roofline(st, fit_ms, achieved, traffic, traffic_source, valu, counters, pipes=1, ms_per_step=None):
    """`achieved` / `frac` (the contract's definition, and what VERDICT r05 recomputed): SURVEY.md 8(d)'s algorithmic bytes of ONE
    k_step launch / that launch's mean HIP-event duration over the timed region (`kernel_ms`; it agrees with the rocprofv3
    --kernel-trace --stats average kept under profiles/), against the 8 TB/s HBM spec.  The library keeps two launches in flight on
    two streams, so a launch lasts about two steps of wall clock: the device's byte RATE is the wall-clock figure `frac_wall`
    (bytes per step / wall clock per step), which rides along and is NOT `frac`.  Neither says what HBM did: the algorithmic
    bytes re-count the 1.5 MB of node records once per pod tile and L2 serves 63 of 64 of those reads - `hbm_counter_frac` (the
    counters' bytes per launch / wall clock per step) is HBM's own figure.  All unit fractions are flat scalars of this object
    (the driver's record keeps scalars only): hbm_counter_frac, lds_frac, valu_issue_frac, wait_frac, bank_conflict_share."""
    secs = fit_ms * 1e-3 / max(1, pipes)                  # kernel time per launch's worth of work at the measured concurrency
    step_s = ms_per_step * 1e-3 if ms_per_step else secs
    hbm = None if not traffic or step_s <= 0 else {
        "achieved": traffic / step_s / 1e9, "frac": traffic / step_s / 1e9 / HBM_PEAK_GBS,
        "note": "bytes the counters saw per launch / the wall clock per step (one launch per step): what HBM really moved"}
    issue = None if not valu or step_s <= 0 else {
        "valu_wave_insts_per_launch": valu, "valu_issue_frac": valu * VALU_NS_PER_WAVE_INST * 1e-9 / SIMDS / step_s,
        "note": "wave64 VALU instructions x 1.2 ns (measured full-rate issue per SIMD, profiles/r02/valu_calib.jsonl) / 1024 SIMDs / wall clock per step"}
    lds = None
    if counters and counters.get("SQ_LDS_IDX_ACTIVE") and step_s > 0:
        cyc = counters["SQ_LDS_IDX_ACTIVE"] / CUS                      # LDS-array cycles per CU and launch
        lds = {"lds_active_cycles_per_cu": cyc, "lds_frac": cyc / (step_s * CLOCK_HZ),
               "bank_conflict_share": (counters.get("SQ_LDS_BANK_CONFLICT", 0.0) / counters["SQ_LDS_IDX_ACTIVE"]),
               "note": "SQ_LDS_IDX_ACTIVE / 256 CUs / (wall clock per step x 2.4 GHz): share of a step the CU's LDS pipe is busy"}
    wait = None
    if counters and counters.get("SQ_WAVE_CYCLES") and counters.get("SQ_WAIT_ANY") is not None:
        wait = counters["SQ_WAIT_ANY"] / counters["SQ_WAVE_CYCLES"]   # share of the resident waves' cycles spent waiting on any counter (vmcnt / lgkmcnt / ...)
    fr = {"hbm": hbm["frac"] if hbm else None, "lds": lds["lds_frac"] if lds else None, "valu": issue["valu_issue_frac"] if issue else None}
    known = {k: v for k, v in fr.items() if v is not None}
    top = max(known, key=known.get) if known else None
    bound = "hbm" if top == "hbm" and known[top] >= 0.5 else ("latency" if not known or known[top] < 0.5 else top)
    wall = None if not ms_per_step else st.bytes_last / (ms_per_step * 1e-3) / 1e9
    per_launch = st.bytes_last / (fit_ms * 1e-3) / 1e9 if fit_ms > 0 else 0.0
    # `bound` names the roofline the kernel is priced against - the contract offers "hbm" | "mfma", and this is byte / integer work with no
    # matrix arithmetic; which unit actually limits the launch (none above half of its peak: latency) is `limiter` / `limited_by`
    return {"bound": "hbm", "limiter": bound, "priced_against": "hbm", "kernel": "k_step", "achieved": per_launch, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": per_launch / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": int(st.bytes_last), "kernel_ms": fit_ms, "concurrency": pipes,
            "frac_kernel": per_launch / HBM_PEAK_GBS,
            "frac_wall": None if wall is None else wall / HBM_PEAK_GBS,
            "hbm_counter_frac": hbm["frac"] if hbm else None,
            "lds_frac": lds["lds_frac"] if lds else None,
            "valu_issue_frac": issue["valu_issue_frac"] if issue else None,
            "wait_frac": wait,
            "bank_conflict_share": lds["bank_conflict_share"] if lds else None,
            "traffic_over_algorithmic": None if not traffic else traffic / float(st.bytes_last),
            "achieved_from_wall": wall,
            "achieved_from_events_x_concurrency": achieved,
            "achieved_note": "achieved / frac = algorithmic bytes of one launch / that launch's mean HIP-event duration (kernel_ms), as the contract "
                             "defines it; `concurrency` launches overlap, so the device's byte rate is achieved_from_wall (frac_wall) - bytes per "
                             "step / wall clock per step of the timed region",
            "bytes_formula": "SURVEY.md 8(d): ceil(P/64) * N * 24 (16-byte node record + 8-byte busy time per node and tile) "
                             "+ P * 128 (requests) + P * N / 8 (verdict matrix) + 8 * P (scores)",
            "frac_note": "algorithmic bytes re-count the node records once per pod tile (SURVEY.md 8(d)'s definition) although L2 / Infinity "
                         "Cache serve those re-reads: frac_wall can pass 1 and says nothing about HBM - hbm_counter_frac is what HBM moved",
            "hbm_counter": hbm, "lds": lds, "issue": issue, "unit_fracs": fr,
            "limited_by": "no unit above half of its peak -> latency: dependent L2 / LDS round trips inside short blocks plus the fixed cost "
                          "of a launch" if bound == "latency" else bound,
            "digest_kernel_ms": st.digest_ms_last, "device_step_ms": st.step_ms_last}


def end_to_end(eng, reqs, now, P, n_total):
    """The whole call as a scheduler makes it: host request records in, winners + mappings out (nhdfit_find: host sort
    into tiles, H2D of the requests, then ONE launch - digest, fit and mapping tile by tile, results into a fine-grained
    host block - or, where that form does not apply, the staged launches and two copies back) - host buffers on both
    sides, so PCIe and the launch latencies are inside.  Never the headline `value`."""
    eng.find(reqs, now, want_bitmap=False, want_map=True)
    before = getattr(eng.stats(), "batch_finds", 0)
    ts = []
    for _ in range(9):
        t0 = time.perf_counter()
        eng.find(reqs, now, want_bitmap=False, want_map=True)
        ts.append(time.perf_counter() - t0)
    one = getattr(eng.stats(), "batch_finds", 0) - before == len(ts)
    t = min(ts)
    ts.sort()
    return {"call": "nhdfit_find (stage + H2D + ONE launch: digest, fit, mapping per tile; results in fine-grained host memory)" if one else
                    "nhdfit_find (stage + H2D + 3 launches: digest, fused step, one-launch drain + D2H of scores and mappings)",
            "ms_per_call": t * 1e3, "ms_per_call_median": ts[len(ts) // 2] * 1e3, "single_launch": bool(one),
            "evals_per_s": P * n_total / t, "decisions_per_s": P / t}


def single_find(eng, reqs, now, n_total, calls=200):
    """One pending pod against the whole mirror - the call the scheduler's pod-at-a-time loop makes (nhd/NHDScheduler.py:277):
    nhdfit_find with P = 1, one kernel launch (digest -> fit -> mapping inside it), request and result in fine-grained host
    memory.  Median and minimum over `calls` calls with different pods, through ctypes."""
    sel = np.flatnonzero(reqs["n_groups"] <= 3)[:calls]
    if not len(sel):
        return None
    before = eng.stats().small_finds
    for k in sel[:8]:
        eng.find(reqs[k:k + 1], now, want_bitmap=False, want_map=True)
    ts = []
    for k in sel:
        one = reqs[k:k + 1]
        t0 = time.perf_counter()
        eng.find(one, now, want_bitmap=False, want_map=True)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    took = eng.stats().small_finds - before
    return {"call": "nhdfit_find, 1 pod, winner + mapping (single launch)" if took else "nhdfit_find, 1 pod (staged path)",
            "ms_per_call_median": ts[len(ts) // 2] * 1e3, "ms_per_call_min": ts[0] * 1e3, "calls": len(ts),
            "single_launch_calls": int(took), "nodes": int(n_total)}


def big_pod_find(eng, pk, pods, pod_groups, tops, reqs, now, n_total, calls=12):
    """Pods beyond the table pass (5..8 processing groups; nhdfit_big_req): one such pod against the whole
    mirror through nhdfit_big_find - the general path, explicit enumeration per (pod, node), lane = node.  A rare path with no
    rate to defend: reported so that its cost is known (milliseconds per pod against the table pass's microseconds).  Parity is
    asserted in the run the only way it can be without a second implementation on the box: ordinary pods digested as big
    requests must get the score word and mapping the table pass gives them.  Never raises: an error is reported in place."""
    try:
        from workload import refmodel
        ordinary = [k for k in range(len(tops)) if len(tops[k].proc_groups) <= 3][:calls]
        big_ord = np.zeros(len(ordinary), pack_mod().BIG_REQ)
        for j, k in enumerate(ordinary):
            big_ord[j] = pk.digest_big(tops[k], pod_groups[k])
        score, _, maps = eng.find(reqs[ordinary], now, want_bitmap=False, want_map=True)
        bscore, bmaps = eng.big_find(big_ord, now)
        same = bool(np.array_equal(score, bscore))
        for j in np.flatnonzero(score != 0):
            G = int(reqs[ordinary[j]]["n_groups"])
            same = same and all(list(maps[j][f][:n]) == list(bmaps[j][f][:n]) for f, n in (("gpu", G), ("cpu", G + 1), ("nic_numa", G), ("nic_idx", G)))
        # real big pods: the groups of two or three of the batch's pods under one topology (5..8 groups)
        merged, k = [], 0
        while len(merged) < calls and k + 3 <= len(pods):
            groups = [g for s in pods[k:k + 3] for g in s["groups"]]
            k += 3
            if 5 <= len(groups) <= 8:
                merged.append((dict(pods[k - 3], groups=groups), pod_groups[k - 3]))
        ts, placed = [], 0
        for sp, grp in merged:
            r = pk.digest_big(refmodel.make_topology(sp), grp).reshape(1)
            eng.big_find(r, now)
            t0 = time.perf_counter()
            sc, _ = eng.big_find(r, now)
            ts.append(time.perf_counter() - t0)
            placed += int(sc[0] != 0)
        ts.sort()
        return {"call": "nhdfit_big_find, 1 pod with 5..8 processing groups, winner + mapping (general path, every node enumerated)",
                "ms_per_call_median": ts[len(ts) // 2] * 1e3 if ts else None, "ms_per_call_min": ts[0] * 1e3 if ts else None, "calls": len(ts),
                "placed": placed, "nodes": int(n_total),
                "parity": {"identical": same, "pods": len(ordinary),
                           "against": "the table-driven pass on the same (ordinary) pods digested both ways: score words and mappings"}}
    except Exception as e:  # noqa: BLE001 - an extra must not cost the run its line
        return {"error": f"{type(e).__name__}: {e}"}


def sched_loop(cfg, device, n=16384, P=1024):
    """Row f4 as a number of the run: what a scheduler gets THROUGH the drop-in class (nhd_amd.matcher.HipMatcher, attached to
    its node dict - Python above the C-ABI included), in placement decisions per second with every winner committed before the
    next pod is matched (nhd/NHDScheduler.py:249-353, 425-437): (1) pod by pod - FindNodes(nodes, [top], pod_groups=[groups])
    (the kernel applies InitialNodeFilter) + CommitPlacement (nhdfit_commit on the mirror, physical ids expanded); (2) pod by
    pod the reference's way - the scheduler builds the filtered dict first (InitialNodeFilter's O(N) Python loop per pod), then
    FindNode(filtered, top) + CommitPlacement; (3) the pending list as ONE ScheduleBatch (decided and committed on the device,
    physical ids of every pod expanded).  Node objects: workload.refmodel stand-ins built from the same NFD labels; the
    reference's own bookkeeping on its Node objects (SetPhysicalIdsFromMapping in Python) is the scheduler's cost and not timed
    here - tools/time_sched_loop.py times the loop with a stand-in for it (profiles/r06).  All three must make the same
    decisions and produce the same ids: asserted.  Never raises: an error is reported in place."""
    try:
        from nhd_amd.matcher import HipMatcher
        from workload import refmodel, synth
        spec = synth.make_cluster(cfg, n_nodes=n)
        pods, groups = synth.make_pods(cfg, n_pods=P)
        tops = [refmodel.make_topology(s) for s in pods]
        now = spec.clock_now
        res = {}
        for leg in ("batched", "pod_by_pod_kernel_filter", "pod_by_pod_filtered_dict"):
            nodes = spec.build_nodes()
            m = HipMatcher(device=device, clock=lambda: now)
            m.attach(nodes)
            if leg == "pod_by_pod_filtered_dict":
                want = [set(g) for g in groups]
            t0 = time.perf_counter()
            if leg == "batched":
                got = m.ScheduleBatch(nodes, tops, pod_groups=groups, now=now, apply=True)
                ids = list(m.last_placements)
            else:
                got, ids = [], []
                for k, top in enumerate(tops):
                    if leg == "pod_by_pod_kernel_filter":
                        r = m.FindNodes(nodes, [top], pod_groups=[groups[k]])[0]
                    else:       # nhd/NHDScheduler.py:235-247: keep node iff its groups meet the pod's and it is active
                        filt = {nm: nd for nm, nd in nodes.items() if nd.active and not want[k].isdisjoint(nd.groups)}
                        r = m.FindNode(filt, top)
                    got.append(r)
                    ids.append(m.CommitPlacement(r[0], top, r[1], busy_time=now) if r[0] is not None else None)
            dt = time.perf_counter() - t0
            res[leg] = (dt, got, ids)
            m.engine.close()
        same = all(res[leg][1] == res["batched"][1] and res[leg][2] == res["batched"][2] for leg in res)
        out = {"nodes": n, "pending_pods": P, "config": cfg, "placed": sum(r[0] is not None for r in res["batched"][1]), "identical_decisions_and_ids": bool(same),
               "call": "HipMatcher attached to the scheduler's node dict; every winner committed on the device mirror before the next pod is matched"}
        for leg, (dt, _, _) in res.items():
            out[leg] = {"pods_per_s": P / dt, "us_per_pod": dt / P * 1e6}
        return out
    except Exception as e:  # noqa: BLE001 - an extra must not cost the run its line
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def pack_mod():
    from nhd_amd import pack
    return pack


def score_only(eng, reqs, now, P, n_total, steps=100):
    """The same step without the verdict matrix: what Matcher.FindNode (mode A) needs is the winners and their mappings;
    the P x N feasibility bits are an extra output (bitmap_out, mode B's scan rows).  SURVEY.md 8(d): "drop any output
    term the build does not materialise and say so" - the headline `value` keeps materialising it."""
    eng.set_outputs(bitmap=False, mapping=True)
    eng.stage(reqs)
    for _ in range(10):
        eng.enqueue(now)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.enqueue(now)
    eng.sync()
    dt = time.perf_counter() - t0
    score, _, _ = eng.fetch(want_bitmap=False, want_map=True)
    return {"ms_per_step": dt * 1e3 / steps, "evals_per_s": float(P) * n_total * steps / dt, "placed_pods": int(np.count_nonzero(score)),
            "note": "verdict matrix (P x N / 8 bytes per step) not written; winners and mappings only"}


def delta_rate(eng, table, n=4096):
    """Row f2 (nhdfit_apply_deltas): n pods' worth of resources taken from n different nodes in one call, then given back
    in a second one (RemoveResourcesFromTopology / AddResourcesFromTopology as delta records: 4 cores, a NIC claim, 2 GB
    of hugepages each); the mirror must come back bit for bit.  Time per call includes the H2D of the records, the
    kernel, the D2H of the statuses and the host-side grouping by node."""
    from nhd_amd import pack
    n = min(n, table.n)
    before = eng.download()
    d = np.zeros(n, pack.DELTA)
    d["node"] = np.arange(n, dtype=np.uint32) * (table.n // n)
    free = before.p0["t0"][d["node"]] & before.p1["t1"][d["node"]]
    pick = np.zeros_like(free)
    for _ in range(4):                                     # the four lowest free cores of socket 0
        low = free[:, 0] & (~free[:, 0] + np.uint64(1))
        pick[:, 0] |= low
        free[:, 0] &= ~low
    d["t0"] = pick
    d["t1"] = pick
    d["hugepages_gb"] = 2
    d["nic_n"] = 1                                         # NIC (0, 0)
    ts = {}
    for name, op in (("take", pack.DELTA_TAKE), ("give", pack.DELTA_GIVE)):
        d["op"] = op
        t0 = time.perf_counter()
        st = eng.apply_deltas(d)
        ts[name] = time.perf_counter() - t0
        if (st != pack.DELTA_OK).any():
            raise SystemExit("delta leg: a delta came back with a status")
    after = eng.download()
    for f in ("p0", "p1", "p2", "p3", "p4", "detail"):
        if not np.array_equal(getattr(before, f), getattr(after, f)):
            raise SystemExit("delta leg: take + give did not restore plane " + f)
    t = max(ts.values())
    return {"call": "nhdfit_apply_deltas (one call per direction, %d nodes each)" % n, "ms_per_call": t * 1e3, "deltas_per_s": n / t,
            "mirror_restored": True}


def other_configs(args, device):
    """BASELINE.json's other shapes on one GPU, same step, same clock (not the headline; 300 steps each after 20 warm-up steps on a
    fresh engine - a region's end costs one launch's tail plus the drain, ~60-100 us, whatever its length): config 2 whole
    (4 096 nodes x 256 pods), config 3 whole (16 384 x 1 024), a config-5 shard cut both ways (32 768 x 2 048: an eighth of
    its nodes, an eighth of its pods) and as one GPU of the 8-GPU run has it (32 768 nodes x all 16 384 pods).  Mode A step rate
    and the mode-B decision rate of each."""
    from nhd_amd import pack
    from nhd_amd.engine import Engine
    from workload import planes, refmodel, synth
    rows = []
    for cfg, n, P in ((2, 4096, 256), (3, 16384, 1024), (5, 32768, 2048), (5, 32768, 16384)):
        if (cfg, n, P) == (args.config, args.nodes_per_gpu, args.pods):
            continue
        spec = synth.make_cluster(cfg, n_nodes=n)
        pods, groups = synth.make_pods(cfg, n_pods=P)
        tops = [refmodel.make_topology(s) for s in pods]
        pk = pack.Packer()
        table = planes.planes_from_spec(pk, spec)
        reqs = pk.digest_many(tops, groups)
        pk.close_signatures()
        eng = Engine(device)
        eng.set_dictionary(pk)
        eng.upload(table)
        eng.stage(reqs)
        for _ in range(20):
            eng.enqueue(spec.clock_now)
        eng.sync()
        steps = 300
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.enqueue(spec.clock_now)
        eng.sync()
        dt = time.perf_counter() - t0
        score, _, maps = eng.fetch(want_bitmap=False, want_map=True)
        # the whole call, host records in / winners and mappings out (nhdfit_find: one launch for a batch) - what a small shape costs a
        # scheduler that asks once: its results must be the pipelined steps' (asserted: a mismatch ends the run)
        e2e = end_to_end(eng, reqs, spec.clock_now, P, n)
        s1, _, m1 = eng.find(reqs, spec.clock_now, want_bitmap=False, want_map=True)
        if not (np.array_equal(s1, score) and np.array_equal(m1, maps)):
            raise SystemExit(f"PARITY FAILURE (nhdfit_find, config {cfg}): the single call's winners / mappings differ from the pipelined steps'")
        mb = mode_b(eng, pk, reqs, spec.clock_now, P, parity=(spec, tops, groups))     # (parity asserted: a mismatch ends the run)
        rows.append({"config": cfg, "nodes": n, "pods": P, "ms_per_step": dt * 1e3 / steps, "evals_per_s": float(P) * n * steps / dt,
                     "steps": steps, "placed_pods": int(np.count_nonzero(score)), "nic_signatures": len(pk.sigs),
                     "find_ms_per_call": e2e["ms_per_call"], "find_single_launch": e2e["single_launch"],
                     "mode_b_decisions_per_s": mb["decisions_per_s"], "mode_b_placed": mb["placed"], "mode_b_parity": mb["parity"]})
        eng.close()
    return rows


def mode_b(eng, pk, reqs, now, P, parity=None):
    """Placement decisions under the scheduler's commit semantics (nhdfit_schedule_batch): every winner is committed to
    the packed node state on the device (physical core / GPU ids out) before the next pod is matched.
    `parity` = (spec, tops, pod_groups): the batch is decided again by the independent oracle (oracle/seq_oracle.py: C scan
    and C commit over per-core / per-GPU / per-NIC records, the winner's mapping from the pure-Python restatement) and
    node, mapping and physical ids of every pod are asserted identical - after the timed calls, never inside them."""
    eng.schedule_batch(reqs, now, pk, apply=False)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        node, maps, places, status = eng.schedule_batch(reqs, now, pk, apply=False)
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    out = {"call": "nhdfit_schedule_batch (snapshot pass + sequential commit on the device, mirror restored)", "decisions_per_s": P / t,
           "ms_per_batch": t * 1e3, "placed": int((node >= 0).sum()), "distinct_nodes": int(len(set(node[node >= 0].tolist()))),
           "commits_that_would_raise": int((status == 1).sum())}
    if parity is not None:
        out["parity"] = mode_b_parity(parity[0], parity[1], parity[2], now, reqs, node, maps, places, status)
    return out


def mode_b_parity(spec, tops, pod_groups, now, reqs, node, maps, places, status):
    from nhd_amd import pack
    from oracle import coracle, seq_oracle
    t0 = time.perf_counter()
    sc = seq_oracle.SeqCluster(coracle.Cluster.from_spec(spec))
    win, omaps, oids, n_def = seq_oracle.schedule_sequence(sc, tops, pod_groups, now)
    secs = time.perf_counter() - t0
    for i in range(n_def):
        w = win[i]
        if int(node[i]) != w:
            raise SystemExit(f"PARITY FAILURE (mode B): pod {i} placed on node {int(node[i])}, the oracle's loop says {w}")
        if w < 0:
            continue
        if int(status[i]) != 0:
            raise SystemExit(f"PARITY FAILURE (mode B): pod {i} came back with commit status {int(status[i])}")
        G = int(reqs[i]["n_groups"])
        m, om = maps[i], omaps[i]
        got = ([int(x) for x in m["gpu"][:G]], [int(x) for x in m["cpu"][:G + 1]], [(int(a), int(b)) for a, b in zip(m["nic_numa"][:G], m["nic_idx"][:G])])
        if got != (list(om["gpu"]), list(om["cpu"]), [tuple(x) for x in om["nic"]]):
            raise SystemExit(f"PARITY FAILURE (mode B): pod {i} mapping {got} != {om}")
        phys = int(spec.phys[w])
        ids = pack.expand_placement(places[i], G, phys // 2, phys, [int(reqs[i]["gpus"][g]) for g in range(G)])
        if ids != oids[i]:
            raise SystemExit(f"PARITY FAILURE (mode B): pod {i} physical ids {ids} != {oids[i]}")
    return {"identical": True, "pods_checked": n_def, "pods": len(tops), "defined_prefix": n_def,
            "checked": "node, NUMA mapping, NIC choice and physical core / GPU ids of every pod",
            "oracle": "oracle/seq_oracle.py (C scan + C commit over flat records, Python set-order mapping of the winner), "
                      f"{secs:.1f} s on {coracle.usable_cpus()} host cores",
            "note": "pods past defined_prefix follow a commit the reference itself raises on (none in this batch)" if n_def < len(tops) else
                    "the reference raises on no commit of this batch: the whole batch is defined"}


def measure_counters(args):
    """HBM bytes and VALU instructions per k_step launch: this script under rocprofv3, one counter set per pass."""
    rocprof = shutil.which("rocprofv3")
    if not rocprof:
        return None
    base = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "60", "--warmup", "5", "--config", str(args.config),
            "--nodes-per-gpu", str(args.nodes_per_gpu), "--pods", str(args.pods), "--no-cpu-baseline", "--no-pmc", "--no-extras"]
    env = dict(os.environ, NHD_BENCH_INNER="1", TMPDIR="/tmp")
    got = {}
    tmp = tempfile.mkdtemp(prefix="nhdbench_", dir="/tmp")
    try:
        for name, ctrs in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("valu", ["SQ_INSTS_VALU"]),
                           ("lds", ["SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES"])):
            d = os.path.join(tmp, name)
            cmd = [rocprof, "--pmc"] + ctrs + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"] + base
            try:
                subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=150, check=True)
            except Exception:  # noqa: BLE001 - counters are optional: no profiler, no counters
                return got or None
            vals = {}
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        if "k_step" in row["Kernel_Name"]:        # steady state: k_step_p (argument block by pointer)
                            key = (row["Counter_Name"], "k_step_p" in row["Kernel_Name"])
                            vals.setdefault(key, []).append(float(row["Counter_Value"]))
            names = {k[0] for k in vals}
            vals = {nm: (vals[(nm, True)] if (nm, True) in vals else vals[(nm, False)]) for nm in names}
            for k, v in vals.items():
                got[k] = sum(v) / len(v)
        if "FETCH_SIZE" in got and "WRITE_SIZE" in got:
            # gfx950: FETCH_SIZE counts the 128-byte requests of wide coalesced reads as 64 bytes -> x 2
            # (guides/MI355X_MICROARCH.md, HBM); both counters are in KiB
            got["hbm_bytes_per_launch"] = int((2 * got["FETCH_SIZE"] + got["WRITE_SIZE"]) * 1024)
        if "SQ_INSTS_VALU" in got:
            got["valu_insts_per_launch"] = got["SQ_INSTS_VALU"]
        return got
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(spec, tops, pod_groups, sample, gpu_score, base, winner_index, args):
    """The C port of the reference path (oracle/nhd_oracle.c) on the same inputs, 1 host core, on the
    first `sample` pods x all of this GPU's nodes; also asserts the GPU picked the same nodes.  `kind` stays "port": the
    reference is Python, and a Python reference may not travel to the GPU box in any form (source, bytecode or otherwise -
    task statement, section 3), so there is no oracle/_ref to time here.  What rides along instead: the unmodified reference's
    figures on this very workload from the build container (tools/cpu_reference.py, newest profiles/r*/cpu_reference.json),
    the pinned Python restatement timed on THIS box in THIS run, and the estimate the two give for the reference on this box."""
    from oracle import coracle
    cl = coracle.Cluster.from_spec(spec)
    sample = min(sample, len(tops))
    op = cl.pods_from_tops(tops[:sample], pod_groups[:sample])
    t0 = time.perf_counter()
    winner, _ = cl.find(op, spec.clock_now, want_feas=False, threads=1)
    dt = time.perf_counter() - t0
    gpu_winner = np.array([winner_index(s) - base if s else -1 for s in gpu_score[:sample]], dtype=np.int64)
    if not np.array_equal(gpu_winner, winner):
        raise SystemExit("PARITY FAILURE: GPU winners differ from the CPU port on the sampled pods")
    # the same sample on every host core (OpenMP over pods), for scale: the primary figure stays the 1-core one
    ncores = coracle.usable_cpus()                        # the cores this process may really use (CPU set, cgroup quota): the count the record states
    many = None
    if ncores > 1:
        t0 = time.perf_counter()
        winner_mt, _ = cl.find(op, spec.clock_now, want_feas=False, threads=ncores)
        dt_mt = time.perf_counter() - t0
        if not np.array_equal(winner_mt, winner):
            raise SystemExit("PARITY FAILURE: the multi-threaded CPU port disagrees with the single-threaded one")
        many = {"value": sample * spec.n / dt_mt, "cores": ncores, "seconds": dt_mt}
    python_port = python_oracle_sample(spec, tops, pod_groups, gpu_score, base, winner_index)
    reference = None
    for rpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "cpu_reference.json")), reverse=True):
        with open(rpath) as f:
            rj = json.load(f)
        if (rj.get("config"), rj.get("nodes"), rj.get("pods_in_batch")) == (args.config, spec.n, args.pods):
            reference = {"source": os.path.relpath(rpath, ROOT) + " (unmodified reference Matcher.FindNode on this workload's inputs, measured in the "
                                   "build container by tools/cpu_reference.py - the reference is Python and may not travel to the GPU box in any form)",
                         "one_core": {"value": rj["one_core"]["evals_per_s"], "decisions_per_s": rj["one_core"]["decisions_per_s"], "cores": 1},
                         "all_cores": {"value": rj["all_cores"]["evals_per_s"], "decisions_per_s": rj["all_cores"]["decisions_per_s"],
                                       "cores": rj["all_cores"]["cores"]},
                         "host": rj.get("host"), "sampled_pods": rj.get("sampled_pods"), "parity": rj.get("parity")}
            cal = (rj.get("python_restatement") or {}).get("calibration")
            if cal and python_port and python_port.get("value"):
                # the same calibration sample of the pinned Python restatement, timed there and here: how much faster one core of this
                # box runs the same CPython work - the only bridge between the two hosts that does not move the reference
                k = python_port["value"] / cal["evals_per_s"]
                reference["estimate_on_this_box"] = {
                    "one_core": rj["one_core"]["evals_per_s"] * k, "core_speed_ratio": k, "cores": 1,
                    "note": "ESTIMATE, not a measurement: the build container's reference figure x (Python restatement on this box / the same "
                            "sample in the build container); the reference itself is timed in the build container only"}
            break
    return {"value": sample * spec.n / dt, "unit": "evals/s", "cores": 1, "kind": "port",
            "all_host_cores": many, "python_restatement": python_port,
            "sample": f"first {sample} pods x {spec.n} nodes, oracle/nhd_oracle.c (gcc -O2), {dt:.1f} s; "
                      f"winners identical to the GPU's on all {sample} pods" +
                      ("" if not reference else f"; unmodified reference Matcher.FindNode on the same inputs, build container "
                       f"({reference['source'].split(' ')[0]}): {reference['one_core']['value']:.0f} evals/s on 1 core, "
                       f"{reference['all_cores']['value']:.0f} on {reference['all_cores']['cores']}" +
                       ("" if "estimate_on_this_box" not in reference else
                        f"; estimated on one core of this box: {reference['estimate_on_this_box']['one_core']:.0f} evals/s")),
            "reference": reference}


def python_oracle_sample(spec, tops, pod_groups, gpu_score, base, winner_index, n_nodes=4096, n_pods=2):
    """The pinned pure-Python restatement of Matcher.FindNode (oracle/nhd_oracle.py - the same per-node enumeration, real
    CPython sets; pinned to the unmodified reference by tests/test_oracle_vs_reference.py) timed on THIS box, one core:
    the first `n_pods` pods against the first `n_nodes` nodes (node objects built from the same labels).  The unmodified
    reference itself cannot travel to the GPU box; its figures from the build container ride along as `reference`."""
    from oracle import nhd_oracle as O
    n_nodes = min(n_nodes, spec.n)
    sub = spec.shard(0, n_nodes)
    nl = sub.build_nodes()
    names = list(nl)
    t0 = time.perf_counter()
    res = [O.find_node(O.initial_node_filter(nl, pod_groups[k]), tops[k], spec.clock_now) for k in range(n_pods)]
    dt = time.perf_counter() - t0
    agree = 0
    for k, r in enumerate(res):                      # where the GPU's winner lies inside the sampled nodes the two must agree
        gw = winner_index(gpu_score[k]) - base if gpu_score[k] else -1
        if 0 <= gw < n_nodes:
            if r[0] is None or names.index(r[0]) != gw:
                raise SystemExit("PARITY FAILURE: the Python restatement picks another node than the GPU on a sampled pod")
            agree += 1
    return {"value": n_pods * n_nodes / dt, "unit": "evals/s", "cores": 1, "kind": "port (pure Python, pinned to the reference)",
            "sample": f"first {n_pods} pods x first {n_nodes} nodes, {dt:.1f} s, same box", "winners_cross_checked": agree}


if __name__ == "__main__":
    main()
