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
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X spec (guides/MI355X_MICROARCH.md, "Chip-level parameters"); ~6300 achievable


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=4, help="BASELINE config whose cluster/pod mix is generated")
    ap.add_argument("--nodes-per-gpu", type=int, default=65536)
    ap.add_argument("--pods", type=int, default=4096)
    ap.add_argument("--cpu-sample-pods", type=int, default=1024, help="pods timed on the CPU port (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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

    dist = None
    if world > 1:
        import torch.distributed as dist          # control plane only (rendezvous, barrier, max-reduce of the time)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    from nhd_amd import pack, refmodel, synth
    from nhd_amd.engine import Engine, winner_index

    n_total = args.nodes_per_gpu * world
    lo, hi = rank * args.nodes_per_gpu, (rank + 1) * args.nodes_per_gpu
    spec_all = synth.make_cluster(args.config, n_nodes=n_total)
    spec = spec_all.shard(lo, hi)
    pods, pod_groups = synth.make_pods(args.config, n_pods=args.pods)
    tops = [refmodel.make_topology(s) for s in pods]

    pk = pack.Packer()
    table = pk.planes_from_spec(spec)
    reqs = pk.digest_many(tops, pod_groups)
    eng = Engine(local_rank)
    eng.set_dictionary(pk)
    eng.upload(table, global_base=lo)
    if world > 1:
        import torch
        uid = [eng.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        eng.comm_init(world, rank, uid[0])
    eng.stage(reqs)
    now = spec.clock_now

    def barrier():
        if dist is not None:
            dist.barrier()

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
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    st = eng.stats()
    score, _, maps = eng.fetch(want_bitmap=False, want_map=True)
    evals = float(args.pods) * n_total * args.steps
    ms_per_step = dt * 1e3 / args.steps
    fit_ms = st.fit_ms_total / max(1, st.launches)
    achieved = st.bytes_last / (fit_ms * 1e-3) / 1e9 if fit_ms > 0 else 0.0

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        w = tj["workload"]
        if (w["config"], w["nodes_per_gpu"], w["pods"]) == (args.config, args.nodes_per_gpu, args.pods):
            traffic = tj["hbm_bytes_per_launch"]      # PMC pass of the same command, committed under profiles/

    out = {
        "metric": "pod-placement filter-and-score throughput (pod x node fit-and-score evaluations/s; decisions/s in decisions_per_s)",
        "value": evals / dt, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64 bitmaps + int32 table look-ups (f64 NIC arithmetic in the request digest)", "data": "synthetic",
        "decisions_per_s": args.pods * args.steps / dt,
        "placed_pods": int(np.count_nonzero(score)),
        "config": {"workload": f"BASELINE config {args.config} cluster: {args.nodes_per_gpu} nodes/GPU x {args.pods} pods, "
                               f"CPU+GPU+NIC predicate, PCI locality for ~half the pods, node axis sharded over {world} GPU(s)",
                   "nodes_total": n_total, "nodes_per_gpu": args.nodes_per_gpu, "pods": args.pods,
                   "parallelism": f"node-shard x{world}, RCCL all-reduce(max) of {args.pods} u64 scores" if world > 1 else "single GPU",
                   "nic_signatures": st.nsig, "lds_bytes_per_block": st.lds_bytes},
        "roofline": {"bound": "hbm", "kernel": "k_step", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": int(st.bytes_last), "kernel_ms": fit_ms,
                     "digest_kernel_ms": st.digest_ms_last, "device_step_ms": st.step_ms_last,
                     "note": "achieved = algorithmic bytes of the fit role / mean HIP-event duration of the whole fused step "
                             "kernel (fit role + next step's digest + earlier steps' mapping roles in the same launch, "
                             "sampled every 8th step on the launch stream); traffic = HBM bytes/launch from rocprofv3 PMC "
                             "(2*FETCH_SIZE + WRITE_SIZE): the node planes and table images are served from L2 / Infinity "
                             "Cache, the kernel is VALU-issue bound (DESIGN.md section 4)"},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(spec, tops, pod_groups, args.cpu_sample_pods, score, lo, winner_index)
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        eng.comm_destroy()
        dist.destroy_process_group()


def cpu_baseline(spec, tops, pod_groups, sample, gpu_score, base, winner_index):
    """The C port of the reference path (oracle/nhd_oracle.c) on the same inputs, 1 host core, on the
    first `sample` pods x all of this GPU's nodes; also asserts the GPU picked the same nodes."""
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
    ncores = os.cpu_count() or 1
    many = None
    if ncores > 1:
        t0 = time.perf_counter()
        winner_mt, _ = cl.find(op, spec.clock_now, want_feas=False, threads=ncores)
        dt_mt = time.perf_counter() - t0
        if not np.array_equal(winner_mt, winner):
            raise SystemExit("PARITY FAILURE: the multi-threaded CPU port disagrees with the single-threaded one")
        many = {"value": sample * spec.n / dt_mt, "cores": ncores, "seconds": dt_mt}
    return {"value": sample * spec.n / dt, "unit": "evals/s", "cores": 1, "kind": "port",
            "all_host_cores": many,
            "sample": f"first {sample} pods x {spec.n} nodes, oracle/nhd_oracle.c (gcc -O2), {dt:.1f} s; "
                      f"winners identical to the GPU's on all {sample} pods",
            "reference_python_note": "the reference itself is Python and absent on the GPU box; measured in the build "
                                     "container it runs ~3-9 k evals/s/core (BASELINE.md section 2)"}


if __name__ == "__main__":
    main()
