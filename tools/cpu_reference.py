#!/usr/bin/env python3
"""CPU baseline as BASELINE.md section 3 words it: the UNMODIFIED reference Matcher.FindNode (imported from
/root/reference; build container only) on exactly the inputs bench.py generates, 1 core and all host cores (8 worker
processes, node axis cut into contiguous shards, unmodified FindNode per shard, shard winners merged in node order).
Parity with the product's own CPU checker (oracle C port) is asserted on every pod the reference ran.

The pinned Python restatement (oracle/nhd_oracle.py) is timed on the same pods and nodes, and on bench.py's own calibration
sample (first 2 pods x first 4 096 nodes), in the same process: bench.py times that calibration sample again on the GPU box, and
the ratio of the two is the only honest bridge between this container's cores and the box's (the reference is Python: it may not
travel to the box in any form - task statement, section 3).

    python tools/cpu_reference.py [--config 4] [--nodes 65536] [--pods 2] [--procs 8] > profiles/r06/cpu_reference.json
"""
import argparse, contextlib, io, json, logging, multiprocessing as mp, os, platform, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(cfg, n, lo=0, hi=None):
    from workload import synth
    from oracle import ref_loader
    ref = ref_loader.load()
    spec = synth.make_cluster(cfg, n_nodes=n)
    clock = ref_loader.VirtualClock(spec.clock_now).install()
    sub = spec if hi is None else spec.shard(lo, hi)
    return ref, spec, sub.build_nodes(ref), clock


def shard_worker(args):
    cfg, n, lo, hi, pod_idx, n_pods = args
    from workload import refmodel, synth
    from oracle import nhd_oracle, ref_loader
    logging.disable(logging.CRITICAL)
    ref, spec, nodes, clock = build(cfg, n, lo, hi)
    pods, groups = synth.make_pods(cfg, n_pods=n_pods)
    out = []
    t0 = time.perf_counter()
    for i in pod_idx:
        top = refmodel.make_topology(pods[i], ref)
        res = ref_loader.find_node(nhd_oracle.initial_node_filter(nodes, groups[i]), top)
        out.append(res[0])
    return lo, out, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--nodes", type=int, default=65536)
    ap.add_argument("--pods-total", type=int, default=4096)
    ap.add_argument("--pods", type=int, default=2, help="sampled pods (evenly spaced over the batch)")
    ap.add_argument("--procs", type=int, default=os.cpu_count() or 1)
    args = ap.parse_args()
    logging.disable(logging.CRITICAL)
    from workload import refmodel, synth
    from oracle import coracle, nhd_oracle, ref_loader
    idx = [int(k * args.pods_total / args.pods) for k in range(args.pods)]
    ref, spec, nodes, clock = build(args.config, args.nodes)
    pods, groups = synth.make_pods(args.config, n_pods=args.pods_total)
    names = list(nodes)
    # (i) one core
    one, per_pod = [], []
    for i in idx:
        top = refmodel.make_topology(pods[i], ref)
        t0 = time.perf_counter()
        res = ref_loader.find_node(nhd_oracle.initial_node_filter(nodes, groups[i]), top)
        per_pod.append(time.perf_counter() - t0)
        one.append(res[0])
    # (ii) all host cores: contiguous node shards, winners merged in node order (+ the GPU-less preference, Matcher.py:393-421)
    bounds = [(args.nodes * k // args.procs, args.nodes * (k + 1) // args.procs) for k in range(args.procs)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(args.procs) as pool:
        parts = pool.map(shard_worker, [(args.config, args.nodes, lo, hi, idx, args.pods_total) for lo, hi in bounds])
    wall_mp = time.perf_counter() - t0
    busy_mp = max(p[2] for p in parts)
    merged = []
    for k, i in enumerate(idx):
        cands = [p[1][k] for p in sorted(parts) if p[1][k] is not None]
        top = refmodel.make_topology(pods[i])
        needs_gpu = any(len(pg.group_gpus) for pg in top.proc_groups)
        pick = None
        if cands:
            pick = cands[0]
            if not needs_gpu:
                nog = [c for c in cands if len(nodes[c].gpus) == 0]
                pick = nog[0] if nog else pick
        merged.append(pick)
    # parity: the C port of the product's checker on the same pods
    cl = coracle.Cluster.from_spec(spec)
    tops = [refmodel.make_topology(pods[i]) for i in idx]
    w, _ = cl.find(cl.pods_from_tops(tops, [groups[i] for i in idx]), spec.clock_now, want_feas=False, threads=1)
    port = [names[int(x)] if x >= 0 else None for x in w]
    assert port == one, (port, one)
    assert merged == one, (merged, one)
    n_eval = len(idx) * args.nodes
    # the pinned Python restatement on the same host: the sampled pods x all nodes, then bench.py's calibration sample
    own_nodes = spec.build_nodes()
    rest = []
    for i in idx:
        top = refmodel.make_topology(pods[i])
        t0 = time.perf_counter()
        r = nhd_oracle.find_node(nhd_oracle.initial_node_filter(own_nodes, groups[i]), top, spec.clock_now)
        rest.append(time.perf_counter() - t0)
        assert r[0] == one[len(rest) - 1]
    cal_n, cal_p = min(4096, args.nodes), 2
    sub = spec.shard(0, cal_n).build_nodes()
    t0 = time.perf_counter()
    for k in range(cal_p):
        nhd_oracle.find_node(nhd_oracle.initial_node_filter(sub, groups[k]), refmodel.make_topology(pods[k]), spec.clock_now)
    cal = time.perf_counter() - t0
    print(json.dumps({
        "what": "unmodified reference nhd.Matcher.Matcher().FindNode (nhd/Matcher.py:27-63), logging disabled, stdout swallowed, on bench.py's inputs",
        "config": args.config, "nodes": args.nodes, "pods_in_batch": args.pods_total, "sampled_pods": idx,
        "host": {"cpu": platform.processor() or open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"),
                 "cores": os.cpu_count(), "python": platform.python_version(), "where": "build container (the reference is not present on the GPU box)"},
        "one_core": {"seconds_per_pod": per_pod, "evals_per_s": n_eval / sum(per_pod), "decisions_per_s": len(idx) / sum(per_pod), "cores": 1},
        "all_cores": {"processes": args.procs, "wall_s_incl_cluster_build": wall_mp, "slowest_worker_findnode_s": busy_mp,
                      "evals_per_s": n_eval / busy_mp, "decisions_per_s": len(idx) / busy_mp, "cores": args.procs},
        "python_restatement": {"what": "oracle/nhd_oracle.py (linear in N; the reference's IntersectResources is quadratic) on the same pods x nodes, same host, 1 core",
                               "seconds_per_pod": rest, "evals_per_s": n_eval / sum(rest),
                               "calibration": {"sample": "first %d pods x first %d nodes (bench.py python_oracle_sample times the same on the GPU box)" % (cal_p, cal_n),
                                               "seconds": cal, "evals_per_s": cal_p * cal_n / cal}},
        "parity": "winners identical: reference (1 core) == reference (sharded, merged) == oracle C port, on every sampled pod"}))


if __name__ == "__main__":
    main()
