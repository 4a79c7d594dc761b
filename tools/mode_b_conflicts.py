#!/usr/bin/env python3
"""CPU only: the dependency structure of a mode-B batch, from the independent oracle's decisions (oracle/seq_oracle.py).
Which pods really wait for which?  A pod's choice depends on an earlier pod only through the nodes both touch, and a commit
only ever removes resources (feasibility is monotone): pod k's pick v is invalidated by an earlier pod j only if j commits
to v itself.  So the chain the decision engine walks today (one wavefront, ~8 us per committed GPU-less pod) is mostly
artificial where pods land on different nodes.    python tools/mode_b_conflicts.py [config nodes pods [--candidates]]
--candidates: for the GPU-less pods that end on nodes with GPUs, how many snapshot-feasible candidates precede the winner and how
many of those an earlier pod of the batch touched (needs the P x N snapshot matrix of the C oracle)."""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import coracle, seq_oracle
from workload import refmodel, synth

cfg, n, P = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (4, 65536, 4096)
spec = synth.make_cluster(cfg, n_nodes=n)
pods, groups = synth.make_pods(cfg, n_pods=P)
tops = [refmodel.make_topology(s) for s in pods]
sc = seq_oracle.SeqCluster(coracle.Cluster.from_spec(spec))
t0 = time.time()
win, maps, ids, n_def = seq_oracle.schedule_sequence(sc, tops, groups, spec.clock_now)
print(f"config {cfg}: {n} nodes x {P} pods, oracle {time.time() - t0:.1f} s, {n_def} pods defined, {sum(w >= 0 for w in win)} placed")
gpu_less = [k for k, p in enumerate(pods) if sum(len(g["gpus"]) for g in p["groups"]) == 0]
with_gpu = [k for k in range(P) if k not in set(gpu_less)]
node_has_gpu = np.array([int(sc.nodes[i]["n_gpus"]) > 0 for i in range(sc.n)])
taken_by_gpu_pod = {win[k] for k in with_gpu if win[k] >= 0}


def describe(name, lst):
    seq = [win[k] for k in lst]
    if not seq:
        print(f"  {name}: none"); return
    cnt = collections.Counter(seq)
    last, gaps = {}, []
    for k in lst:
        if win[k] in last:
            gaps.append(k - last[win[k]])
        last[win[k]] = k
    print(f"  {name}: {len(seq)} pods on {len(cnt)} distinct nodes (at most {max(cnt.values())} per node); "
          f"{len(gaps)} of them land on a node an earlier pod of the same kind took" +
          (f" (median {int(np.median(gaps))} pods earlier, closest {min(gaps)})" if gaps else ""))


describe("pods with GPUs", [k for k in with_gpu if win[k] >= 0])
on_gpu_nodes = [k for k in gpu_less if win[k] >= 0 and node_has_gpu[win[k]]]
describe("GPU-less pods placed on nodes WITH GPUs", on_gpu_nodes)
print(f"     ... {sum(win[k] in taken_by_gpu_pod for k in on_gpu_nodes)} of them on a node a pod with GPUs of this batch also took")
describe("GPU-less pods placed on GPU-less nodes", [k for k in gpu_less if win[k] >= 0 and not node_has_gpu[win[k]]])

if "--candidates" in sys.argv and on_gpu_nodes:
    cl0 = coracle.Cluster.from_spec(spec)
    op = cl0.pods_from_tops([tops[k] for k in on_gpu_nodes], [groups[k] for k in on_gpu_nodes] if groups else None)
    _, feas = cl0.find(op, spec.clock_now, threads=min(16, os.cpu_count() or 1))
    pos = {k: i for i, k in enumerate(on_gpu_nodes)}
    touched, stats, n_cand, n_touched = {}, collections.Counter(), [], []
    for k in range(P):
        v = win[k]
        if v < 0:
            continue
        if k in pos:
            cands = np.flatnonzero(feas[pos[k]][:v + 1] & node_has_gpu[:v + 1])
            n_cand.append(len(cands)); n_touched.append(sum(1 for c in cands if c in touched))
            stats["winner untouched so far" if v not in touched else "winner touched before by " + "+".join(sorted(set(touched[v])))] += 1
        touched.setdefault(v, []).append("a GPU-less pod" if k in set(gpu_less) else "a pod with GPUs")
    print(f"  GPU-less pods on nodes with GPUs, snapshot-feasible candidates up to the winner: median {int(np.median(n_cand))}, max {max(n_cand)}; "
          f"of them touched by an earlier pod of the batch: median {int(np.median(n_touched))}, max {max(n_touched)}")
    for key, val in sorted(stats.items()):
        print(f"     {key}: {val}")
