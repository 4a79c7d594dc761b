#!/usr/bin/env python3
"""CPU model of the fit role's LDS row fetches (no GPU needed): how many LDS cycles the ds_read_b128 gathers of one 64-node chunk
cost on gfx950 - a wavefront's access is serviced in four fixed groups of 16 lanes, a group takes as many cycles as its deepest
16-byte bank group holds DIFFERENT rows (MI355X_MICROARCH.md, LDS) - for the node order of a BASELINE cluster and for the order
step_kernel.h k_xorder deals the chunk's records in (the same greedy, restated).  Round 5 used it to decide what to build:

    python tools/lds_bank_model.py [config] [nodes]          (default: config 4, 65 536 nodes; ~1 min)

Rows of the model per row width W: pair form (W = 2, 4): C row (plane-split: bank group = row mod 16), the two class rows;
six-fetch form (W = 8): both sockets' CPU rows (stride 144 B) and the two class rows (stride 80 B)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nhd_amd import pack
from workload import planes, refmodel, synth

GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
GROUPS += [[x + 32 for x in g] for g in GROUPS]


def cycles(keys_by_lane):
    """keys_by_lane: [families][64] row ids (bank group = id mod 16) in LANE order -> LDS cycles of one fetch per family."""
    tot = 0
    for fam in keys_by_lane:
        for g in GROUPS:
            rows = {}
            for lane in g:
                rows.setdefault(fam[lane] % 16, set()).add(fam[lane])
            tot += max(len(v) for v in rows.values())
    return tot


def deal(keys, weights):
    """step_kernel.h k_xorder: records dealt one after the other to the lane group where they add the fewest turns."""
    K = len(keys)
    load = [[[0] * 16 for _ in range(K)] for _ in range(4)]
    rows = [[set() for _ in range(K)] for _ in range(4)]
    deepest = [[1] * K for _ in range(4)]
    members = [[] for _ in range(4)]
    for i in range(64):
        best = None
        for g in range(4):
            if len(members[g]) >= 16:
                continue
            inc, fresh = 0, []
            for k in range(K):
                f = load[g][k][keys[k][i] % 16] + (0 if keys[k][i] in rows[g][k] else 1)
                fresh.append(f)
                if f > deepest[g][k]:
                    inc += weights[k] * (f - deepest[g][k])
            cost = inc * 32 + len(members[g])
            if best is None or cost < best[0]:
                best = (cost, g, fresh)
        _, g, fresh = best
        for k in range(K):
            if keys[k][i] not in rows[g][k]:
                rows[g][k].add(keys[k][i]); load[g][k][keys[k][i] % 16] += 1
            deepest[g][k] = max(deepest[g][k], fresh[k])
        members[g].append(i)
    lane_of = [0] * 64
    for g in range(4):
        for idx, i in enumerate(members[g]):
            lane_of[i] = GROUPS[g][idx]
    return lane_of


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    spec = synth.make_cluster(cfg, n_nodes=n)
    pk = pack.Packer(); t = planes.planes_from_spec(pk, spec)
    pc = np.vectorize(lambda v: bin(int(v)).count("1"))
    free = t.p0["t0"] & t.p1["t1"]
    c0, c1 = pc(free[:, 0]), pc(free[:, 1])
    smt = ((t.p2["flags"] & pack.NF_SMT) != 0).astype(int)
    gf, g1 = t.p2["gpu_free"], t.p2["gpu_numa1"]
    ids = {}
    k0 = np.array([ids.setdefault((0, int(a), int(b), int(c)), len(ids)) for a, b, c in zip(pc(gf & ~g1), t.p3["sig_numa"][:, 0], t.p3["sig_pci"][:, 0])])
    k1 = np.array([ids.setdefault((1, int(a), int(b), int(c)), len(ids)) for a, b, c in zip(pc(gf & g1), t.p3["sig_numa"][:, 1], t.p3["sig_pci"][:, 1])])
    fc = int(max(c0.max(), c1.max())) + 1
    crow = lambda D: (smt * D + np.minimum(c0, D - 1)) * D + np.minimum(c1, D - 1)
    forms = {  # W: (row ids per fetch family, weights, fetches per family and chunk)
        2: ([crow(14), k0, k1], [1, 1, 1], [1, 1, 1]),
        4: ([crow(24), k0 * 3, k1 * 3], [1, 1, 1], [2, 2, 2]),
        8: ([(smt * fc + c0) * 9, (smt * fc + c1) * 9, k0 * 5, k1 * 5], [2, 2, 1, 1], [8, 8, 4, 4]),
    }
    rng = np.random.default_rng(0)
    chunks = rng.choice(n // 64, min(200, n // 64), replace=False)
    print(f"config {cfg}, {n} nodes, {len(ids)} node classes; LDS cycles of the row fetches per 64-node chunk (ideal = 4 per fetch)")
    for W, (fams, wts, reps) in forms.items():
        nat = dealt = ideal = 0
        for c in chunks:
            keys = [[int(f[c * 64 + i]) for i in range(64)] for f in fams]
            lane_of = deal(keys, wts)
            inv = [0] * 64
            for i, l in enumerate(lane_of):
                inv[l] = i
            assert sorted(lane_of) == list(range(64))
            nat_k = [cycles([k]) for k in keys]
            dealt_k = [cycles([[k[inv[l]] for l in range(64)]]) for k in keys]
            nat += sum(r * x for r, x in zip(reps, nat_k)); dealt += sum(r * x for r, x in zip(reps, dealt_k)); ideal += 4 * sum(reps)
        m = len(chunks)
        print(f"  W = {W}: node order {nat / m:6.1f}   dealt {dealt / m:6.1f}   ideal {ideal / m:5.0f}   (conflict share {1 - ideal / nat:.2f} -> {1 - ideal / dealt:.2f})")


if __name__ == "__main__":
    main()
