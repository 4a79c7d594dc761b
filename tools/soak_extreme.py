"""Offline soak at the EDGES of the record formats (not part of the suite; `python tools/soak_extreme.py <seeds> [first seed] [--ref]`,
CPU only; `--ref` needs /root/reference): clusters and pods drawn so that the limits of include/nhdfit.h are met often - sockets
of 1..64 physical cores next to wide ones of 65..128, up to 16 NICs and 8 GPUs per NUMA node, a dozen distinct NIC speeds (the
capacity classes), up to 14 PCIe switches, pods_used of 0..3, arbitrary isolcpus sets, busy times on either side of the 30 s window,
hugepage requests around the tile's table (1 022 GiB) - and pods of 1..6 processing groups with core counts that often do not fit.
The product's host build (HipMatcher on tests/harness: the kernels' own arithmetic compiled for the host) against the Python oracle
- FindNodes, then ScheduleBatch with commits and physical ids - and with --ref against the UNMODIFIED reference Matcher on every
pod the reference answers in reasonable time."""
import os
import re
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from nhd_amd.matcher import HipMatcher  # noqa: E402
from oracle import nhd_oracle as O  # noqa: E402
from tests import harness, sched_standin, util  # noqa: E402
from tests import delta_check as D  # noqa: E402
from tests.test_wide_core import norm  # noqa: E402
from workload import refmodel  # noqa: E402
from workload.refmodel import NFD  # noqa: E402

SPEEDS = [9000, 10999, 11000, 12000, 20000, 25000, 40000, 50000, 56000, 100000, 200000, 400000, 11001, 33000]   # two below Node.py:403's threshold
NAMES = ["default"] + ["grp%02d" % k for k in range(40)]


def edge_labels(rng, heavy, few=False, beyond=False):
    wide = rng.random() < 0.25
    if wide:
        sockets = int(rng.choice([1, 2, 3, 4], p=[0.1, 0.4, 0.25, 0.25]))
        cpp = int(rng.choice([65, 66, 96, 127, 128, 7, 20]))
    else:
        sockets = int(rng.choice([1, 2], p=[0.2, 0.8]))
        cpp = int(rng.choice([2, 3, 5, 8, 16, 31, 32, 33, 48, 63, 64]))
    phys = cpp * sockets
    smt = rng.random() < 0.6
    lab = {NFD + "nfd-extras-cpu.numSockets": str(sockets), NFD + "nfd-extras-cpu.num_cores": str(phys)}
    if smt:
        lab[NFD + "cpu-hardware_multithreading"] = "true"
    mode = rng.random()
    if mode < 0.5:                                         # arbitrary isolcpus: a few ranges anywhere in the logical id space
        total = phys * (2 if smt else 1)
        spans = []
        for _ in range(int(rng.integers(1, 5))):
            a = int(rng.integers(0, total))
            spans.append((a, min(total - 1, a + int(rng.integers(0, max(1, total // 2))))))
        lab[NFD + "nfd-extras-cpu.isolcpus"] = "_".join(f"{a}-{b}" for a, b in spans)
    elif mode < 0.8:
        spans = [(s * cpp + 1, (s + 1) * cpp - 1) for s in range(sockets) if cpp > 1]
        if smt:
            spans += [(phys + s * cpp + 1, phys + (s + 1) * cpp - 1) for s in range(sockets) if cpp > 1]
        if spans:
            lab[NFD + "nfd-extras-cpu.isolcpus"] = "_".join(f"{a}-{b}" for a, b in spans)
    n_sw = int(rng.integers(1, 8 if wide else 7))          # switches per NUMA node (<= 14 per node on the fast layout)
    per_numa = [int(rng.integers(0, 17)) if heavy else int(rng.choice([0, 1, 2] if few else [0, 1, 2, 3, 4])) for _ in range(sockets)]
    if beyond and rng.random() < 0.5:
        per_numa[int(rng.integers(0, sockets))] = int(rng.integers(17, 21))      # more NICs on a NUMA node than any record holds: the node never matches
    speeds = rng.choice(SPEEDS, size=int(rng.integers(1, 5)), replace=False)
    j = 0
    for numa in range(sockets):
        for _ in range(per_numa[numa]):
            sw = 0x10 * (numa + 1) + int(rng.integers(0, n_sw))
            if wide and rng.random() < 0.05:
                sw = 0x90                                  # one switch seen from several NUMA nodes (general path only)
            lab[NFD + f"nfd-extras-nic.eth{j}.mlx.{0xABE000 + j:012x}.{int(rng.choice(speeds))}Mbs.{numa}.{sw:x}.{j}.0"] = "true"
            j += 1
    g = 0
    for numa in range(sockets):
        for _ in range(9 if beyond and rng.random() < 0.3 else int(rng.choice([0, 1, 2, 4, 8], p=[0.4, 0.15, 0.2, 0.15, 0.1]))):
            if g >= 32:
                break
            sw = 0x10 * (numa + 1) + int(rng.integers(0, n_sw))
            lab[NFD + f"nfd-extras-gpu.{g}.V100.{numa}.{sw:x}"] = "true"
            g += 1
    lab["DATA_PLANE_VLAN"] = "7"
    lab["DATA_DEFAULT_GW"] = "10.1.0.1/32"
    if rng.random() < 0.6:
        lab["NHD_GROUP"] = ".".join(rng.choice(NAMES, size=int(rng.integers(1, 5)), replace=False))
    if rng.random() < 0.04:
        lab[refmodel.MAINT_LABEL] = "scheduled"
    return lab


def edge_node(rng, name, heavy, occupancy, few=False, beyond=False):
    lab = edge_labels(rng, heavy, few, beyond)
    phys = int(lab[NFD + "nfd-extras-cpu.num_cores"])
    smt = (NFD + "cpu-hardware_multithreading") in lab
    used = []
    for c in range(phys):
        r = rng.random()
        if r < occupancy:
            used.append(c)
            if smt and rng.random() < 0.7:
                used.append(c + phys)
        elif smt and r < occupancy + 0.08:
            used.append(c + phys)
    ngpu = sum(1 for k in lab if "nfd-extras-gpu" in k)
    nnic = 0
    for k in lab:
        if "nfd-extras-nic" in k and int(re.search(r"\.(\d+)Mbs\.", k).group(1)) >= 11000:          # (Node.py:403: slower NICs are not kept)
            nnic += 1
    return dict(name=name, labels=lab, hugepages=[2048, int(rng.choice([0, 1, 16, 1021, 1022, 1023, 2047]))], active=bool(rng.random() > 0.04),
                used_cores=sorted(used), used_gpus=[g for g in range(ngpu) if rng.random() < 0.3],
                nic_pods_used=[int(rng.choice([0, 0, 0, 1, 2, 3])) for _ in range(nnic)],
                busy_time=util.CLOCK - float(rng.choice([0.0, 29.99, 30.0, 30.01, 500.0, 500.0, 500.0])))


def edge_pod(rng, max_groups):
    groups = []
    G = int(rng.integers(1, max_groups + 1))
    for _ in range(G):
        ng = int(rng.choice([0, 1, 2, 3], p=[0.5, 0.3, 0.15, 0.05]))
        groups.append(dict(proc=int(rng.choice([2, 2, 3, 4, 6, 9, 17, 33])) if G <= 3 else int(rng.integers(2, 5)),
                           helpers=int(rng.choice([0, 0, 1, 2, 5])),
                           rx=float(rng.choice([0, 0, 0.1, 1e-9, 5, 9.9, 10.8, 18, 22.5, 22.500001, 36, 45, 50.4, 89.99999, 90, 90.00001, 180, 360])),
                           tx=float(rng.choice([0, 0, 5, 9.9, 10.8, 12.25, 22.5, 45, 90, 180])),
                           proc_smt=bool(rng.random() < 0.5), helper_smt=bool(rng.random() < 0.5),
                           gpus=[int(rng.integers(0, 4)) for _ in range(ng)]))
    return dict(map_type=str(rng.choice(["NUMA", "PCI", "NONE", "BOGUS"], p=[0.5, 0.42, 0.04, 0.04])),
                hugepages_gb=int(rng.choice([0, 0, 1, 16, 17, 1021, 1022, 1023, 2000])), misc=int(rng.choice([0, 1, 2, 3, 7])),
                misc_smt=bool(rng.random() < 0.5), groups=groups)


def states_agree(nodes, m):
    """delta_check.state_of(nodes) == delta_check.mirror_state(m) for the nodes the five planes hold (a wide node's entry there is a
    placeholder; its record is re-uploaded whole and the finds that follow check it)."""
    from nhd_amd import pack
    skip = set(m.wide_nodes) | set(m.unmirrored)
    pk = pack.Packer()
    t_obj = pk.pack_nodes(nodes)
    t_dev = m.engine.download()
    for i, name in enumerate(m._names):
        if name in skip:
            continue
        if D._row(pk, t_obj, i) != D._row(m.packer, t_dev, i):
            print("   state differs on", name, D._row(pk, t_obj, i), D._row(m.packer, t_dev, i), flush=True)
            return False
        sn, sp = m.packer.sigs_from_detail(t_dev.detail[i])
        if [int(x) for x in t_dev.p3[i]["sig_numa"]] != sn or [int(x) for x in t_dev.p3[i]["sig_pci"]] != sp:
            print("   signature ids differ on", name, flush=True)
            return False
        if m.packer.group_sets[int(t_dev.p4[i]["group_set"])] != int(t_dev.p3[i]["groups"]):
            print("   group set differs on", name, flush=True)
            return False
    return True


DEVICE = "--device" in sys.argv          # the same checks through the C-ABI on the GPU (the sharded parts need three devices: host twin only)
EF = {} if DEVICE else {"engine_factory": harness.HarnessEngine}


def main():
    n_seeds = int(sys.argv[1])
    first = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 0
    with_ref = "--ref" in sys.argv
    ref = None
    if with_ref:
        from oracle import ref_loader
        ref = ref_loader.load()
        ref_loader.VirtualClock(util.CLOCK).install()
    t0 = time.time()
    bad = pods = placed = refchecked = unmirrored = streams = sharded = 0
    for seed in range(first, first + n_seeds):
        rng = np.random.default_rng(880000 + seed)
        heavy_share = 0.12 if seed % 3 == 0 else 0.0
        # (the oracle enumerates K^G NIC choices per NUMA assignment in Python: NIC-heavy nodes meet pods of one or two groups, pods of
        # five or six groups meet nodes of at most two NICs per NUMA node)
        max_groups = 2 if heavy_share else 6 if seed % 5 == 0 else 4 if seed % 2 else 3
        # seed % 7 == 3: a few nodes beyond EVERY record (17..20 NICs or nine GPUs on a NUMA node) - they never match (listed in
        # HipMatcher.unmirrored), every other node is answered for as the oracle answers without them
        descs = [edge_node(rng, f"e{i:04d}", rng.random() < heavy_share, occupancy=float(rng.choice([0.0, 0.1, 0.3, 0.6])), few=max_groups > 4,
                           beyond=bool(heavy_share) and seed % 7 == 3 and rng.random() < 0.3) for i in range(14)]
        nl = util.build_cluster(descs)
        specs = [edge_pod(rng, max_groups) for _ in range(16)]
        tops = [refmodel.make_topology(s) for s in specs]
        pgs = [list(rng.choice(NAMES, size=int(rng.integers(1, 4)), replace=False)) if rng.random() < 0.3 else None for _ in specs]
        m = HipMatcher(clock=lambda: util.CLOCK, **EF)
        m.attach(nl)
        unmirrored += len(m.unmirrored)
        live = {k: v for k, v in nl.items() if k not in m.unmirrored}      # (a node no record holds never matches: a documented deviation)
        got = m.FindNodes(nl, tops)
        for p, (s, top) in enumerate(zip(specs, tops)):
            want = O.find_node(live, top, util.CLOCK)
            pods += 1
            placed += want[0] is not None
            if norm(got[p]) != norm(want):
                bad += 1
                print("FIND product != oracle seed", seed, "pod", p, s, norm(got[p]), norm(want), flush=True)
            if ref is not None and len(s["groups"]) <= 3 and not heavy_share:
                nl_ref = util.build_cluster([d for d in descs if d["name"] in live], ref)
                rwant = ref_loader.find_node(nl_ref, refmodel.make_topology(s, ref))
                refchecked += 1
                if norm(rwant) != norm(want):
                    bad += 1
                    print("FIND oracle != REFERENCE seed", seed, "pod", p, s, norm(want), norm(rwant), flush=True)
        if seed % 4 == 1 and not DEVICE:                    # the same through a mirror sharded over three host-twin shards (engine.GroupEngine)
            ms = HipMatcher(clock=lambda: util.CLOCK, devices=[0, 1, 2], **EF)
            ms.attach(util.build_cluster(descs))
            gs = ms.FindNodes(ms._attached, tops)
            if [norm(x) for x in gs] != [norm(x) for x in got]:
                bad += 1
                print("SHARDED FIND mismatch seed", seed, flush=True)
            sharded += 1
        # InitialNodeFilter in front (filtered dict handed to FindNode) for the pods that carry groups
        for p, (top, pg) in enumerate(zip(tops, pgs)):
            if pg is None:
                continue
            sub = O.initial_node_filter(nl, pg)
            g1 = m.FindNode(sub, top)
            want = O.find_node({k: v for k, v in sub.items() if k in live}, top, util.CLOCK)
            if norm(g1) != norm(want):
                bad += 1
                print("FILTERED FIND mismatch seed", seed, "pod", p, pg, norm(g1), norm(want), flush=True)
        # mode B on twin clusters: the product's batch against the oracle's sequential loop with commits
        nl_b, nl_o = util.build_cluster(descs), util.build_cluster(descs)
        for s in specs:
            if s["map_type"] not in ("NUMA", "PCI"):
                s["map_type"] = "NUMA"
        tops = [refmodel.make_topology(s) for s in specs]
        mb = HipMatcher(clock=lambda: util.CLOCK, **EF)
        mb.attach(nl_b)
        live_o = {k: v for k, v in nl_o.items() if k not in mb.unmirrored}
        try:
            res = mb.ScheduleBatch(nl_b, tops, now=util.CLOCK)
        except AssertionError as e:
            print("mode B assertion seed", seed, e, flush=True)
            bad += 1
            continue
        want, ids = [], []
        for top in tops:
            r = O.find_node(live_o, top, util.CLOCK)
            rec = {}
            if r[0] is not None:
                try:
                    O.commit(live_o[r[0]], top, r[1], util.CLOCK, rec)
                except O.CommitFailure:
                    break                                   # the reference's commit step would raise here: the batch is defined up to this pod
            want.append(r)
            ids.append(rec if r[0] is not None else None)
        k = len(want)
        if seed % 4 == 1 and not DEVICE:
            nl_s = util.build_cluster(descs)
            ms = HipMatcher(clock=lambda: util.CLOCK, devices=[0, 1, 2], **EF)
            ms.attach(nl_s)
            rs = ms.ScheduleBatch(nl_s, tops, now=util.CLOCK)
            if [norm(x) for x in rs[:k]] != [norm(w) for w in want] or ms.last_placements[:k] != ids:
                bad += 1
                print("SHARDED MODE B mismatch seed", seed, flush=True)
        if [norm(x) for x in res[:k]] != [norm(w) for w in want]:
            bad += 1
            print("MODE B decisions mismatch seed", seed, [(norm(a), norm(b)) for a, b in zip(res[:k], want) if norm(a) != norm(b)][:2], flush=True)
        elif mb.last_placements[:k] != ids:
            bad += 1
            print("MODE B ids mismatch seed", seed, flush=True)
        # op streams on the edge cluster (attached mode): one ScheduleBatch(apply=True) for the pending list, the node objects brought
        # along by their own mutators, then releases / reclaims / resets / scalar writes mirrored as deltas - finds on the way
        # against the oracle on the objects' state, objects == device mirror at the end
        if seed % 2 == 0:
            clock = D.Clock(util.CLOCK)
            nodes = sched_standin.adopt(util.build_cluster(descs), clock)
            P = 12
            for s in specs:
                s["misc_smt"] = True                        # (the reference's own unwind path is broken)
            fresh_specs = [edge_pod(rng, min(3, max_groups)) for _ in range(D.N_FRESH)]
            for s in fresh_specs:
                s["misc_smt"] = True
            tops = [refmodel.make_topology(s) for s in specs[:P] + fresh_specs]
            grps = [["default"] + list(rng.choice(NAMES, size=2, replace=False)) for _ in tops]
            md = HipMatcher(clock=clock, **EF)
            md.attach(nodes)
            try:
                binds = sched_standin.check_pending_pods_batched(nodes, md, tops[:P], grps[:P], now=clock.t)
            except (IndexError, AssertionError, O.CommitFailure):   # a commit the reference itself would fail on (short of cores / of GPUs on the NIC's switch)
                binds = None
            if binds is not None:
                placed_b = [(i, b) for i, b in enumerate(binds) if b is not None]
                for k, op in enumerate(D.make_ops(seed, list(nodes), placed_b, 60, util.CLOCK)):
                    if op[0] == "find":
                        j = P + op[2]
                        sub = O.initial_node_filter(nodes, grps[j])
                        got1 = md.FindNode(sub, tops[j])
                        want1 = O.find_node({k2: v for k2, v in sub.items() if k2 not in md.unmirrored}, tops[j], clock.t)
                        if norm(got1) != norm(want1):
                            bad += 1
                            print("OP-STREAM find mismatch seed", seed, "op", k, norm(got1), norm(want1), flush=True)
                    else:
                        D.apply_op(nodes, tops, op)
                md.FindNode(nodes, tops[P])
                if not states_agree(nodes, md):
                    bad += 1
                    print("OP-STREAM objects != mirror seed", seed, flush=True)
                final = md.FindNodes(nodes, tops[P:])
                for j, g1 in enumerate(final):
                    w1 = O.find_node({k2: v for k2, v in nodes.items() if k2 not in md.unmirrored}, tops[P + j], clock.t)
                    if norm(g1) != norm(w1):
                        bad += 1
                        print("OP-STREAM final find mismatch seed", seed, "pod", j, norm(g1), norm(w1), flush=True)
                streams += 1
        if (seed - first) % 10 == 9:
            print("seed", seed, "pods", pods, "placed", placed, "ref-checked", refchecked, "unmirrored nodes", unmirrored, "op streams", streams, "sharded", sharded, "mismatches", bad,
                  "seconds", round(time.time() - t0, 1), flush=True)
    print("seeds", n_seeds, "from", first, "pods", pods, "placed", placed, "ref-checked", refchecked, "unmirrored nodes", unmirrored,
          "op streams", streams, "sharded", sharded, "mismatches", bad, "seconds", round(time.time() - t0, 1))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
