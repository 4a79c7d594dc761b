"""Vectorised packer of synthetic clusters (TEST / BENCH INFRASTRUCTURE, not part of the product package): the planes
nhd_amd.pack.Packer.pack_nodes(spec.build_nodes()) would produce, without materialising 65 536 node objects
(tests/test_core_vs_oracle.py::test_spec_route_equals_object_route asserts the equality)."""
import numpy as np

from nhd_amd.pack import (ALL_ONES, GLIMIT_NONE, MAX_GROUPS, MAX_NICS_PER_NUMA, NF_ACTIVE, NF_HAS_GPU, NF_MAINTENANCE, NF_SMT,
                          NIC_BW_AVAIL_PERCENT, NodeTable, empty_table)
from workload.synth import GROUP_NAMES


def planes_from_spec(packer, spec) -> NodeTable:
    """Same planes as pack_nodes(spec.build_nodes()) without materialising objects
    (tests/test_core_vs_oracle.py asserts equality).  Layout knowledge of synth.ClusterSpec.labels():
    2 sockets, GPU g on NUMA g//2 / switch g, NIC (numa,j) on switch numa*2 + (j % 2) or, with
    SR-IOV, on its physical function's switch numa*2 + j // (K/2); every NIC is 100 GbE."""
    n = spec.n
    t = empty_table(n)
    if n == 0:                                           # (a shard without nodes: more ranks than 64-node blocks)
        return t
    cpp = (spec.phys // 2).astype(np.uint64)
    if int(cpp.max()) > packer.max_cores_per_numa:
        packer.max_cores_per_numa = int(cpp.max())
        packer.dict_version += 1
    valid = (np.uint64(1) << cpp) - np.uint64(1)
    free = (~spec.core_used) & valid[:, None]
    t.p0["t0"] = free
    t.p1["t1"] = np.where(spec.smt[:, None], free, ALL_ONES)
    has_gpu = spec.n_gpus > 0
    if has_gpu.any() and packer.max_gpus_per_numa < 2:
        packer.max_gpus_per_numa = 2
        packer.dict_version += 1
    gvalid = np.where(has_gpu, 0xF, 0).astype(np.uint32)
    t.p2["gpu_free"] = gvalid & ~spec.gpu_used
    t.p2["gpu_numa1"] = gvalid & np.uint32(0xC)
    t.p2["hp_free"] = spec.hp_free
    t.p2["flags"] = (np.where(spec.maintenance, NF_MAINTENANCE, 0) | np.where(spec.active, NF_ACTIVE, 0) |
                     np.where(spec.smt, NF_SMT, 0) | np.where(has_gpu, NF_HAS_GPU, 0)).astype(np.uint32)
    lut = np.array([packer.group_bits([nm]) for nm in GROUP_NAMES], dtype=np.uint64)
    gb = np.zeros(n, np.uint64)
    for k in range(16):
        gb |= np.where((spec.group_bits >> k) & 1, lut[k], np.uint64(0)).astype(np.uint64)
    t.p3["groups"] = gb
    uniq, inv = np.unique(gb, return_inverse=True)
    ids = np.array([packer.group_set_id(int(u)) for u in uniq], dtype=np.uint32)
    t.p4["group_set"] = ids[inv]
    t.p4["busy_time"] = np.where(spec.busy, spec.clock_now - 5.0, spec.clock_now - 1000.0)

    K = spec.nics_per_numa
    c_used = packer.cap_class(0)
    c_free = packer.cap_class(100000 / 1e3 * NIC_BW_AVAIL_PERCENT)
    half = K // 2 if spec.sriov else None
    det = t.detail
    det["numa_nodes"] = 2
    det["nic_cnt"] = K
    det["n_gpus"] = np.where(has_gpu, 4, 0)
    det["gpu_sw"][:, :4] = np.where(has_gpu[:, None], np.arange(4)[None, :], 0)
    # local switch ids follow first appearance: GPUs (switch g -> id g) then NICs
    sw_of_nic = np.zeros((2, K), np.int64)
    for numa in range(2):
        for j in range(K):
            sw_of_nic[numa, j] = numa * 2 + (j // half if spec.sriov else j % 2)
    gfree_sw = np.stack([((t.p2["gpu_free"] >> g) & 1) for g in range(4)], axis=1).astype(np.uint8)   # [n,4]
    # nodes without GPUs number their switches in NIC order instead
    nic_order = []
    for numa in range(2):
        for j in range(K):
            s = int(sw_of_nic[numa, j])
            if s not in nic_order:
                nic_order.append(s)
    local_nogpu = {s: k for k, s in enumerate(nic_order)}
    used_bits = spec.nic_used
    pods_word = np.zeros((2, n), np.uint64)                  # 16 three-bit counters per NUMA node = 48 bits each
    for numa in range(2):
        for j in range(K):
            used = ((used_bits >> (numa * K + j)) & 1).astype(bool)
            det["nic_cls"][:, numa, j] = np.where(used, c_used, c_free)
            pods_word[numa] |= used.astype(np.uint64) << np.uint64(3 * j)                   # pods_used = 1 on a used NIC
            t.origin["nic_base"][:, numa, j] = c_free
            s = int(sw_of_nic[numa, j])
            det["nic_sw"][:, numa, j] = np.where(has_gpu, s, local_nogpu[s])
    det["sw_free"][:, :4] = np.where(has_gpu[:, None], gfree_sw, 0)
    for b in range(12):
        det["nic_pods"][:, b] = ((pods_word[b // 6] >> np.uint64(8 * (b % 6))) & np.uint64(0xFF)).astype(np.uint8)
    t.origin["t0"] = valid[:, None] & ~np.uint64(3)          # cores 0, 1 of each socket are reserved (synth.ClusterSpec.labels)
    t.origin["t1"] = np.where(spec.smt[:, None], valid[:, None] & ~np.uint64(3), ALL_ONES)
    t.origin["hp_total"] = getattr(spec, "hp_total", 64)

    # signatures: enumerate the distinct (used-count per pool, free GPUs per pool) patterns
    for numa in range(2):
        pool_ids = sorted(set(int(s) for s in sw_of_nic[numa]))
        n_used_pool = {s: np.zeros(n, np.int64) for s in pool_ids}
        n_tot_pool = {s: 0 for s in pool_ids}
        for j in range(K):
            s = int(sw_of_nic[numa, j])
            n_used_pool[s] += (used_bits >> (numa * K + j)) & 1
            n_tot_pool[s] += 1
        n_used_all = sum(n_used_pool.values())
        # NUMA-mode signature: one pool, all NICs
        key_numa = n_used_all
        sig_numa = np.zeros(n, np.uint16)
        for v in np.unique(key_numa):
            d = {}
            if v:
                d[c_used] = int(v)
            if K - v:
                d[c_free] = int(K - v)
            pairs = tuple(sorted((c, min(m, MAX_GROUPS)) for c, m in d.items()))
            sig_numa[key_numa == v] = packer.sig_id([(GLIMIT_NONE, pairs)])
        t.p3["sig_numa"][:, numa] = sig_numa
        # PCI-mode signature: one pool per switch, limited by that switch's free GPUs
        code = np.zeros(n, np.int64)
        for s in pool_ids:
            code = code * 64 + n_used_pool[s] * 2 + np.where(has_gpu, gfree_sw[:, s], 0)
        sig_pci = np.zeros(n, np.uint16)
        for v in np.unique(code):
            sel = code == v
            first = int(np.argmax(sel))
            pools = []
            for s in pool_ids:
                gl = int(gfree_sw[first, s]) if has_gpu[first] else 0
                if gl <= 0:
                    continue
                nu = int(n_used_pool[s][first])
                d = {}
                if nu:
                    d[c_used] = nu
                if n_tot_pool[s] - nu:
                    d[c_free] = n_tot_pool[s] - nu
                pools.append((min(gl, MAX_GROUPS), tuple(sorted((c, min(m, MAX_GROUPS)) for c, m in d.items()))))
            sig_pci[sel] = packer.sig_id(pools)
        t.p3["sig_pci"][:, numa] = sig_pci
    t.names = []
    return t
