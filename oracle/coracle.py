"""ctypes front-end of oracle/nhd_oracle.c (TEST INFRASTRUCTURE, see the C file's header).

`Cluster.from_nodes(nl)` flattens duck-typed node objects, `Cluster.from_spec(spec)` flattens a
synthetic ClusterSpec without materialising objects (tests check both agree), `pods_from_tops`
flattens CfgTopology-like requests.  No product code (nhd_amd.pack / the HIP library) is involved.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libnhdoracle.so")
MAXG = 8

ONODE = np.dtype([("numa_nodes", "<i4"), ("smt", "<i4"), ("n_scan", "<i4"), ("core_off", "<i4"), ("n_cores", "<i4"),
                  ("gpu_off", "<i4"), ("n_gpus", "<i4"), ("nic_off", "<i4"), ("n_nics", "<i4"), ("hp_free", "<i4"),
                  ("maintenance", "<i4"), ("active", "<i4"), ("groups", "<u8"), ("busy_time", "<f8")])
OPOD = np.dtype([("G", "<i4"), ("map_type", "<i4"), ("hp", "<i4"), ("n_misc", "<i4"), ("misc_smt_truthy", "<i4"),
                 ("use_filter", "<i4"), ("pad0", "<i4"), ("pad1", "<i4"),
                 ("n_gpus", "<i4", (MAXG,)), ("n_proc", "<i4", (MAXG,)), ("proc_smt", "<i4", (MAXG,)),
                 ("n_help", "<i4", (MAXG,)), ("help_smt", "<i4", (MAXG,)),
                 ("rx", "<f8", (MAXG,)), ("tx", "<f8", (MAXG,)), ("groups", "<u8")])


class _OCluster(ctypes.Structure):
    _fields_ = [("nodes", ctypes.c_void_p), ("n", ctypes.c_int64),
                ("core_used", ctypes.c_void_p), ("core_socket", ctypes.c_void_p), ("core_sibling", ctypes.c_void_p),
                ("gpu_used", ctypes.c_void_p), ("gpu_numa", ctypes.c_void_p), ("gpu_sw", ctypes.c_void_p),
                ("nic_numa", ctypes.c_void_p), ("nic_speed", ctypes.c_void_p), ("nic_pods", ctypes.c_void_p),
                ("nic_sw", ctypes.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(HERE, "nhd_oracle.c")
        if not os.path.exists(SO) or os.path.getmtime(src) > os.path.getmtime(SO):
            subprocess.check_call(["make", "-s", "-C", HERE])
        _lib = ctypes.CDLL(SO)
        assert _lib.oracle_sizeof_node() == ONODE.itemsize and _lib.oracle_sizeof_pod() == OPOD.itemsize
    return _lib


def usable_cpus() -> int:
    """CPUs this process may really use: its affinity set, cut down by a cgroup CPU quota if there is one (a container on a 256-core
    host may be allowed a few cores' worth of time; OpenMP teams sized by os.cpu_count() then spend their time spinning)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts and parts[0] != "max":
                    n = min(n, max(1, -(-int(parts[0]) // int(parts[1]))))
            else:
                quota = int(parts[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        n = min(n, max(1, -(-quota // int(f2.read().split()[0]))))
            break
        except (OSError, ValueError, IndexError, ZeroDivisionError):
            continue
    return max(1, n)


class GroupIds:
    def __init__(self):
        self.ids = {}

    def bits(self, names):
        b = 0
        for nm in names:
            b |= 1 << self.ids.setdefault(nm, len(self.ids))
        return b


class Cluster:
    def __init__(self, nodes, arrays, group_ids):
        self.nodes = nodes
        self.arrays = arrays
        self.group_ids = group_ids
        self.n = len(nodes)

    @staticmethod
    def from_nodes(nl, group_ids=None):
        gid = group_ids or GroupIds()
        nodes = np.zeros(len(nl), ONODE)
        cu, cs, cb, gu, gn, gs, nn, ns, npods, nsw = ([] for _ in range(10))
        for i, v in enumerate(nl.values()):
            nodes[i] = (v.numa_nodes, int(v.smt_enabled), v.cores_per_proc * v.sockets, len(cu), len(v.cores),
                        len(gu), len(v.gpus), len(nn), len(v.nics), int(v.mem.free_hugepages_gb), int(v.maintenance),
                        int(v.active), gid.bits(v.groups), float(v.busy_time))
            for c in v.cores:
                cu.append(int(c.used)); cs.append(c.socket); cb.append(c.sibling)
            for g in v.gpus:
                gu.append(int(g.used)); gn.append(g.numa_node); gs.append(g.pciesw)
            for k in v.nics:
                nn.append(k.numa_node); ns.append(float(k.speed)); npods.append(int(k.pods_used)); nsw.append(k.pciesw)
        arrays = dict(core_used=np.asarray(cu, np.uint8), core_socket=np.asarray(cs, np.int32),
                      core_sibling=np.asarray(cb, np.int32), gpu_used=np.asarray(gu, np.uint8),
                      gpu_numa=np.asarray(gn, np.int32), gpu_sw=np.asarray(gs, np.int32),
                      nic_numa=np.asarray(nn, np.int32), nic_speed=np.asarray(ns, np.float64),
                      nic_pods=np.asarray(npods, np.int32), nic_sw=np.asarray(nsw, np.int32))
        return Cluster(nodes, arrays, gid)

    @staticmethod
    def from_spec(spec, group_ids=None):
        """Vectorised flattening of workload.synth.ClusterSpec (layout facts: see ClusterSpec.labels)."""
        from workload import synth
        gid = group_ids or GroupIds()
        n, K = spec.n, spec.nics_per_numa
        phys = spec.phys.astype(np.int64)
        cpp = phys // 2
        ncores = np.where(spec.smt, 2 * phys, phys)
        core_off = np.concatenate([[0], np.cumsum(ncores)[:-1]])
        total = int(ncores.sum())
        used = np.zeros(total, np.uint8); sock = np.zeros(total, np.int32); sib = np.full(total, -1, np.int32)
        for s in range(2):
            for b in range(32):
                m = b < cpp
                bit = ((spec.core_used[:, s] >> np.uint64(b)) & np.uint64(1)).astype(np.uint8)
                lid = s * cpp + b
                pos = core_off + lid
                used[pos[m]] = bit[m]; sock[pos[m]] = s
                msm = m & spec.smt
                sib[pos[msm]] = (lid + phys)[msm]
                pos2 = core_off + lid + phys
                used[pos2[msm]] = bit[msm]; sock[pos2[msm]] = s; sib[pos2[msm]] = lid[msm]
        ng = spec.n_gpus.astype(np.int64)
        gpu_off = np.concatenate([[0], np.cumsum(ng)[:-1]])
        tg = int(ng.sum())
        gu = np.zeros(tg, np.uint8); gn = np.zeros(tg, np.int32); gs = np.zeros(tg, np.int32)
        has = ng > 0
        for g in range(4):
            pos = gpu_off[has] + g
            gu[pos] = ((spec.gpu_used[has] >> g) & 1).astype(np.uint8); gn[pos] = g // 2; gs[pos] = synth.SWITCH_IDS[g]
        nic_off = np.arange(n, dtype=np.int64) * 2 * K
        nn = np.zeros(n * 2 * K, np.int32); nsp = np.full(n * 2 * K, 100000 / 1e3); npd = np.zeros(n * 2 * K, np.int32)
        nsw = np.zeros(n * 2 * K, np.int32)
        for numa in range(2):
            for j in range(K):
                pos = nic_off + numa * K + j
                nn[pos] = numa
                npd[pos] = (spec.nic_used >> (numa * K + j)) & 1
                nsw[pos] = synth.SWITCH_IDS[numa * 2 + (j // (K // 2) if spec.sriov else j % 2)]
        lut = np.array([gid.bits([nm]) for nm in synth.GROUP_NAMES], dtype=np.uint64)
        gb = np.zeros(n, np.uint64)
        for k in range(16):
            gb |= np.where((spec.group_bits >> k) & 1, lut[k], np.uint64(0)).astype(np.uint64)
        nodes = np.zeros(n, ONODE)
        nodes["numa_nodes"] = 2; nodes["smt"] = spec.smt; nodes["n_scan"] = phys
        nodes["core_off"] = core_off; nodes["n_cores"] = ncores; nodes["gpu_off"] = gpu_off; nodes["n_gpus"] = ng
        nodes["nic_off"] = nic_off; nodes["n_nics"] = 2 * K; nodes["hp_free"] = spec.hp_free
        nodes["maintenance"] = spec.maintenance; nodes["active"] = spec.active; nodes["groups"] = gb
        nodes["busy_time"] = np.where(spec.busy, spec.clock_now - 5.0, spec.clock_now - 1000.0)
        arrays = dict(core_used=used, core_socket=sock, core_sibling=sib, gpu_used=gu, gpu_numa=gn, gpu_sw=gs,
                      nic_numa=nn, nic_speed=nsp, nic_pods=npd, nic_sw=nsw)
        return Cluster(nodes, arrays, gid)

    def pods_from_tops(self, tops, pod_groups=None):
        pods = np.zeros(len(tops), OPOD)
        for i, top in enumerate(tops):
            p = pods[i]
            G = len(top.proc_groups)
            assert G <= MAXG
            p["G"] = G
            mt = getattr(top.map_type, "value", top.map_type)
            p["map_type"] = int(mt)
            p["hp"] = int(top.hugepages_gb)
            p["n_misc"] = len(top.misc_cores)
            p["misc_smt_truthy"] = int(bool(top.misc_cores_smt))
            for g, pg in enumerate(top.proc_groups):
                p["n_gpus"][g] = len(pg.group_gpus)
                p["n_proc"][g] = len(pg.proc_cores) + sum(len(x.cpu_cores) for x in pg.group_gpus)
                p["proc_smt"][g] = int(bool(pg.proc_smt.value))
                p["n_help"][g] = len(pg.misc_cores)
                p["help_smt"][g] = int(bool(pg.helper_smt.value))
                rx = tx = 0
                for c in pg.proc_cores:
                    d = getattr(c.nic_dir, "value", c.nic_dir)
                    if d == 1:
                        rx += c.nic_speed
                    elif d == 2:
                        tx += c.nic_speed
                p["rx"][g] = float(rx); p["tx"][g] = float(tx)
            if pod_groups is not None and pod_groups[i] is not None:
                p["use_filter"] = 1
                p["groups"] = self.group_ids.bits(pod_groups[i])
        return pods

    def find(self, pods, now, want_feas=True, threads=1):
        L = lib()
        threads = max(1, min(int(threads), usable_cpus()))   # never more threads than this process may run on (CPU set, cgroup quota)
        os.environ["OMP_NUM_THREADS"] = str(threads)
        try:
            ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(threads))
        except OSError:
            pass
        a = {k: np.ascontiguousarray(v) for k, v in self.arrays.items()}
        nodes = np.ascontiguousarray(self.nodes)
        oc = _OCluster(nodes.ctypes.data, self.n, *[a[k].ctypes.data for k in
                       ("core_used", "core_socket", "core_sibling", "gpu_used", "gpu_numa", "gpu_sw", "nic_numa",
                        "nic_speed", "nic_pods", "nic_sw")])
        pods = np.ascontiguousarray(pods)
        winner = np.zeros(len(pods), np.int64)
        feas = np.zeros((len(pods), self.n), np.uint8) if want_feas else None
        L.oracle_find(ctypes.byref(oc), pods.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(len(pods)),
                      ctypes.c_double(now), winner.ctypes.data_as(ctypes.c_void_p),
                      feas.ctypes.data_as(ctypes.c_void_p) if want_feas else None)
        return winner, feas
