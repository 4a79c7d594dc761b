"""Stand-in for the matcher-facing part of the reference scheduler loop, for boxes without the reference tree (the
GPU box).  TEST INFRASTRUCTURE.

* SchedNode        stand-in node (workload.refmodel) + the three mutators AttemptScheduling calls after a match,
                    restated on top of the oracle's commit step (oracle/nhd_oracle.py, pinned to the reference)
* attempt_scheduling   nhd/NHDScheduler.py:249-353 without the K8s side: InitialNodeFilter -> matcher.FindNode -> SetBusy ->
                    SetPhysicalIdsFromMapping -> ClaimPodNICResources -> "bind"
* check_pending_pods   the loop of CheckPendingPods (:425-437) over a pending list, and its batched form: ONE
                    matcher.ScheduleBatch call decides and commits the whole list on the device, the bookkeeping on the
                    node objects follows.
What a replay must reproduce comes from the unmodified reference (tests/golden/sched, oracle/gen_golden_sched.py)."""
from typing import Dict, List, Optional, Sequence

from oracle import nhd_oracle as O
from workload import refmodel


class SchedNode(refmodel.StandInNode):
    def SetBusy(self):                                               # nhd/Node.py:843-845 under the test's virtual clock
        self.busy_time = self._clock()

    def SetPhysicalIdsFromMapping(self, mapping, top):               # nhd/Node.py:663-841
        ids: dict = {}
        bt = self.busy_time
        self._claim = O.commit(self, top, mapping, bt, ids, claim_nics=False)
        for pg, g in zip(top.proc_groups, ids["groups"]):            # what the reference writes into the pod's topology
            batch = list(g["cores"])
            for gpu, pos in zip(pg.group_gpus, g["gpus"]):
                gpu.device_id = self.gpus[pos].device_id
                for c in gpu.cpu_cores:
                    c.core = batch.pop(0)
            for c in pg.proc_cores:
                c.core = batch.pop(0)
            for c, k in zip(pg.misc_cores, g["helpers"]):
                c.core = k
        for c, k in zip(top.misc_cores, ids["misc"]):
            c.core = k
        for pair in top.nic_core_pairing:                            # NICGroup.AddInterface(mac), nhd/Node.py:758-764
            gi = next(i for i, pg in enumerate(top.proc_groups) if pair.rx_core in pg.proc_cores)
            numa, idx = mapping["nic"][gi]
            pair.mac = next(n.mac for n in self.nics if n.numa_node == numa and n.idx == idx)
        return [(i, 0, 0) for i in self._claim]

    def ClaimPodNICResources(self, nidx):                            # nhd/Node.py:644-646
        for i in nidx:
            self.nics[i].pods_used += 1

    # release / reclaim / reset and the scalar setters (row f2): restated for the stand-in objects, pinned to the
    # reference by tests/test_delta_core.py::test_standin_mutators_match_reference
    def _topology_cores(self, top):
        for pg in top.proc_groups:
            yield from (c.core for c in pg.misc_cores)
            yield from (c.core for c in pg.proc_cores)
            for g in pg.group_gpus:
                yield from (c.core for c in g.cpu_cores)
        yield from (c.core for c in top.misc_cores)

    def _set_topology(self, top, used: bool):
        for c in self._topology_cores(top):
            self.cores[c].used = used
        for pg in top.proc_groups:
            for g in pg.group_gpus:
                dev = next((d for d in self.gpus if d.device_id == g.device_id), None)
                if dev is not None:
                    dev.used = used
        for p in top.nic_core_pairing:
            nic = next((n for n in self.nics if n.mac == p.mac), None)
            if nic is None:
                continue
            sign = 1 if used else -1
            nic.speed_used[0] += sign * p.rx_core.nic_speed
            nic.speed_used[1] += sign * p.tx_core.nic_speed
            nic.pods_used += sign
        if top.hugepages_gb > 0:
            self.mem.free_hugepages_gb += -top.hugepages_gb if used else top.hugepages_gb

    def RemoveResourcesFromTopology(self, top):                      # nhd/Node.py:530-585
        self._set_topology(top, True)
        return True

    def AddResourcesFromTopology(self, top):                         # nhd/Node.py:587-636
        self._set_topology(top, False)

    def ResetResources(self):                                        # nhd/Node.py:144-161
        for c in self.cores:
            if c.core not in self.reserved_cores:
                c.used = False
        for g in self.gpus:
            g.used = False
        for n in self.nics:
            n.pods_used = 0
            n.speed_used = [0, 0]
        self.mem.free_hugepages_gb = self.mem.ttl_hugepages_gb

    def SetGroups(self, groups: str):                                # nhd/Node.py:308-310
        self.groups = groups.split('.')

    def SetHugepages(self, alloc: int, free: int):                   # nhd/Node.py:489-493
        self.mem.ttl_hugepages_gb = alloc
        self.mem.free_hugepages_gb = free - self.mem.res_hugepages_gb
        return True


def adopt(nodes: Dict[str, object], clock) -> Dict[str, object]:
    for n in nodes.values():
        n.__class__ = SchedNode
        n._clock = clock
    return nodes


def attempt_scheduling(nodes, matcher, top, pod_groups: Sequence[str], match=None) -> Optional[str]:
    """Returns the bound node or None (FailedScheduling).  `match`: a decision taken earlier (batched form)."""
    if match is None:
        filt = O.initial_node_filter(nodes, pod_groups)              # NHDScheduler.py:274
        match = matcher.FindNode(filt, top)                          # :277
    if match[0] is None:
        return None
    node = nodes[match[0]]
    node.SetBusy()                                                   # :289
    nic_list = node.SetPhysicalIdsFromMapping(match[1], top)         # :292
    node.ClaimPodNICResources(list({x[0] for x in nic_list}))        # :302-304
    return match[0]


def check_pending_pods(nodes, matcher, tops, groups, tick=None) -> List[Optional[str]]:
    out = []
    for top, grp in zip(tops, groups):
        if tick:
            tick()
        out.append(attempt_scheduling(nodes, matcher, top, grp))
    return out


def check_pending_pods_batched(nodes, matcher, tops, groups, now) -> List[Optional[str]]:
    """One device pass decides and commits the whole pending list (ScheduleBatch, apply=True); the node objects are then
    brought along with the reference's own mutators, which find the device mirror already current."""
    matches = matcher.ScheduleBatch(nodes, tops, pod_groups=groups, now=now, apply=True)
    return [attempt_scheduling(nodes, matcher, top, grp, match=m) for top, grp, m in zip(tops, groups, matches)]
