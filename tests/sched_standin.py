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
        return [(i, 0, 0) for i in self._claim]

    def ClaimPodNICResources(self, nidx):                            # nhd/Node.py:644-646
        for i in nidx:
            self.nics[i].pods_used += 1


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
