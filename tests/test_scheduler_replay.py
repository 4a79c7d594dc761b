"""Row f4 without the reference tree: the scheduler-loop fixtures the unmodified reference produced (tests/golden/sched)
replayed through HipMatcher in attached mode on the host-twin engine.  The same replay runs on the GPU in
tests/test_gpu_parity.py."""
import os

import pytest

from tests import harness, sched_check, sched_standin
from workload import refmodel, synth


@pytest.mark.parametrize("path", sched_check.FIXTURES, ids=[os.path.basename(p)[:-5] for p in sched_check.FIXTURES])
def test_attempt_scheduling_replay(path):
    assert sched_check.check_per_pod(sched_check.load(path), engine_factory=harness.HarnessEngine) >= 10


@pytest.mark.parametrize("cfg,n,P", [(3, 40, 100), (4, 32, 120), (5, 64, 160)])
def test_pending_list_batched_equals_pod_by_pod(cfg, n, P):
    """CheckPendingPods over a pending list (nhd/NHDScheduler.py:425-437): ONE ScheduleBatch(apply=True) call + the
    bookkeeping mutators must leave binds, node objects and the mirror exactly where the pod-by-pod loop leaves them
    (same instant for every pod: the batch is matched at one clock reading)."""
    case = {"config": cfg, "n_nodes": n, "n_pods": P, "clock0": synth.make_cluster(cfg, n_nodes=n).clock_now, "dt": 0.0}
    nodes_a, m_a, binds_a, up_a = sched_check.replay(case, batched=False, engine_factory=harness.HarnessEngine)
    nodes_b, m_b, binds_b, up_b = sched_check.replay(case, batched=True, engine_factory=harness.HarnessEngine)
    assert binds_a == binds_b and sum(b is not None for b in binds_a) >= 10
    assert sched_check.packed(nodes_a) == sched_check.packed(nodes_b)
    assert sched_check.mirror_state(m_a) == sched_check.mirror_state(m_b) == sched_check.packed(nodes_b)
    assert up_a == 0 and up_b == 0
