"""first_nic_choice (winner_map.h): depth-first with prefix pruning == the reference's plain enumeration order
(nhd/Matcher.py:242-268 product order, :267 the f64 test, :312-322 the PCI switch count).  CPU, host build of the header."""
import ctypes

import numpy as np
import pytest

from nhd_amd import pack
from tests import harness


def _case(rng, heavy):
    d = np.zeros(1, pack.DETAIL)
    U = 2 if rng.random() < 0.9 else 1
    d["numa_nodes"] = U
    nsw = int(rng.integers(1, 6))
    for u in range(U):
        k = int(rng.integers(0, 9 if not heavy else 17))
        d["nic_cnt"][0, u] = k
        d["nic_cls"][0, u, :k] = rng.integers(0, 4, k)
        d["nic_sw"][0, u, :k] = rng.integers(0, nsw, k)
    d["sw_free"][0, :nsw] = rng.integers(0, 4, nsw)
    caps = np.array([0.0, 22.5, 90.0, 50.400000000000006] + [0.0] * 12)
    if heavy:                                  # most NICs claimed: the plain walk visits n^G combinations
        m = rng.random((2, pack.MAX_NICS_PER_NUMA)) < 0.85
        d["nic_cls"][0][m] = 0
    r = np.zeros(1, pack.REQ)
    G = int(rng.integers(1, 5))
    r["n_groups"] = G
    speeds = [0.0, 10.0, 25.0, 40.0, 45.0, 50.4, 90.0, 22.5, 0.1]
    r["rx"][0, :G] = rng.choice(speeds, G)
    r["tx"][0, :G] = rng.choice(speeds, G)
    kind = rng.random()
    if kind < 0.05:
        r["rx"][0, int(rng.integers(0, G))] = -5.0            # not monotone: must fall back to the plain walk
    elif kind < 0.08:
        r["tx"][0, int(rng.integers(0, G))] = float("nan")
    gcode = int(rng.integers(0, 1 << G)) if U == 2 else 0
    return r, d, caps, gcode, int(rng.random() < 0.5)


@pytest.mark.parametrize("heavy", [False, True])
def test_pruned_walk_equals_plain_enumeration(heavy):
    L = harness.lib()
    rng = np.random.default_rng(0x51C + heavy)
    found = 0
    for _ in range(4000 if not heavy else 600):
        r, d, caps, gcode, pci = _case(rng, heavy)
        a = np.zeros(4, np.int8)
        b = np.zeros(4, np.int8)
        rc = L.hh_first_nic_choice(harness._p(r), harness._p(d), harness._p(caps), ctypes.c_uint32(gcode), pci, harness._p(a), harness._p(b))
        assert rc in (0, 3), (rc, r, d, gcode, pci)
        if rc:
            found += 1
            assert (a == b).all(), (a, b, r, d, gcode, pci)
    assert found > 50
