"""Wire digest (nhdfit_digest_triad_config, host C++ in libnhdfit.so) against the committed fixtures that
oracle/gen_golden_wire.py produced with the unmodified reference parser.  Runs anywhere the library loads
(no GPU, no reference tree)."""
import json
import os

import numpy as np
import pytest

from nhd_amd import pack, wire

PATH = os.path.join(os.path.dirname(__file__), "golden", "wire", "wire_configs.json")
with open(PATH) as f:
    CASES = json.load(f)["cases"]


@pytest.mark.parametrize("case", CASES, ids=[c["tag"] for c in CASES])
def test_wire_golden(case):
    try:
        req = wire.digest_config(case["text"])
        got = "none" if req is None else "ok"
    except wire.ConfigError:
        got, req = "raise", None
    except pack.UnsupportedNode:
        got, req = "limit", None
    assert got == case["outcome"]
    if got == "ok":
        assert req.tobytes().hex() == case["req_hex"]
        for field, want in case["req"].items():
            assert np.array_equal(np.asarray(req[field]), np.asarray(want)), field


def test_fixture_covers_every_outcome():
    assert {c["outcome"] for c in CASES} == {"ok", "none", "raise", "limit"}
