#!/usr/bin/env python3
"""Golden fixtures for the wire digest: config texts and what the UNMODIFIED reference
parser (nhd/TriadCfgParser.py on the oracle/_shim stand-ins) plus Packer.digest make of them.

    python oracle/gen_golden_wire.py        # build container only (needs /root/reference)

Writes tests/golden/wire/wire_configs.json; tests/test_wire_golden.py replays it against libnhdfit.so anywhere."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from nhd_amd import pack# noqa: E402
from tests import wire_gen                            # noqa: E402
from tests.test_wire_digest import reference_outcome  # noqa: E402

FIELDS = ["n_groups", "map_type", "hugepages_gb", "gpus", "cpu_smt", "cpu_nosmt", "misc_smt", "misc_nosmt", "n_proc",
          "rx", "tx", "n_help", "n_misc", "smt_bits", "misc_smt_enabled", "nic_use"]


def main():
    cases = []
    for seed in range(48):
        cases.append(("random%02d" % seed, wire_gen.make_config(seed)))
    for k, defect in enumerate(d for d in wire_gen.DEFECTS if d):
        for seed in (0, 1):
            cases.append((f"{defect}{seed}", wire_gen.make_config(20_000 + 7 * k + seed, defect)))
    out = []
    for tag, text in cases:
        res = reference_outcome(text)
        entry = {"tag": tag, "text": text, "outcome": res[0]}
        if res[0] == "ok":
            entry["req"] = {f: res[1][f].tolist() for f in FIELDS}
            entry["req_hex"] = res[1].tobytes().hex()
        out.append(entry)
    path = os.path.join(ROOT, "tests", "golden", "wire", "wire_configs.json")
    with open(path, "w") as f:
        json.dump({"generator": "oracle/gen_golden_wire.py", "dtype": str(pack.REQ), "cases": out}, f, indent=0)
    print(path, len(out), {o: sum(1 for c in out if c["outcome"] == o) for o in ("ok", "none", "raise", "limit")})


if __name__ == "__main__":
    main()
