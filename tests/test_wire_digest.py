"""Row f3 (request digest straight from the wire format): nhdfit_digest_triad_config (host C++ in libnhdfit.so)
against the UNMODIFIED reference parser nhd/TriadCfgParser.py (run on the libconf / magicattr stand-ins of
oracle/_shim, the two third-party packages being absent) followed by Packer.digest - the path the scheduler takes
today (nhd/NHDScheduler.py:262-277).  Needs the reference tree (build container only); the committed fixtures of
tests/golden/wire/wire_configs.json carry the same check to the GPU box (tests/test_wire_golden.py)."""
import numpy as np
import pytest

from nhd_amd import pack, wire
from oracle import ref_loader
from tests import wire_gen

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="reference tree not present")


def reference_outcome(text):
    """('ok', req) | ('none',) | ('raise',) | ('limit',) - what the scheduler would see with the reference."""
    try:
        top = ref_loader.config_to_topology(text)
    except Exception:  # noqa: BLE001 - any exception kills the scheduler thread the same way
        return ("raise",)
    if top is None:
        return ("none",)
    try:
        req = pack.Packer().digest(top)
    except pack.UnsupportedNode:
        return ("limit",)
    except Exception:  # noqa: BLE001 - e.g. a string NIC speed: FindNode's getters raise on it
        return ("raise",)
    return ("ok", req)


def product_outcome(text):
    try:
        req = wire.digest_config(text)
    except wire.ConfigError:
        return ("raise",)
    except pack.UnsupportedNode:
        return ("limit",)
    return ("none",) if req is None else ("ok", req)


def assert_same(text, tag):
    want, got = reference_outcome(text), product_outcome(text)
    assert want[0] == got[0], (tag, want[0], got[0], text)
    if want[0] == "ok":
        assert want[1].tobytes() == got[1].tobytes(), (tag, want[1], got[1], text)
    return want[0]


def test_random_well_formed_configs():
    seen = set()
    for seed in range(1500):
        seen.add(assert_same(wire_gen.make_config(seed), seed))
    assert "ok" in seen


@pytest.mark.parametrize("defect", [d for d in wire_gen.DEFECTS if d])
def test_defective_configs(defect):
    outcomes = set()
    for seed in range(120):
        outcomes.add(assert_same(wire_gen.make_config(10_000 + seed, defect), (defect, seed)))
    if defect not in ("helper_missing", "int_of_string", "speed_index", "two_numa_dp", "nic_cores_len", "helper_smt_missing"):
        assert outcomes <= {"none", "raise"}, (defect, outcomes)      # these defects always bite


def test_hand_written_edge_cases():
    base = wire_gen.make_config(3)
    cases = {
        "empty": "",
        "only comment": "# nothing\n",
        "scalar topology": 'TopologyCfg = 5; Hugepages_GB = 1;',
        "unterminated": 'TopologyCfg = { cpu_arch = "ANY";',
        "string speed": base.replace("rx_speeds", "rx_speeds_old", 1) + 'Extra = 1;',
        "duplicate setting": base + "\nHugepages_GB = 7;\n",
        "huge pages float": base.replace("Hugepages_GB", "Hugepages_GB_x", 1) + "\nHugepages_GB = 3.9;\n",
        "hugepages string": base.replace("Hugepages_GB", "Hugepages_GB_x", 1) + '\nHugepages_GB = " 12 ";\n',
    }
    for tag, text in cases.items():
        assert_same(text, tag)


def test_pod_groups_annotation():
    pk = pack.Packer()
    text = next(t for t in (wire_gen.make_config(s) for s in range(50)) if product_outcome(t)[0] == "ok")
    req = wire.digest_config(text, pod_groups=["default", "edge"], packer=pk)
    top = ref_loader.config_to_topology(text)
    assert req.tobytes() == pk.digest(top, ["default", "edge"]).tobytes()


def test_matcher_from_config_texts_equals_matcher_from_topologies():
    """HipMatcher.FindNodesFromConfigs(texts) == HipMatcher.FindNodes(reference-parsed topologies) == the reference
    Matcher on those topologies; texts the reference rejects give (None,).  Host build of the kernel arithmetic."""
    from nhd_amd.matcher import HipMatcher
    from oracle import nhd_oracle as O
    from tests import harness, util
    nl = util.random_cluster(31, 80)
    texts, tops = [], []
    for seed in range(400):
        t = wire_gen.make_config(seed, "no_hugepages" if seed % 9 == 0 else None)
        if reference_outcome(t)[0] in ("ok", "none"):
            texts.append(t)
            tops.append(ref_loader.config_to_topology(t))
        if len(texts) == 60:
            break
    assert any(t is None for t in tops) and sum(t is not None for t in tops) > 30
    m = HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine)
    got = m.FindNodesFromConfigs(nl, texts)
    live = [t for t in tops if t is not None]
    want_live = iter(m.FindNodes(nl, live))
    want = [(None,) if t is None else next(want_live) for t in tops]
    assert got == want
    placed = 0
    for t, g in zip(tops, got):
        if t is not None and len(t.proc_groups):
            ref = O.find_node(nl, t, util.CLOCK)
            assert (g[0], g[1:] and g[1]["gpu"]) == (ref[0], ref[1:] and tuple(ref[1]["gpu"]))
            placed += g[0] is not None
    assert placed > 0


def test_mutated_configs_agree():
    """Fuzz: 1-3 random character edits (delete / insert / replace, biased to libconfig punctuation) of generated
    configs - outcome (request bytes, None, raise, limit) must stay identical to the reference parser's."""
    rng = np.random.default_rng(20260921)
    alphabet = list('{}[]()=:;,."\\\\#/* -+eExL0123456789abtrue\\n')
    outcomes = set()
    for _ in range(1200):
        t = list(wire_gen.make_config(int(rng.integers(0, 500))))
        for _e in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(0, len(t)))
            op = rng.random()
            if op < 0.4:
                del t[pos]
            elif op < 0.8:
                t.insert(pos, alphabet[int(rng.integers(0, len(alphabet)))])
            else:
                t[pos] = alphabet[int(rng.integers(0, len(alphabet)))]
        outcomes.add(assert_same("".join(t), "fuzz"))
    assert outcomes == {"ok", "none", "raise", "limit"}


def test_sequential_batch_from_config_texts():
    """Mode B (commit after every pod) fed with config texts == mode B fed with the reference-parsed topologies."""
    from nhd_amd.matcher import HipMatcher
    from tests import harness, util
    nl = util.random_cluster(44, 50)
    texts = [t for t in (wire_gen.make_config(s) for s in range(200)) if reference_outcome(t)[0] == "ok"][:40]
    tops = [ref_loader.config_to_topology(t) for t in texts]
    m = HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine)
    assert m.FindNodesFromConfigs(nl, texts, sequential=True) == m.ScheduleBatch(nl, tops)


def test_configs_with_more_groups_than_the_table_pass_holds():
    """Texts with 5..8 processing groups: digest_config turns them away (WIRE_LIMIT), digest_config_big gives the bytes
    Packer.digest_big gives for the reference parser's CfgTopology - the record the general path answers (nhdfit_big_find)."""
    seen = {}
    for seed in range(700):
        text = wire_gen.make_config(50_000 + seed, types_hi=5, inst_hi=3)
        try:
            top = ref_loader.config_to_topology(text)
        except Exception:  # noqa: BLE001
            continue
        if top is None or len(top.proc_groups) <= pack.MAX_GROUPS:
            continue
        G = len(top.proc_groups)
        try:
            want = pack.Packer().digest_big(top)
        except Exception:  # noqa: BLE001 - e.g. a string NIC speed
            continue
        with pytest.raises(pack.UnsupportedNode):
            wire.digest_config(text)
        got = wire.digest_config_big(text)
        assert got is not None and got.tobytes() == want.tobytes(), (seed, G, want, got)
        seen[G] = seen.get(G, 0) + 1
    assert set(seen) >= {5, 6, 8}, seen


def test_matcher_from_config_texts_with_big_pods():
    """FindNodesFromConfigs on texts of which some have 5..8 processing groups: equal to FindNodes / ScheduleBatch on the
    reference-parsed topologies (the big ones through nhdfit_big_req both ways), and to the oracle.  Host build."""
    from nhd_amd.matcher import HipMatcher
    from oracle import nhd_oracle as O
    from tests import harness, util
    descs = util.random_cluster_desc(52, 40, occupancy=0.05)
    nl = util.build_cluster(descs)
    texts, tops = [], []
    for seed in range(900):
        t = wire_gen.make_config(50_000 + seed, types_hi=5, inst_hi=3)
        try:
            top = ref_loader.config_to_topology(t)
            if top is None or not len(top.proc_groups):
                continue
            (pack.Packer().digest_big if len(top.proc_groups) > 4 else pack.Packer().digest)(top)
        except Exception:  # noqa: BLE001
            continue
        if len(top.proc_groups) > 4 or len(texts) % 3 == 0:
            texts.append(t)
            tops.append(top)
        if len(texts) == 36:
            break
    assert sum(len(t.proc_groups) > 4 for t in tops) >= 15
    m = HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine)
    got = m.FindNodesFromConfigs(nl, texts)
    assert got == m.FindNodes(nl, tops)
    placed_big = 0
    for t, g in zip(tops, got):
        ref = O.find_node(nl, t, util.CLOCK)
        assert (g[0], g[1:] and g[1]["gpu"]) == (ref[0], ref[1:] and tuple(ref[1]["gpu"]))
        placed_big += g[0] is not None and len(t.proc_groups) > 4
    assert placed_big >= 1
    assert m.FindNodesFromConfigs(nl, texts, sequential=True) == m.ScheduleBatch(nl, tops)
    # a text asking for more hugepages than the pod tile's table holds digests without a code: it must ride the general path like
    # the same pod through FindNodes(tops), not fail the whole call
    import re
    k = next(i for i, t in enumerate(tops) if len(t.proc_groups) <= 4 and re.search(r"Hugepages_GB\s*=\s*[\d.]+L?;", texts[i]))
    heavy = re.sub(r"(Hugepages_GB\s*=\s*)[\d.]+L?;", r"\g<1>2000;", texts[k], count=1)
    heavy_top = ref_loader.config_to_topology(heavy)
    assert int(heavy_top.hugepages_gb) == 2000
    mixed_texts, mixed_tops = [texts[0], heavy, texts[1]], [tops[0], heavy_top, tops[1]]
    assert m.FindNodesFromConfigs(nl, mixed_texts) == m.FindNodes(nl, mixed_tops)


def test_mutated_big_configs_agree():
    """The fuzz of test_mutated_configs_agree on texts with up to eight processing groups, through the big record: outcome (request
    bytes, None, raise, limit) identical to the reference parser's followed by Packer.digest_big."""
    def ref_outcome(text):
        try:
            top = ref_loader.config_to_topology(text)
        except Exception:  # noqa: BLE001
            return ("raise",)
        if top is None:
            return ("none",)
        try:
            return ("ok", pack.Packer().digest_big(top))
        except pack.UnsupportedNode:
            return ("limit",)
        except Exception:  # noqa: BLE001
            return ("raise",)

    def big_outcome(text):
        try:
            req = wire.digest_config_big(text)
        except wire.ConfigError:
            return ("raise",)
        except pack.UnsupportedNode:
            return ("limit",)
        return ("none",) if req is None else ("ok", req)

    rng = np.random.default_rng(20260922)
    alphabet = list('{}[]()=:;,."\\\\#/* -+eExL0123456789abtrue\\n')
    outcomes = set()
    for _ in range(500):
        t = list(wire_gen.make_config(50_000 + int(rng.integers(0, 300)), types_hi=6, inst_hi=3))
        for _e in range(int(rng.integers(0, 3))):
            pos = int(rng.integers(0, len(t)))
            op = rng.random()
            if op < 0.4:
                del t[pos]
            elif op < 0.8:
                t.insert(pos, alphabet[int(rng.integers(0, len(alphabet)))])
            else:
                t[pos] = alphabet[int(rng.integers(0, len(alphabet)))]
        text = "".join(t)
        want, got = ref_outcome(text), big_outcome(text)
        assert want[0] == got[0], (want[0], got[0], text)
        if want[0] == "ok":
            assert want[1].tobytes() == got[1].tobytes(), text
        outcomes.add(want[0])
    assert outcomes >= {"ok", "none", "raise", "limit"}, outcomes


def test_config_texts_under_enable_sharing(monkeypatch):
    """nhd/Node.py:20 ENABLE_SHARING = True and pods digested from their texts: the digest marks a pod whose group has several RX / TX
    core pairs (NHDFIT_RF_NIC_SPLIT, with NHDFIT_RF_NIC_SPLIT_DYADIC when every such speed is a multiple of 2^-20) as Packer.digest does,
    and the matcher admits it on the same condition - mode A and mode B from texts equal the same calls on the parsed topologies; with a
    speed that is not dyadic in the mirror, such a pod is answered (None,) on both paths (the wire path had no such check)."""
    from nhd_amd.matcher import HipMatcher
    from oracle import nhd_oracle as O
    from tests import harness, util
    from workload import refmodel
    monkeypatch.setattr(refmodel, "ENABLE_SHARING", True)
    monkeypatch.setattr(O, "ENABLE_SHARING", True)
    nl = util.random_cluster(45, 24)
    texts = [t for t in (wire_gen.make_config(s) for s in range(300)) if reference_outcome(t)[0] == "ok"][:40]
    tops = [ref_loader.config_to_topology(t) for t in texts]
    reqs, codes = wire.digest_configs(texts)
    split = [bool(int(r["flags"]) & pack.RF_NIC_SPLIT) for r in reqs]
    assert any(split) and not all(split)
    for r, t in zip(reqs, tops):
        assert int(r["flags"]) == int(pack.Packer().digest(t)["flags"])
    new = lambda: HipMatcher(clock=lambda: util.CLOCK, engine_factory=harness.HarnessEngine)      # noqa: E731
    a, b = new().FindNodesFromConfigs(nl, texts), new().FindNodes(nl, tops)
    assert a == b and sum(r[0] is not None for r in a) >= 5
    assert [O.find_node(nl, t, util.CLOCK)[0] for t in tops] == [r[0] for r in a]
    assert new().FindNodesFromConfigs(nl, texts, sequential=True) == new().ScheduleBatch(nl, tops)
    # a speed of 0.1 Gb/s among the nodes' own speed_used: sums are no longer exact in any order - the pods with split groups are out
    next(iter(nl.values())).nics[0].speed_used[0] = 0.1
    a, b = new().FindNodesFromConfigs(nl, texts), new().FindNodes(nl, tops)
    assert a == b
    assert all(r == (None,) for r, s in zip(a, split) if s) and any(r != (None,) for r in a)
