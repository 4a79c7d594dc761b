"""The CPython set/tuple-hash model in nhd_amd/csrc/winner_map.h against the running interpreter."""
import ctypes
import itertools

import numpy as np
import pytest

from tests import harness


def codes_of(tuples, base):
    out = []
    for t in tuples:
        c = 0
        for d in t:
            c = c * base + d
        out.append(c)
    return np.asarray(out, np.int16)


def tuples_of(codes, length, base):
    out = []
    for c in codes:
        c = int(c)
        t = []
        for _ in range(length):
            t.append(c % base)
            c //= base
        out.append(tuple(reversed(t)))
    return out


def model_list(tuples, length, base):
    L = harness.lib()
    codes = codes_of(tuples, base)
    out = np.zeros(64, np.int16)
    n = L.hh_set_list(codes.ctypes.data_as(ctypes.c_void_p), len(codes), length, base, out.ctypes.data_as(ctypes.c_void_p))
    return tuples_of(out[:n], length, base)


@pytest.mark.parametrize("length", [1, 2, 3, 4, 5])
def test_tuple_hash(length):
    L = harness.lib()
    for base in (1, 2):
        for t in itertools.product(range(base), repeat=length):
            assert L.hh_tuple_hash(int(codes_of([t], base)[0]), length, base) == hash(t) & (2 ** 64 - 1)


@pytest.mark.parametrize("length", [1, 2, 3])
def test_list_of_set_all_subsets_in_product_order(length):
    """Every subset of U^G inserted in product order (how Matcher.py:116-141 fills `stmp`)."""
    universe = list(itertools.product(range(2), repeat=length))
    for mask in range(1, 1 << len(universe)):
        sub = [t for i, t in enumerate(universe) if mask >> i & 1]
        s = set()
        for t in sub:
            s.add(t)
        assert model_list(sub, length, 2) == list(s)


@pytest.mark.parametrize("length", [4, 5])
def test_list_of_set_random_subsets(length):
    rng = np.random.default_rng(length)
    universe = list(itertools.product(range(2), repeat=length))
    for _ in range(3000):
        keep = rng.random(len(universe)) < rng.random()
        sub = [t for t, k in zip(universe, keep) if k]
        if rng.random() < 0.3:
            rng.shuffle(sub)
        s = set()
        for t in sub:
            s.add(t)
        assert model_list(sub, length, 2) == list(s)


@pytest.mark.parametrize("length", [1, 2, 3, 4])
def test_three_way_intersection_order(length):
    """list(set(a) & set(b) & set(c)) with a, b, c given as lists with duplicates (Matcher.py:342-346)."""
    L = harness.lib()
    rng = np.random.default_rng(100 + length)
    universe = list(itertools.product(range(2), repeat=length))
    for _ in range(4000):
        lists = []
        for _k in range(3):
            n = int(rng.integers(1, 2 * len(universe) + 1))
            lists.append([universe[int(i)] for i in rng.integers(0, len(universe), size=n)])
        want = list(set(lists[0]) & set(lists[1]) & set(lists[2]))
        arrs = [codes_of(x, 2) for x in lists]
        out = np.zeros(64, np.int16)
        n = L.hh_set_isect3(*(v for a in arrs for v in (a.ctypes.data_as(ctypes.c_void_p), len(a))), length, 2,
                            out.ctypes.data_as(ctypes.c_void_p))
        assert tuples_of(out[:n], length, 2) == want


def small_list(tuples, length, base):
    L = harness.lib()
    codes = codes_of(tuples, base)
    out = np.zeros(64, np.int16)
    n = L.hh_small_set_list(codes.ctypes.data_as(ctypes.c_void_p), len(codes), length, base, out.ctypes.data_as(ctypes.c_void_p))
    return tuples_of(out[:n], length, base)


@pytest.mark.parametrize("length", [1, 2, 3, 4])
def test_register_model_matches_python(length):
    """SmallSet (the register-resident model used for G <= 3, up to 16 keys) == CPython."""
    L = harness.lib()
    rng = np.random.default_rng(7 + length)
    universe = list(itertools.product(range(2), repeat=length))
    masks = range(1, 1 << len(universe)) if length <= 3 else [int(x) for x in rng.integers(1, 1 << 16, size=6000)]
    for mask in masks:
        sub = [t for i, t in enumerate(universe) if mask >> i & 1]
        if length == 4 and rng.random() < 0.3:
            rng.shuffle(sub)
        s = set()
        for t in sub:
            s.add(t)
        assert small_list(sub, length, 2) == list(s)
    for _ in range(3000):
        lists = []
        for _k in range(3):
            n = int(rng.integers(1, 2 * len(universe) + 1))
            lists.append([universe[int(i)] for i in rng.integers(0, len(universe), size=n)])
        want = list(set(lists[0]) & set(lists[1]) & set(lists[2]))
        arrs = [codes_of(x, 2) for x in lists]
        out = np.zeros(64, np.int16)
        n = L.hh_small_isect3(*(v for a in arrs for v in (a.ctypes.data_as(ctypes.c_void_p), len(a))), length, 2,
                              out.ctypes.data_as(ctypes.c_void_p))
        assert tuples_of(out[:n], length, 2) == want


@pytest.mark.parametrize("length", [1, 2, 3, 4])
def test_ascending_table_matches_python(length):
    """The static table of ascending-filled sets (AscEntry; what the choose role reads instead of inserting
    code by code) == list(set) of CPython for every subset (length 4: all 65 536 of them)."""
    L = harness.lib()
    universe = list(itertools.product(range(2), repeat=length))
    out = np.zeros(64, np.int16)
    for subset in range(1 << len(universe)):
        s = set()
        for i, t in enumerate(universe):
            if subset >> i & 1:
                s.add(t)
        n = L.hh_asc_set_list(subset, length, 2, out.ctypes.data_as(ctypes.c_void_p))
        assert tuples_of(out[:n], length, 2) == list(s), (length, subset)


def test_choose_tuples_table_vs_insertion_vs_generic():
    """choose_tuples: table-backed register model == insertion-built register model == generic PySet model."""
    L = harness.lib()
    rng = np.random.default_rng(99)
    g1, c1, g2, c2, g3, c3 = (ctypes.c_uint32(), ctypes.c_int(), ctypes.c_uint32(), ctypes.c_int(), ctypes.c_uint32(), ctypes.c_int())
    for _ in range(40000):
        G = int(rng.integers(1, 4)); U = int(rng.integers(1, 3))
        nG, nC = (1 << G, 1 << (G + 1)) if U == 2 else (1, 1)
        sg = int(rng.integers(1, 1 << nG)); sc = int(rng.integers(1, 1 << nC)); nic = int(rng.integers(1, 1 << nG))
        a = L.hh_choose(G, U, sg, sc, nic, 1, ctypes.byref(g1), ctypes.byref(c1))
        b = L.hh_choose(G, U, sg, sc, nic, 0, ctypes.byref(g2), ctypes.byref(c2))
        c = L.hh_choose_generic(G, U, sg, sc, nic, ctypes.byref(g3), ctypes.byref(c3))
        assert a == b == c, (G, U, sg, sc, nic)
        if a:
            assert (g1.value, c1.value) == (g2.value, c2.value) == (g3.value, c3.value), (G, U, sg, sc, nic)


def test_tabulated_choose_equals_the_model_everywhere():
    """The 256 + 65 536-entry table the shapes role answers from (U = 2, G <= 2) against the insertion-built
    register model and the generic PySet model, for EVERY input (index / encode / decode plumbing included)."""
    L = harness.lib()
    L.hh_choose_from_table.restype = ctypes.c_uint32
    g, c = ctypes.c_uint32(), ctypes.c_int()
    assert L.hh_choose_tabulated(1, 2) and L.hh_choose_tabulated(2, 2)
    assert not L.hh_choose_tabulated(3, 2) and not L.hh_choose_tabulated(1, 1) and not L.hh_choose_tabulated(4, 2)
    rng = np.random.default_rng(5)
    for G in (1, 2):
        nG, nC = 1 << G, 2 << G
        total = 1 << (2 * nG + nC)
        picks = range(total) if G == 1 else [int(x) for x in rng.integers(0, total, size=20000)]
        for x in picks:
            sg, sc, nic = x & ((1 << nG) - 1), (x >> nG) & ((1 << nC) - 1), x >> (nG + nC)
            word = L.hh_choose_from_table(G, sg, sc, nic)
            if sg and sc and nic:
                ok = L.hh_choose(G, 2, sg, sc, nic, 0, ctypes.byref(g), ctypes.byref(c))
                ok2 = L.hh_choose_generic(G, 2, sg, sc, nic, ctypes.byref(g), ctypes.byref(c)) if ok else 0
                assert ok == ok2
            else:
                ok = 0
            assert (word >> 8 & 1) == ok, (G, sg, sc, nic)
            if ok:
                assert ((word >> 4) & 7, word & 15) == (g.value, c.value), (G, sg, sc, nic)


def test_set_layout_state_machine_equals_the_model():
    """choose_tuples for three proc groups on two NUMA nodes through the set-layout state machine
    (nhd_amd/csrc/set_states.h: enumerated layouts + transition table) == the insertion-by-insertion register model
    == the generic PySet model, on 200 000 random inputs plus structured sweeps (all GPU / NIC subsets for sampled CPU
    subsets, all CPU subsets for sampled GPU / NIC subsets)."""
    L = harness.lib()
    L.hh_choose_g3.restype = ctypes.c_uint32
    L.hh_set_state_count.restype = ctypes.c_uint32
    assert 100 < L.hh_set_state_count() < 5000          # 338 layouts for keys 0..7
    g, c = ctypes.c_uint32(), ctypes.c_int()
    rng = np.random.default_rng(11)

    def check(sg, sc, nic, generic=False):
        word = L.hh_choose_g3(sg, sc, nic)
        ok = L.hh_choose(3, 2, sg, sc, nic, 0, ctypes.byref(g), ctypes.byref(c)) if (sg and sc and nic) else 0
        assert (word >> 8 & 1) == ok, (sg, sc, nic)
        if ok:
            assert ((word >> 4) & 7, word & 15) == (g.value, c.value), (sg, sc, nic)
            if generic:
                assert L.hh_choose_generic(3, 2, sg, sc, nic, ctypes.byref(g), ctypes.byref(c)) == 1
                assert ((word >> 4) & 7, word & 15) == (g.value, c.value)

    for k in range(200000):
        check(int(rng.integers(0, 256)), int(rng.integers(0, 65536)), int(rng.integers(0, 256)), generic=k % 50 == 0)
    for sc in [int(x) for x in rng.integers(1, 65536, size=4)] + [0xFFFF, 0x5555, 0x0F0F]:
        for sg in range(1, 256, 3):
            for nic in range(1, 256, 5):
                check(sg, sc, nic)
    for sg, nic in [(0xFF, 0xFF), (0x81, 0xFF), (0x7E, 0x7E), (0x18, 0xFF), (0xFF, 0x01)]:
        for sc in range(1, 65536, 7):
            check(sg, sc, nic)


def test_matcher_start_up_probes_agree_with_interpreter_and_model():
    """HipMatcher() checks at construction that the running interpreter orders sets the way the device model assumes
    (nhd_amd.matcher.check_interpreter_set_model): the literals it compares with must be what this interpreter AND the
    model produce."""
    import itertools
    from nhd_amd import matcher
    matcher.check_interpreter_set_model()
    L = harness.lib()
    for k, want in matcher._SET_ORDER_PROBES.items():
        codes = np.arange(1 << k, dtype=np.int16)                      # product order = ascending tuple code
        out = np.zeros(1 << k, np.int16)
        n = L.hh_set_list(codes.ctypes.data_as(ctypes.c_void_p), len(codes), k, 2, out.ctypes.data_as(ctypes.c_void_p))
        got = [tuple((int(c) >> (k - 1 - i)) & 1 for i in range(k)) for c in out[:n]]
        assert got == want == list(set(itertools.product(range(2), repeat=k)))
