// winner_map.h - the resource mapping FindNode returns for the selected node
// (nhd/Matcher.py:423-452 GetNumaGroupIdx over the lists left by 337-391).
//
// The reference keeps candidate NUMA assignments in Python sets and converts them with list(set),
// so WHICH assignment it returns depends on CPython's set iteration order for tuples of small
// ints (hard part 1).  To return the identical mapping without running
// Python per pod, the relevant slice of CPython 3.8-3.12 is modelled here bit for bit:
//   * tuple hashing     Objects/tupleobject.c  tuplehash()   (xxHash-style, 64-bit build)
//   * set insertion     Objects/setobject.c    set_add_entry(), LINEAR_PROBES 9, PERTURB_SHIFT 5
//   * set growth        set_table_resize() / set_insert_clean()  (fill*5 >= mask*3 -> used*4)
//   * set & set         set_intersection(): iterate the smaller operand (the right one on ties)
// tests/test_pyset_emulation.py checks this model against the running interpreter exhaustively
// for every set the path can build.  Host- and device-compilable, no allocation.
#pragma once
#include "fit_core.h"

namespace nhdfit {

// ---- CPython model --------------------------------------------------------------------------------
constexpr uint64_t kXX1 = 11400714785074694791ULL;
constexpr uint64_t kXX2 = 14029467366897019727ULL;
constexpr uint64_t kXX5 = 2870177450012600261ULL;

// hash(tuple of `len` ints); hash(int k>=0) == k.  Tuples are coded in base U in {1, 2} (NUMA nodes per
// node) with the first element most significant: digit i = bit (len-1-i) of `code` (all zero for U=1),
// which keeps integer division out of the device code.
NHD_HD uint64_t py_tuple_hash(uint32_t code, int len, int base) {
    (void)base;
    uint64_t acc = kXX5;
    for (int i = 0; i < len; ++i) {
        const uint64_t lane = (code >> (len - 1 - i)) & 1u;
        acc += lane * kXX2;
        acc = (acc << 31) | (acc >> 33);
        acc *= kXX1;
    }
    acc += (uint64_t)len ^ (kXX5 ^ 3527539ULL);
    if (acc == (uint64_t)-1) return 1546275796ULL;
    return acc;
}

constexpr int kSetCap = 128;      // 32 distinct tuples at most (2^(G+1), G=4) -> table never exceeds 128 slots

struct PySet {
    int mask, fill;
    int16_t key[kSetCap];         // -1 = unused slot
    uint64_t hash[kSetCap];
};

NHD_HD void ps_init(PySet& s) {
    s.mask = 7;
    s.fill = 0;
    for (int i = 0; i < kSetCap; ++i) s.key[i] = -1;
}

NHD_HD void ps_insert_clean(int16_t* key, uint64_t* hash, int mask, int16_t k, uint64_t h) {
    uint64_t perturb = h;
    uint64_t i = h & (uint64_t)mask;
    for (;;) {
        if (key[i] < 0) break;
        bool found = false;
        if (i + 9 <= (uint64_t)mask) {
            for (int j = 1; j <= 9; ++j)
                if (key[i + j] < 0) { i += j; found = true; break; }
        }
        if (found) break;
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & (uint64_t)mask;
    }
    key[i] = k;
    hash[i] = h;
}

NHD_HD void ps_resize(PySet& s, int minused) {
    int newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    int16_t ok[kSetCap];
    uint64_t oh[kSetCap];
    const int oldmask = s.mask;
    for (int i = 0; i <= oldmask; ++i) { ok[i] = s.key[i]; oh[i] = s.hash[i]; }
    for (int i = 0; i < kSetCap; ++i) s.key[i] = -1;
    s.mask = newsize - 1;
    for (int i = 0; i <= oldmask; ++i)
        if (ok[i] >= 0) ps_insert_clean(s.key, s.hash, s.mask, ok[i], oh[i]);
}

// returns slot of key or -1
NHD_HD int ps_find(const PySet& s, int16_t k, uint64_t h) {
    uint64_t perturb = h;
    uint64_t i = h & (uint64_t)s.mask;
    for (;;) {
        const int probes = (i + 9 <= (uint64_t)s.mask) ? 9 : 0;
        for (int j = 0; j <= probes; ++j) {
            if (s.key[i + j] < 0) return -1;
            if (s.hash[i + j] == h && s.key[i + j] == k) return (int)(i + j);
        }
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & (uint64_t)s.mask;
    }
}

NHD_HD void ps_add(PySet& s, int16_t k, uint64_t h) {
    uint64_t perturb = h;
    uint64_t i = h & (uint64_t)s.mask;
    for (;;) {
        const int probes = (i + 9 <= (uint64_t)s.mask) ? 9 : 0;
        for (int j = 0; j <= probes; ++j) {
            if (s.key[i + j] < 0) {
                s.key[i + j] = k;
                s.hash[i + j] = h;
                s.fill++;
                if (s.fill * 5 >= s.mask * 3) ps_resize(s, s.fill * 4);
                return;
            }
            if (s.hash[i + j] == h && s.key[i + j] == k) return;    // already present
        }
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & (uint64_t)s.mask;
    }
}

// result = a & b   (set_intersection: iterate the smaller, b on ties; probe the other)
NHD_HD void ps_intersect(const PySet& a, const PySet& b, PySet& out) {
    ps_init(out);
    const PySet* probe = &a;
    const PySet* iter = &b;
    if (b.fill > a.fill) { probe = &b; iter = &a; }
    for (int i = 0; i <= iter->mask; ++i)
        if (iter->key[i] >= 0 && ps_find(*probe, iter->key[i], iter->hash[i]) >= 0)
            ps_add(out, iter->key[i], iter->hash[i]);
}

// list(set): keys in slot order
NHD_HD int ps_list(const PySet& s, int16_t* out) {
    int n = 0;
    for (int i = 0; i <= s.mask; ++i)
        if (s.key[i] >= 0) out[n++] = s.key[i];
    return n;
}


// ---- register-resident variant for tables that never outgrow 32 slots (<= 18 keys) -----------------
// Same insertion / growth / intersection rules as PySet above.  Keys live in four 64-bit registers
// (8 bits per slot) and occupancy in a 32-bit mask, so the winner-mapping kernel does not touch
// scratch memory for pods with G <= 3 (at most 2^(G+1) = 16 distinct tuples).  Hashes are recomputed
// from the key (hash(tuple) is a pure function of its digits).
struct SmallSet {
    uint32_t used;
    int mask, fill, len, base;
    uint64_t k0, k1, k2, k3;
};

NHD_HD void ss_init(SmallSet& s, int len, int base) {
    s.used = 0; s.mask = 7; s.fill = 0; s.len = len; s.base = base;
    s.k0 = s.k1 = s.k2 = s.k3 = 0;
}
NHD_HD int ss_key(const SmallSet& s, int slot) {
    const uint64_t w = slot < 16 ? (slot < 8 ? s.k0 : s.k1) : (slot < 24 ? s.k2 : s.k3);
    return (int)((w >> ((slot & 7) * 8)) & 0xFF);
}
NHD_HD void ss_put(SmallSet& s, int slot, int key) {
    const uint64_t v = (uint64_t)key << ((slot & 7) * 8);
    if (slot < 8) s.k0 |= v; else if (slot < 16) s.k1 |= v; else if (slot < 24) s.k2 |= v; else s.k3 |= v;
    s.used |= 1u << slot;
}
// slot where `key` lives (>= 0), or -(free slot)-1 where it would be inserted
NHD_HD int ss_probe(const SmallSet& s, int key, uint64_t h, bool match) {
    uint64_t perturb = h;
    uint32_t i = (uint32_t)(h & (uint64_t)s.mask);
    for (;;) {
        const int probes = (i + 9 <= (uint32_t)s.mask) ? 9 : 0;
        for (int j = 0; j <= probes; ++j) {
            if (!(s.used >> (i + j) & 1)) return -(int)(i + j) - 1;
            if (match && ss_key(s, (int)(i + j)) == key) return (int)(i + j);
        }
        perturb >>= 5;
        i = (uint32_t)((i * 5 + 1 + perturb) & (uint64_t)s.mask);
    }
}
NHD_HD void ss_add(SmallSet& s, int key) {
    const uint64_t h = py_tuple_hash((uint32_t)key, s.len, s.base);
    const int r = ss_probe(s, key, h, true);
    if (r >= 0) return;
    ss_put(s, -r - 1, key);
    s.fill++;
    if (s.fill * 5 >= s.mask * 3) {                   // 8 -> 32 slots (set_table_resize(used*4)); never further
        SmallSet o = s;
        s.used = 0; s.mask = 31; s.k0 = s.k1 = s.k2 = s.k3 = 0;
        for (int i = 0; i <= o.mask; ++i)
            if (o.used >> i & 1) {
                const int k = ss_key(o, i);
                ss_put(s, -ss_probe(s, k, py_tuple_hash((uint32_t)k, s.len, s.base), false) - 1, k);
            }
    }
}
NHD_HD bool ss_has(const SmallSet& s, int key) {
    return ss_probe(s, key, py_tuple_hash((uint32_t)key, s.len, s.base), true) >= 0;
}
NHD_HD void ss_intersect(const SmallSet& a, const SmallSet& b, SmallSet& out) {
    ss_init(out, a.len, a.base);
    const bool swap = b.fill > a.fill;
    const SmallSet& probe = swap ? b : a;
    const SmallSet& iter = swap ? a : b;
    for (int i = 0; i <= iter.mask; ++i)
        if (iter.used >> i & 1) {
            const int k = ss_key(iter, i);
            if (ss_has(probe, k)) ss_add(out, k);
        }
}
NHD_HD int ss_list(const SmallSet& s, int16_t* out) {
    int n = 0;
    for (int i = 0; i <= s.mask; ++i)
        if (s.used >> i & 1) out[n++] = (int16_t)ss_key(s, i);
    return n;
}

// uniform front-ends so map_winner can be written once
struct GenericOps {
    typedef PySet Set;
    NHD_HD static void init(Set& s, int, int) { ps_init(s); }
    NHD_HD static void add(Set& s, int key, int len, int base) { ps_add(s, (int16_t)key, py_tuple_hash((uint32_t)key, len, base)); }
    NHD_HD static void isect(const Set& a, const Set& b, Set& o) { ps_intersect(a, b, o); }
    NHD_HD static int list(const Set& s, int16_t* o) { return ps_list(s, o); }
    NHD_HD static int size(const Set& s) { return s.fill; }
};
struct SmallOps {
    typedef SmallSet Set;
    NHD_HD static void init(Set& s, int len, int base) { ss_init(s, len, base); }
    NHD_HD static void add(Set& s, int key, int, int) { ss_add(s, key); }
    NHD_HD static void isect(const Set& a, const Set& b, Set& o) { ss_intersect(a, b, o); }
    NHD_HD static int list(const Set& s, int16_t* o) { return ss_list(s, o); }
    NHD_HD static int size(const Set& s) { return s.fill; }
};

// ---- the winner's resource state -------------------------------------------------------------------
struct WinnerState {
    int U;                        // Node.numa_nodes (1 or 2)
    bool smt;
    int free_c[2], free_g[2];
    nhdfit_detail d;
    const double* caps;           // capacity per class
};

// digit i of tuple code (length len, base U in {1,2}), i = 0 is the first element
NHD_HD int tup_digit(uint32_t code, int len, int U, int i) {
    (void)U;
    return (int)((code >> (len - 1 - i)) & 1u);
}

NHD_HD uint32_t ipow(int b, int e) { return b == 1 ? 1u : 1u << e; }   // b in {1, 2}

// First NIC choice (in the reference's enumeration order, Matcher.py:242-268) that hosts every group
// on the NUMA node `assign` gives it, or false.  Order: itertools.product over NUMA nodes of
// itertools.product(range(K_u), repeat=#groups on u) - i.e. an odometer whose most significant
// digits are the NUMA-0 groups (ascending group index), then the NUMA-1 groups.
NHD_HD bool first_nic_choice(const nhdfit_req& r, const WinnerState& w, uint32_t gcode, bool pci, int8_t nic_idx[kMaxG]) {
    const int G = (int)r.n_groups;
    int order[kMaxG], numa_of[kMaxG], n = 0;
    for (int u = 0; u < w.U; ++u)
        for (int g = 0; g < G; ++g)
            if (tup_digit(gcode, G, w.U, g) == u) { order[n] = g; numa_of[g] = u; ++n; }
    for (int g = 0; g < G; ++g)
        if (w.d.nic_cnt[numa_of[g]] == 0) return false;
    int pick[kMaxG] = {0, 0, 0, 0};
    for (;;) {
        // evaluate: subtract in group order per NIC
        bool ok = true;
        double rx[2][NHDFIT_MAX_NICS_PER_NUMA], tx[2][NHDFIT_MAX_NICS_PER_NUMA];
        for (int u = 0; u < w.U; ++u)
            for (int k = 0; k < w.d.nic_cnt[u]; ++k) rx[u][k] = tx[u][k] = w.caps[w.d.nic_cls[u][k]];
        for (int g = 0; g < G; ++g) {
            const int u = numa_of[g], k = pick[g];
            rx[u][k] = rx[u][k] - r.rx[g];
            tx[u][k] = tx[u][k] - r.tx[g];
        }
        for (int u = 0; u < w.U && ok; ++u)
            for (int k = 0; k < w.d.nic_cnt[u]; ++k)
                if (rx[u][k] < 0 || tx[u][k] < 0) { ok = false; break; }
        if (ok && pci) {                                  // Matcher.py:312-322
            uint8_t cnt[NHDFIT_MAX_SWITCHES] = {0};
            for (int g = 0; g < G; ++g) cnt[w.d.nic_sw[numa_of[g]][pick[g]]]++;
            for (int s = 0; s < NHDFIT_MAX_SWITCHES; ++s)
                if (cnt[s] > w.d.sw_free[s]) { ok = false; break; }
        }
        if (ok) {
            for (int g = 0; g < G; ++g) nic_idx[g] = (int8_t)pick[g];
            return true;
        }
        int pos = G - 1;                                  // advance the odometer (last digit fastest)
        while (pos >= 0) {
            const int g = order[pos];
            if (++pick[g] < w.d.nic_cnt[numa_of[g]]) break;
            pick[g] = 0;
            --pos;
        }
        if (pos < 0) return false;
    }
}

// NIC-feasible assignments as tuple codes, from the table bits (bit p: bit i of p = NUMA of group i;
// tuple code: first group most significant digit).
NHD_HD uint32_t nic_codes_from_table_bits(uint32_t bits, int G, int U) {
    if (U == 1) return bits & 1u;
    uint32_t out = 0;
    for (uint32_t p = 0; p < (1u << G); ++p) {
        if (!(bits >> p & 1)) continue;
        uint32_t code = 0;
        for (int i = 0; i < G; ++i) code |= (p >> i & 1u) << (G - 1 - i);
        out |= 1u << code;
    }
    return out;
}

// Restatement of the winner-only tail of FindNode (Matcher.py:337-391 + 423-452).
// `nic_codes`: bit c set = assignment with tuple code c has at least one valid NIC choice (after the
// PCI pruning) - taken from the same reach tables the fit kernel used.  Returns false if infeasible.
template <class Ops>
NHD_HD bool map_winner_t(const nhdfit_req& r, const WinnerState& w, uint32_t nic_codes, nhdfit_mapping& out) {
    typedef typename Ops::Set Set;
    const int G = (int)r.n_groups, U = w.U;
    const bool pci = r.map_type == NHDFIT_MAP_PCI;
    const uint32_t nG = ipow(U, G), nC = ipow(U, G + 1);
    out.valid = 0;

    // candidate sets in product order (Matcher.py:116-141, 206-220)
    Set sg, sc;
    Ops::init(sg, G, U);
    Ops::init(sc, G + 1, U);
    uint32_t demand[kMaxG + 1];
    for (int g = 0; g < G; ++g) demand[g] = w.smt ? r.cpu_smt[g] : r.cpu_nosmt[g];
    demand[G] = w.smt ? r.misc_smt : r.misc_nosmt;
    for (uint32_t code = 0; code < nG; ++code) {
        uint32_t tot[2] = {0, 0};
        for (int g = 0; g < G; ++g) tot[tup_digit(code, G, U, g)] += r.gpus[g];
        bool ok = true;
        for (int u = 0; u < U; ++u) ok = ok && tot[u] <= (uint32_t)w.free_g[u];
        if (ok) Ops::add(sg, (int)code, G, U);
    }
    for (uint32_t code = 0; code < nC; ++code) {
        uint32_t tot[2] = {0, 0};
        for (int g = 0; g <= G; ++g) tot[tup_digit(code, G + 1, U, g)] += demand[g];
        bool ok = true;
        for (int u = 0; u < U; ++u) ok = ok && tot[u] <= (uint32_t)w.free_c[u];
        if (ok) Ops::add(sc, (int)code, G + 1, U);
    }
    if (Ops::size(sg) == 0 || Ops::size(sc) == 0 || !(nic_codes & ((nG >= 32 ? 0u : (1u << nG)) - 1u))) return false;

    // intersection of the three prefix sets (Matcher.py:342-346): set(list) re-inserts in list order
    int16_t lg[1 << kMaxG], lc[2 << kMaxG];
    const int ng = Ops::list(sg, lg), nc = Ops::list(sc, lc);
    Set a, b, c, ab, abc;
    Ops::init(a, G, U); Ops::init(b, G, U); Ops::init(c, G, U);
    for (int i = 0; i < ng; ++i) Ops::add(a, lg[i], G, U);
    for (int i = 0; i < nc; ++i) Ops::add(b, lc[i] >> (U - 1), G, U);           // tuple[:-1]
    for (uint32_t code = 0; code < nG; ++code)
        if (nic_codes >> code & 1) Ops::add(c, (int)code, G, U);
    Ops::isect(a, b, ab);
    Ops::isect(ab, c, abc);
    if (Ops::size(abc) == 0) return false;

    // GPU list: replaced by the intersection only if that drops something (Matcher.py:363-366)
    int16_t gl[1 << kMaxG];
    int ngl;
    if (Ops::size(abc) < Ops::size(sg)) ngl = Ops::list(abc, gl);
    else { ngl = ng; for (int i = 0; i < ng; ++i) gl[i] = lg[i]; }

    // GetNumaGroupIdx (Matcher.py:427-437): first maximiser of max-min per-NUMA group count
    int best = -1, best_spread = -1;
    for (int i = 0; i < ngl; ++i) {
        int cnt[2] = {0, 0};
        for (int g = 0; g < G; ++g) cnt[tup_digit((uint32_t)gl[i], G, U, g)]++;
        int mx = cnt[0], mn = cnt[0];
        for (int u = 1; u < U; ++u) { mx = cnt[u] > mx ? cnt[u] : mx; mn = cnt[u] < mn ? cnt[u] : mn; }
        if (mx - mn > best_spread) { best_spread = mx - mn; best = gl[i]; }
    }
    const uint32_t gcode = (uint32_t)best;
    int16_t ccode = -1;                                                    // Matcher.py:441-444
    for (int i = 0; i < nc; ++i)
        if ((uint32_t)(lc[i] >> (U - 1)) == gcode) { ccode = lc[i]; break; }
    int8_t nic_idx[kMaxG];
    if (ccode < 0 || !first_nic_choice(r, w, gcode, pci, nic_idx)) return false;   // Matcher.py:446-449

    for (int g = 0; g < kMaxG; ++g) { out.gpu[g] = out.nic_numa[g] = out.nic_idx[g] = -1; }
    for (int g = 0; g <= kMaxG; ++g) out.cpu[g] = -1;
    for (int g = 0; g < G; ++g) {
        out.gpu[g] = (int8_t)tup_digit(gcode, G, U, g);
        out.nic_numa[g] = out.gpu[g];
        out.nic_idx[g] = nic_idx[g];
    }
    for (int g = 0; g <= G; ++g) out.cpu[g] = (int8_t)tup_digit((uint32_t)ccode, G + 1, U, g);
    out.valid = 1;
    return true;
}

// G <= 3: every set stays within 32 slots -> register-resident model; G == 4: generic model.
NHD_HD bool map_winner(const nhdfit_req& r, const WinnerState& w, uint32_t nic_codes, nhdfit_mapping& out) {
    if (r.n_groups <= 3) return map_winner_t<SmallOps>(r, w, nic_codes, out);
    return map_winner_t<GenericOps>(r, w, nic_codes, out);
}

// NIC-feasible assignment bits of one (pod, node) pair out of the pod's table column.
NHD_HD uint32_t nic_table_bits(const uint32_t* tab, uint32_t row_r, uint32_t col, bool pci, uint32_t sig0_numa,
                               uint32_t sig1_numa, uint32_t sig0_pci, uint32_t sig1_pci) {
    const uint32_t r0 = tab[(row_r + (pci ? sig0_pci : sig0_numa)) * kRowStride + col];
    const uint32_t r1 = tab[(row_r + (pci ? sig1_pci : sig1_numa)) * kRowStride + col];
    return (r0 >> 16) & r1 & 0xFFFFu;
}

}  // namespace nhdfit
