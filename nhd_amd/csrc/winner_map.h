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
// node) with the first element most significant: digit i = bit (len-1-i) of `code` (all zero for U=1).
constexpr uint64_t py_tuple_hash_calc(uint32_t code, int len) {
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
// all hashes the path can need (tuples of length <= 5 over {0,1}), evaluated at compile time: the
// sequential mapping kernel spends most of its instructions here otherwise
struct TupleHashTable { uint64_t h[6][32]; };
constexpr TupleHashTable make_tuple_hash_table() {
    TupleHashTable t{};
    for (int len = 0; len <= 5; ++len)
        for (uint32_t code = 0; code < 32; ++code) t.h[len][code] = code < (1u << len) ? py_tuple_hash_calc(code, len) : 0;
    return t;
}
#if defined(__HIPCC__)
__device__ const TupleHashTable kTupleHashDev = make_tuple_hash_table();
#endif
static constexpr TupleHashTable kTupleHashHost = make_tuple_hash_table();

NHD_HD uint64_t py_tuple_hash(uint32_t code, int len, int base) {
    (void)base;
#if defined(__HIP_DEVICE_COMPILE__)
    return kTupleHashDev.h[len][code & 31u];
#else
    return kTupleHashHost.h[len][code & 31u];
#endif
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

// NOTE: every helper takes and returns the set BY VALUE.  With reference parameters LLVM turns the
// "which 64-bit word holds this slot" selects into address arithmetic on the struct, which pins all
// sets in scratch memory (400 B/lane, and a scratch-limited occupancy) instead of registers.
NHD_HD SmallSet ss_make(int len, int base) {
    SmallSet s;
    s.used = 0; s.mask = 7; s.fill = 0; s.len = len; s.base = base;
    s.k0 = s.k1 = s.k2 = s.k3 = 0;
    return s;
}
// next occupied slot >= from, or -1 (sets are walked in slot order = CPython iteration order)
NHD_HD int ss_next(SmallSet s, int from) {
    if (from > s.mask) return -1;
    const uint32_t m = s.used >> from;
    return m ? from + __builtin_ctz(m) : -1;
}
NHD_HD int ss_key(SmallSet s, int slot) {
    const uint64_t w = slot < 16 ? (slot < 8 ? s.k0 : s.k1) : (slot < 24 ? s.k2 : s.k3);
    return (int)((w >> ((slot & 7) * 8)) & 0xFF);
}
NHD_HD SmallSet ss_put(SmallSet s, int slot, int key) {
    const uint64_t v = (uint64_t)key << ((slot & 7) * 8);
    s.k0 |= slot < 8 ? v : 0;
    s.k1 |= (slot >= 8 && slot < 16) ? v : 0;
    s.k2 |= (slot >= 16 && slot < 24) ? v : 0;
    s.k3 |= slot >= 24 ? v : 0;
    s.used |= 1u << slot;
    return s;
}
// slot where `key` lives (>= 0), or -(free slot)-1 where it would be inserted
NHD_HD int ss_probe(SmallSet s, int key, uint64_t h, bool match) {
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
NHD_HD SmallSet ss_add(SmallSet s, int key) {
    const uint64_t h = py_tuple_hash((uint32_t)key, s.len, s.base);
    const int r = ss_probe(s, key, h, true);
    if (r >= 0) return s;
    s = ss_put(s, -r - 1, key);
    s.fill++;
    if (s.fill * 5 >= s.mask * 3) {                   // 8 -> 32 slots (set_table_resize(used*4)); never further
        const SmallSet o = s;
        s.used = 0; s.mask = 31; s.k0 = s.k1 = s.k2 = s.k3 = 0;
        for (int i = ss_next(o, 0); i >= 0; i = ss_next(o, i + 1)) {
            const int k = ss_key(o, i);
            s = ss_put(s, -ss_probe(s, k, py_tuple_hash((uint32_t)k, s.len, s.base), false) - 1, k);
        }
    }
    return s;
}
NHD_HD bool ss_has(SmallSet s, int key) {
    return ss_probe(s, key, py_tuple_hash((uint32_t)key, s.len, s.base), true) >= 0;
}
// a & b: iterate the smaller operand (b on ties) in slot order, probe the other (set_intersection)
NHD_HD SmallSet ss_intersect(SmallSet a, SmallSet b) {
    SmallSet out = ss_make(a.len, a.base);
    const bool swap = b.fill > a.fill;
    const SmallSet probe = swap ? b : a;
    const SmallSet iter = swap ? a : b;
    for (int i = ss_next(iter, 0); i >= 0; i = ss_next(iter, i + 1)) {
        const int k = ss_key(iter, i);
        if (ss_has(probe, k)) out = ss_add(out, k);
    }
    return out;
}
NHD_HD int ss_list(SmallSet s, int16_t* out) {
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
    NHD_HD static int next(const Set& s, int from) { for (int i = from; i <= s.mask; ++i) if (s.key[i] >= 0) return i; return -1; }
    NHD_HD static int key_at(const Set& s, int slot) { return s.key[slot]; }
};
struct SmallOps {
    typedef SmallSet Set;
    NHD_HD static void init(Set& s, int len, int base) { s = ss_make(len, base); }
    NHD_HD static void add(Set& s, int key, int, int) { s = ss_add(s, key); }
    NHD_HD static void isect(const Set& a, const Set& b, Set& o) { o = ss_intersect(a, b); }
    NHD_HD static int list(const Set& s, int16_t* o) { return ss_list(s, o); }
    NHD_HD static int size(const Set& s) { return s.fill; }
    NHD_HD static int next(const Set& s, int from) { return ss_next(s, from); }
    NHD_HD static int key_at(const Set& s, int slot) { return ss_key(s, slot); }
};

// ---- the winner's resource state -------------------------------------------------------------------
struct WinnerState {
    int U;                        // Node.numa_nodes (1 or 2)
    bool smt;
    int free_c[2], free_g[2];
    const nhdfit_detail* d;       // the winner's cold record (left in global memory: indexed dynamically)
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
// All small per-group state is nibble-packed into scalars so nothing lives in scratch memory.
NHD_HD uint32_t nib_get(uint32_t v, int i) { return (v >> (4 * i)) & 15u; }
NHD_HD uint32_t nib_set(uint32_t v, int i, uint32_t x) { return (v & ~(15u << (4 * i))) | (x << (4 * i)); }

NHD_HD bool first_nic_choice(const nhdfit_req& r, const WinnerState& w, uint32_t gcode, bool pci, int8_t nic_idx[kMaxG]) {
    const int G = (int)r.n_groups;
    uint32_t order = 0, numa = 0;                    // order: nibble pos -> group; numa: bit g -> NUMA of group g
    int n = 0;
    for (int u = 0; u < w.U; ++u)
        for (int g = 0; g < G; ++g)
            if (tup_digit(gcode, G, w.U, g) == u) { order = nib_set(order, n, (uint32_t)g); numa |= (uint32_t)u << g; ++n; }
    for (int g = 0; g < G; ++g)
        if (w.d->nic_cnt[(numa >> g) & 1] == 0) return false;
    uint32_t pick = 0;                               // nibble g -> NIC ordinal chosen for group g
    for (;;) {
        bool ok = true;
        for (int g = 0; g < G && ok; ++g) {
            const uint32_t u = (numa >> g) & 1, k = nib_get(pick, g);
            bool first_on_nic = true;
            for (int h = 0; h < g; ++h)
                if (((numa >> h) & 1) == u && nib_get(pick, h) == k) first_on_nic = false;
            if (!first_on_nic) continue;
            // the reference subtracts every group placed on this NIC from [cap, cap] in group order (Matcher.py:261-263)
            double rx = w.caps[w.d->nic_cls[u][k]], tx = rx;
            for (int h = g; h < G; ++h)
                if (((numa >> h) & 1) == u && nib_get(pick, h) == k) { rx = rx - r.rx[h]; tx = tx - r.tx[h]; }
            if (rx < 0 || tx < 0) ok = false;                                    // Matcher.py:267
        }
        if (ok && pci) {                                                         // Matcher.py:312-322
            for (int g = 0; g < G && ok; ++g) {
                const uint32_t s = w.d->nic_sw[(numa >> g) & 1][nib_get(pick, g)];
                uint32_t cnt = 0;
                for (int h = 0; h < G; ++h)
                    if (w.d->nic_sw[(numa >> h) & 1][nib_get(pick, h)] == s) ++cnt;
                if (cnt > w.d->sw_free[s]) ok = false;
            }
        }
        if (ok) {
            for (int g = 0; g < G; ++g) nic_idx[g] = (int8_t)nib_get(pick, g);
            return true;
        }
        int pos = G - 1;                                  // advance the odometer (last digit fastest)
        while (pos >= 0) {
            const int g = (int)nib_get(order, pos);
            const uint32_t v = nib_get(pick, g) + 1;
            if (v < w.d->nic_cnt[(numa >> g) & 1]) { pick = nib_set(pick, g, v); break; }
            pick = nib_set(pick, g, 0);
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

// GetNumaGroupIdx (Matcher.py:427-437) over one candidate list (= a set walked in slot order)
template <class Ops>
NHD_HD int pick_gpu_tuple(const typename Ops::Set& gset, int G, int U) {
    int best = -1, best_spread = -1;
    for (int i = Ops::next(gset, 0); i >= 0; i = Ops::next(gset, i + 1)) {
        const int k = Ops::key_at(gset, i);
        int ones = 0;
        for (int g = 0; g < G; ++g) ones += tup_digit((uint32_t)k, G, U, g);
        const int zeros = G - ones;
        const int spread = U == 1 ? 0 : (ones > zeros ? ones - zeros : zeros - ones);
        if (spread > best_spread) { best_spread = spread; best = k; }
    }
    return best;
}

// The order-dependent core of the mapping as a pure function of three small bit sets:
//   sg_mask / sc_mask / nic_codes : bit c = tuple code c is a valid GPU / CPU(+misc) / NIC assignment
// Returns false if the three prefix sets do not intersect, else the chosen GPU tuple and CPU tuple codes.
template <class Ops>
NHD_HD bool choose_tuples(int G, int U, uint32_t sg_mask, uint32_t sc_mask, uint32_t nic_codes, uint32_t& gcode, int& ccode) {
    typedef typename Ops::Set Set;
    const uint32_t nG = ipow(U, G), nC = ipow(U, G + 1);
    // candidate sets in product order (Matcher.py:116-141, 206-220).  list(set) = keys in slot order, so the
    // "lists" of the reference are never materialised: the sets' slots are walked instead.
    Set sg, sc;
    Ops::init(sg, G, U);
    Ops::init(sc, G + 1, U);
    for (uint32_t code = 0; code < nG; ++code)
        if (sg_mask >> code & 1) Ops::add(sg, (int)code, G, U);
    for (uint32_t code = 0; code < nC; ++code)
        if (sc_mask >> code & 1) Ops::add(sc, (int)code, G + 1, U);
    // intersection of the three prefix sets (Matcher.py:342-346): set(list) re-inserts in list order
    Set a, b, c, ab, abc;
    Ops::init(a, G, U); Ops::init(b, G, U); Ops::init(c, G, U);
    for (int i = Ops::next(sg, 0); i >= 0; i = Ops::next(sg, i + 1)) Ops::add(a, Ops::key_at(sg, i), G, U);
    for (int i = Ops::next(sc, 0); i >= 0; i = Ops::next(sc, i + 1)) Ops::add(b, Ops::key_at(sc, i) >> (U - 1), G, U);   // tuple[:-1]
    for (uint32_t code = 0; code < nG; ++code)
        if (nic_codes >> code & 1) Ops::add(c, (int)code, G, U);
    Ops::isect(a, b, ab);
    Ops::isect(ab, c, abc);
    if (Ops::size(abc) == 0) return false;
    // GPU list: replaced by the intersection only if that drops something (Matcher.py:363-366);
    // GetNumaGroupIdx (Matcher.py:427-437): first maximiser of max-min per-NUMA group count
    gcode = (uint32_t)(Ops::size(abc) < Ops::size(sg) ? pick_gpu_tuple<Ops>(abc, G, U) : pick_gpu_tuple<Ops>(sg, G, U));
    ccode = -1;                                                            // Matcher.py:441-444
    for (int i = Ops::next(sc, 0); i >= 0 && ccode < 0; i = Ops::next(sc, i + 1)) {
        const int k = Ops::key_at(sc, i);
        if ((uint32_t)(k >> (U - 1)) == gcode) ccode = k;
    }
    return ccode >= 0;
}

// Valid GPU / CPU(+misc) assignments of a pod on the winner as bit sets over tuple codes
// (Matcher.py:116-141, 206-220: the sets `stmp` before they are turned into lists).
NHD_HD void candidate_masks(const nhdfit_req& r, const WinnerState& w, uint32_t& sg_mask, uint32_t& sc_mask) {
    const int G = (int)r.n_groups, U = w.U;
    const uint32_t nG = ipow(U, G), nC = ipow(U, G + 1);
    sg_mask = sc_mask = 0;
    for (uint32_t code = 0; code < nG; ++code) {
        uint32_t t0 = 0, t1 = 0;
        for (int g = 0; g < G; ++g) { if (tup_digit(code, G, U, g)) t1 += r.gpus[g]; else t0 += r.gpus[g]; }
        if (t0 <= (uint32_t)w.free_g[0] && t1 <= (uint32_t)w.free_g[1]) sg_mask |= 1u << code;
    }
    for (uint32_t code = 0; code < nC; ++code) {
        uint32_t t0 = 0, t1 = 0;
        for (int g = 0; g <= G; ++g) {
            const uint32_t d = g < G ? (w.smt ? r.cpu_smt[g] : r.cpu_nosmt[g]) : (w.smt ? r.misc_smt : r.misc_nosmt);
            if (tup_digit(code, G + 1, U, g)) t1 += d; else t0 += d;
        }
        if (t0 <= (uint32_t)w.free_c[0] && t1 <= (uint32_t)w.free_c[1]) sc_mask |= 1u << code;
    }
}

// choose_tuples' whole input for G <= 3 in 35 bits (bit 63 marks "occupied" in the dedup table of the mapping
// kernels): pods whose winners offer the same candidate sets share one run of the sequential set model.
NHD_HD uint64_t shape_key(int G, int U, uint32_t sg_mask, uint32_t sc_mask, uint32_t nic_codes) {
    return (1ull << 63) | (uint64_t)(G & 3) | ((uint64_t)(U - 1) << 2) | ((uint64_t)(sg_mask & 0xFF) << 3) |
           ((uint64_t)(nic_codes & 0xFF) << 11) | ((uint64_t)(sc_mask & 0xFFFF) << 19);
}

// Fills the mapping once the GPU and CPU tuples are chosen (Matcher.py:446-452).
NHD_HD bool finish_mapping(const nhdfit_req& r, const WinnerState& w, uint32_t gcode, int ccode, nhdfit_mapping& out) {
    const int G = (int)r.n_groups, U = w.U;
    for (int g = 0; g < kMaxG; ++g) { out.gpu[g] = out.nic_numa[g] = out.nic_idx[g] = -1; }
    for (int g = 0; g <= kMaxG; ++g) out.cpu[g] = -1;
    if (!first_nic_choice(r, w, gcode, r.map_type == NHDFIT_MAP_PCI, out.nic_idx)) return false;
    for (int g = 0; g < G; ++g) {
        out.gpu[g] = (int8_t)tup_digit(gcode, G, U, g);
        out.nic_numa[g] = out.gpu[g];
    }
    for (int g = 0; g <= G; ++g) out.cpu[g] = (int8_t)tup_digit((uint32_t)ccode, G + 1, U, g);
    out.valid = 1;
    return true;
}

// Restatement of the winner-only tail of FindNode (Matcher.py:337-391 + 423-452).
// `nic_codes`: bit c set = assignment with tuple code c has at least one valid NIC choice (after the
// PCI pruning) - taken from the same reach tables the fit kernel used.  Returns false if infeasible.
template <class Ops>
NHD_HD bool map_winner_t(const nhdfit_req& r, const WinnerState& w, uint32_t nic_codes, nhdfit_mapping& out) {
    const int G = (int)r.n_groups, U = w.U;
    const uint32_t nG = ipow(U, G);
    out.valid = 0;
    nic_codes &= (nG >= 32 ? 0u : (1u << nG)) - 1u;
    uint32_t sg_mask, sc_mask;
    candidate_masks(r, w, sg_mask, sc_mask);
    if (!sg_mask || !sc_mask || !nic_codes) return false;
    uint32_t gcode = 0;
    int ccode = -1;
    if (!choose_tuples<Ops>(G, U, sg_mask, sc_mask, nic_codes, gcode, ccode)) return false;
    return finish_mapping(r, w, gcode, ccode, out);
}

// G <= 3: every set stays within 32 slots -> register-resident model; G == 4: generic model.
NHD_HD bool map_winner(const nhdfit_req& r, const WinnerState& w, uint32_t nic_codes, nhdfit_mapping& out) {
    if (r.n_groups <= 3) return map_winner_t<SmallOps>(r, w, nic_codes, out);
    return map_winner_t<GenericOps>(r, w, nic_codes, out);
}

}  // namespace nhdfit
