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


// ---- register-resident variant for tables that never outgrow 32 slots (<= 16 keys < 16) ------------
// Same insertion / growth / intersection rules as PySet above.  Keys live in two 64-bit registers (4 bits per
// slot) and occupancy in a 32-bit mask, so the winner-mapping code does not touch scratch memory for pods with
// G <= 3 (tuples of length <= 4: at most 16 distinct tuple codes).
struct SmallSet {
    uint32_t used;                // occupied slots
    uint32_t present;             // keys in the set (bit = tuple code): membership without probing
    int mask, fill, len, base;
    uint64_t k0, k1;              // key of slot i: 4 bits each
};

// The slots CPython examines for a key, in order (set_add_entry: the home slot, its LINEAR_PROBES successors when
// they exist, then the perturbed recurrence), depend only on (tuple length, key, table size).
//  * home slots: compile-time constants per tuple length (3 / 5 bits per key) - the common case costs no memory access;
//  * the rest of the probe order: tabulated at compile time, 24 probes of 5 bits per (length, table size, key), read
//    only after a collision.  Rows that run out (never observed: a 32-slot table holds at most 16 keys here) fall
//    back to the live recurrence.
constexpr uint64_t home_slots(int len, uint64_t mask, int bits, uint32_t first, uint32_t count) {
    uint64_t w = 0;
    for (uint32_t k = 0; k < count; ++k)
        if (first + k < (1u << len)) w |= (py_tuple_hash_calc(first + k, len) & mask) << (bits * k);
    return w;
}
template <int LEN> struct HomeSlots {
    static constexpr uint64_t h7 = home_slots(LEN, 7, 3, 0, 16);        // keys 0..15, 3 bits each
    static constexpr uint64_t h31lo = home_slots(LEN, 31, 5, 0, 12);    // keys 0..11, 5 bits each
    static constexpr uint64_t h31hi = home_slots(LEN, 31, 5, 12, 4);    // keys 12..15
};
NHD_HD int home_slot(int len, int mask, int key) {
    if (mask == 7) {
        const uint64_t w = len == 1 ? HomeSlots<1>::h7 : len == 2 ? HomeSlots<2>::h7 : len == 3 ? HomeSlots<3>::h7 : HomeSlots<4>::h7;
        return (int)((w >> (3 * key)) & 7u);
    }
    const uint64_t lo = len == 1 ? HomeSlots<1>::h31lo : len == 2 ? HomeSlots<2>::h31lo : len == 3 ? HomeSlots<3>::h31lo : HomeSlots<4>::h31lo;
    const uint64_t hi = len == 1 ? HomeSlots<1>::h31hi : len == 2 ? HomeSlots<2>::h31hi : len == 3 ? HomeSlots<3>::h31hi : HomeSlots<4>::h31hi;
    return (int)(((key < 12 ? lo >> (5 * key) : hi >> (5 * (key - 12)))) & 31u);
}

constexpr int kProbeLen = 24;
struct ProbeRow { uint64_t lo, hi; };                     // probes 0..11 / 12..23
struct ProbeTable { ProbeRow row[5][2][16]; };            // [tuple length][0: 8 slots, 1: 32 slots][key]
constexpr ProbeTable make_probe_table() {
    ProbeTable t{};
    for (int len = 0; len <= 4; ++len)
        for (int mi = 0; mi < 2; ++mi)
            for (uint32_t key = 0; key < 16; ++key) {
                if (key >= (1u << len)) continue;
                const uint64_t mask = mi ? 31 : 7;
                const uint64_t h = py_tuple_hash_calc(key, len);
                uint64_t perturb = h, i = h & mask, lo = 0, hi = 0;
                int n = 0;
                while (n < kProbeLen) {
                    const int probes = (i + 9 <= mask) ? 9 : 0;
                    for (int j = 0; j <= probes && n < kProbeLen; ++j, ++n) {
                        if (n < 12) lo |= (i + j) << (5 * n); else hi |= (i + j) << (5 * (n - 12));
                    }
                    perturb >>= 5;
                    i = (i * 5 + 1 + perturb) & mask;
                }
                t.row[len][mi][key] = ProbeRow{lo, hi};
            }
    return t;
}
#if defined(__HIPCC__)
__device__ const ProbeTable kProbeDev = make_probe_table();
#endif
static constexpr ProbeTable kProbeHost = make_probe_table();
NHD_HD ProbeRow probe_row(int len, int mask, uint32_t key) {
#if defined(__HIP_DEVICE_COMPILE__)
    return kProbeDev.row[len][mask > 7][key & 15u];
#else
    return kProbeHost.row[len][mask > 7][key & 15u];
#endif
}

// NOTE: every helper takes and returns the set BY VALUE.  With reference parameters LLVM turns the
// "which 64-bit word holds this slot" selects into address arithmetic on the struct, which pins all
// sets in scratch memory instead of registers.
NHD_HD SmallSet ss_make(int len, int base) {
    SmallSet s;
    s.used = 0; s.present = 0; s.mask = 7; s.fill = 0; s.len = len; s.base = base;
    s.k0 = s.k1 = 0;
    return s;
}
// next occupied slot >= from, or -1 (sets are walked in slot order = CPython iteration order)
NHD_HD int ss_next(SmallSet s, int from) {
    if (from > s.mask) return -1;
    const uint32_t m = s.used >> from;
    return m ? from + __builtin_ctz(m) : -1;
}
NHD_HD int ss_key(SmallSet s, int slot) {
    return (int)(((slot < 16 ? s.k0 : s.k1) >> ((slot & 15) * 4)) & 15u);
}
NHD_HD SmallSet ss_put(SmallSet s, int slot, int key) {
    const uint64_t v = (uint64_t)key << ((slot & 15) * 4);
    s.k0 |= slot < 16 ? v : 0;
    s.k1 |= slot >= 16 ? v : 0;
    s.used |= 1u << slot;
    return s;
}
// first free slot in the order CPython examines them for `key` (the key is known to be absent)
NHD_HD int ss_free_slot(uint32_t used, int mask, int len, int key) {
    const int home = home_slot(len, mask, key);
    if (!(used >> home & 1)) return home;
    const ProbeRow r = probe_row(len, mask, (uint32_t)key);
    uint64_t w = r.lo;
    for (int n = 0; n < kProbeLen; ++n) {
        if (n == 12) w = r.hi;
        const int slot = (int)(w & 31u);
        if (!(used >> slot & 1)) return slot;
        w >>= 5;
    }
    // beyond the tabulated prefix: the live recurrence, from the start
    const uint64_t h = py_tuple_hash((uint32_t)key, len, 2);
    uint64_t perturb = h;
    uint32_t i = (uint32_t)(h & (uint64_t)mask);
    for (;;) {
        const int probes = (i + 9 <= (uint32_t)mask) ? 9 : 0;
        for (int j = 0; j <= probes; ++j)
            if (!(used >> (i + j) & 1)) return (int)(i + j);
        perturb >>= 5;
        i = (uint32_t)((i * 5 + 1 + perturb) & (uint64_t)mask);
    }
}
NHD_HD SmallSet ss_add(SmallSet s, int key) {
    if (s.present >> key & 1) return s;
    s = ss_put(s, ss_free_slot(s.used, s.mask, s.len, key), key);
    s.present |= 1u << key;
    s.fill++;
    if (s.fill * 5 >= s.mask * 3) {                   // 8 -> 32 slots (set_table_resize(used*4)); never further
        const SmallSet o = s;
        s.used = 0; s.mask = 31; s.k0 = s.k1 = 0;
        for (int i = ss_next(o, 0); i >= 0; i = ss_next(o, i + 1)) {
            const int k = ss_key(o, i);
            s = ss_put(s, ss_free_slot(s.used, 31, s.len, k), k);
        }
    }
    return s;
}
NHD_HD bool ss_has(SmallSet s, int key) { return (s.present >> key & 1) != 0; }

// A set filled in ascending code order (how the reference builds its candidate sets, Matcher.py:116-141, 206-220)
// is a pure function of (tuple length, which codes): its final layout is tabulated once per process - by this very
// model (asc_entry_build below, on the device at context creation / on the host on first use) - for all
// 4 + 16 + 256 + 65536 subsets of the 2, 4, 8 or 16 tuples of length 1..4.
struct AscEntry { uint32_t used, pad; uint64_t k0, k1; };                  // 24 bytes
constexpr uint32_t kAscOffset[5] = {0, 0, 4, 20, 276};                    // first entry of tuple length 1..4
constexpr uint32_t kAscEntries = 276 + 65536;
NHD_HD AscEntry asc_entry_build(int len, uint32_t subset) {
    SmallSet s = ss_make(len, 2);
    for (uint32_t code = 0; code < (1u << len); ++code)
        if (subset >> code & 1) s = ss_add(s, (int)code);
    return AscEntry{s.used, 0u, s.k0, s.k1};
}
NHD_HD SmallSet ss_from_asc(const AscEntry* table, int len, int base, uint32_t subset) {
    const AscEntry e = table[kAscOffset[len] + subset];
    SmallSet s;
    s.used = e.used; s.present = subset; s.fill = popc32(subset); s.mask = s.fill >= 5 ? 31 : 7;
    s.len = len; s.base = base; s.k0 = e.k0; s.k1 = e.k1;
    return s;
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
    // set filled with the codes of `subset` in ascending order
    NHD_HD static void from_subset(Set& s, const AscEntry*, int len, int base, uint32_t subset) {
        ps_init(s);
        for (uint32_t code = 0; code < 32; ++code)
            if (subset >> code & 1) ps_add(s, (int16_t)code, py_tuple_hash(code, len, base));
    }
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
    NHD_HD static void from_subset(Set& s, const AscEntry* table, int len, int base, uint32_t subset) {
        if (table) { s = ss_from_asc(table, len, base, subset); return; }
        s = ss_make(len, base);
        for (uint32_t code = 0; code < (1u << len); ++code)
            if (subset >> code & 1) s = ss_add(s, (int)code);
    }
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

// the enumeration as the reference writes it: every combination until one passes (kept for requests the pruning of
// first_nic_choice is not valid for)
NHD_HD bool first_nic_choice_plain(const nhdfit_req& r, const WinnerState& w, uint32_t gcode, bool pci, int8_t nic_idx[kMaxG]) {
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

NHD_HD bool first_nic_choice(const nhdfit_req& r, const WinnerState& w, uint32_t gcode, bool pci, int8_t nic_idx[kMaxG]) {
    const int G = (int)r.n_groups;
    uint32_t order = 0, numa = 0;                    // order: nibble pos -> group; numa: bit g -> NUMA of group g
    int n = 0;
    for (int u = 0; u < w.U; ++u)
        for (int g = 0; g < G; ++g)
            if (tup_digit(gcode, G, w.U, g) == u) { order = nib_set(order, n, (uint32_t)g); numa |= (uint32_t)u << g; ++n; }
    for (int g = 0; g < G; ++g)
        if (w.d->nic_cnt[(numa >> g) & 1] == 0) return false;
    for (int g = 0; g < G; ++g)
        if (!(r.rx[g] >= 0) || !(r.tx[g] >= 0)) return first_nic_choice_plain(r, w, gcode, pci, nic_idx);   // negative / NaN speeds: no pruning
    // The reference walks itertools.product over the groups' NIC lists (groups ordered by NUMA node, then index; last
    // group fastest) and takes the first combination that passes (Matcher.py:261-267, 312-322).  Both tests only get
    // harder as groups are added - a NIC's head-room goes down with every subtraction (requests are >= 0; the groups of a
    // NIC are subtracted in group order, which is the order they are assigned here), a switch's group count goes up - so
    // a prefix that already fails has no valid completion: depth-first in the same order with that pruning returns the
    // same first combination without visiting n^G of them when the leading NICs are claimed.
    uint32_t pick = 0;                               // nibble g -> NIC ordinal chosen for group g
    auto prefix_ok = [&](int pos) {                  // groups order[0..pos] assigned: does the newest one still fit?
        const int g = (int)nib_get(order, pos);
        const uint32_t u = (numa >> g) & 1, k = nib_get(pick, g);
        double rx = w.caps[w.d->nic_cls[u][k]], tx = rx;
        for (int q = 0; q <= pos; ++q) {             // same NUMA node => ascending group index along `order`
            const int h = (int)nib_get(order, q);
            if (((numa >> h) & 1) == u && nib_get(pick, h) == k) { rx = rx - r.rx[h]; tx = tx - r.tx[h]; }
        }
        if (rx < 0 || tx < 0) return false;                                      // Matcher.py:267
        if (pci) {                                                               // Matcher.py:312-322
            const uint32_t sw = w.d->nic_sw[u][k];
            uint32_t cnt = 0;
            for (int q = 0; q <= pos; ++q) {
                const int h = (int)nib_get(order, q);
                if (w.d->nic_sw[(numa >> h) & 1][nib_get(pick, h)] == sw) ++cnt;
            }
            if (cnt > w.d->sw_free[sw]) return false;
        }
        return true;
    };
    int pos = 0;
    for (;;) {
        if (prefix_ok(pos)) {
            if (pos == G - 1) {
                for (int g = 0; g < G; ++g) nic_idx[g] = (int8_t)nib_get(pick, g);
                return true;
            }
            ++pos;                                    // next group starts at its first NIC (its nibble is 0)
            continue;
        }
        for (;;) {                                    // next candidate: advance this digit, or back up
            const int g = (int)nib_get(order, pos);
            const uint32_t v = nib_get(pick, g) + 1;
            if (v < w.d->nic_cnt[(numa >> g) & 1]) { pick = nib_set(pick, g, v); break; }
            pick = nib_set(pick, g, 0);
            if (--pos < 0) return false;
        }
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
NHD_HD bool choose_tuples(int G, int U, uint32_t sg_mask, uint32_t sc_mask, uint32_t nic_codes, uint32_t& gcode, int& ccode,
                          const AscEntry* asc = nullptr) {
    typedef typename Ops::Set Set;
    const uint32_t nG = ipow(U, G), nC = ipow(U, G + 1);
    // candidate sets in product order (Matcher.py:116-141, 206-220).  list(set) = keys in slot order, so the
    // "lists" of the reference are never materialised: the sets' slots are walked instead.  `asc`: the table of
    // ascending-filled sets (AscEntry), or null to build them insertion by insertion.
    Set sg, sc;
    Ops::from_subset(sg, asc, G, U, sg_mask & ((1u << nG) - 1u));
    Ops::from_subset(sc, asc, G + 1, U, sc_mask & (nC >= 32 ? ~0u : (1u << nC) - 1u));
    // intersection of the three prefix sets (Matcher.py:342-346): set(list) re-inserts in list order
    Set a, b, c, ab, abc;
    Ops::init(a, G, U); Ops::init(b, G, U);
    for (int i = Ops::next(sg, 0); i >= 0; i = Ops::next(sg, i + 1)) Ops::add(a, Ops::key_at(sg, i), G, U);
    for (int i = Ops::next(sc, 0); i >= 0; i = Ops::next(sc, i + 1)) Ops::add(b, Ops::key_at(sc, i) >> (U - 1), G, U);   // tuple[:-1]
    Ops::from_subset(c, asc, G, U, nic_codes & ((1u << nG) - 1u));
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

// ---- the whole of choose_tuples, tabulated for the small shapes ------------------------------------------
// With two NUMA nodes and G <= 2 proc groups the order-dependent core has only 2+4+2 resp. 4+8+4 input bits:
// 256 + 65 536 one-byte answers, filled once per context by choose_tuples itself (k_build_choose) - pods of these
// shapes (90 % of the BASELINE mix) need no run of the sequential model at all, only the G = 3 shapes do.
constexpr uint32_t kChooseOffset[3] = {0, 0, 256};                        // first entry of G = 1, 2
constexpr uint32_t kChooseEntries = 256 + 65536;
NHD_HD bool choose_tabulated(int G, int U) { return U == 2 && G >= 1 && G <= 2; }
NHD_HD uint32_t choose_index(int G, uint32_t sg_mask, uint32_t sc_mask, uint32_t nic_codes) {
    const uint32_t nG = 1u << G, nC = 2u << G;                            // tuple codes per set (U = 2)
    return kChooseOffset[G] + ((sg_mask & ((1u << nG) - 1u)) | (sc_mask & ((1u << nC) - 1u)) << nG |
                               (nic_codes & ((1u << nG) - 1u)) << (nG + nC));
}
// result word as the choose role writes it: ok << 8 | gcode << 4 | ccode
NHD_HD uint32_t choose_result_word(bool ok, uint32_t gcode, int ccode) {
    return ((uint32_t)ok << 8) | ((gcode & 7u) << 4) | ((uint32_t)ccode & 15u);
}
NHD_HD uint8_t choose_entry_build(const AscEntry* asc, uint32_t entry) {
    const int G = entry >= kChooseOffset[2] ? 2 : 1;
    const uint32_t nG = 1u << G, nC = 2u << G, x = entry - kChooseOffset[G];
    const uint32_t sg = x & ((1u << nG) - 1u), sc = (x >> nG) & ((1u << nC) - 1u), nic = x >> (nG + nC);
    uint32_t gcode = 0;
    int ccode = -1;
    const bool ok = sg && sc && nic && choose_tuples<SmallOps>(G, 2, sg, sc, nic, gcode, ccode, asc);
    return (uint8_t)(ok ? 0x80u | ((gcode & 7u) << 4) | ((uint32_t)ccode & 15u) : 0u);   // G <= 2: gcode < 4, ccode < 8
}
NHD_HD uint32_t choose_from_table(const uint8_t* table, int G, uint32_t sg_mask, uint32_t sc_mask, uint32_t nic_codes) {
    const uint32_t e = table[choose_index(G, sg_mask, sc_mask, nic_codes)];
    return choose_result_word((e & 0x80u) != 0, (e >> 4) & 7u, (int)(e & 15u));
}

// Valid GPU / CPU(+misc) assignments of a pod on the winner as bit sets over tuple codes
// (Matcher.py:116-141, 206-220: the sets `stmp` before they are turned into lists).
NHD_HD void candidate_masks(const nhdfit_req& r, const WinnerState& w, uint32_t& sg_mask, uint32_t& sc_mask) {
    const int G = (int)r.n_groups, U = w.U;
    const uint32_t nG = ipow(U, G), nC = ipow(U, G + 1);
    // The groups' demands are read ONCE into registers (groups past G count zero): on the device the request sits in LDS or global
    // memory, and the (code, group) loops below used to re-read a field per iteration - ~90 dependent round trips for a three-group
    // pod, 11 of a drain launch's 34 us (round 6, profiles/r06/drain_phases.log).
    uint32_t gp[kMaxG], cd[kMaxG];
#pragma unroll
    for (int g = 0; g < kMaxG; ++g) {
        gp[g] = g < G ? (uint32_t)r.gpus[g] : 0u;
        cd[g] = g < G ? (uint32_t)(w.smt ? r.cpu_smt[g] : r.cpu_nosmt[g]) : 0u;
    }
    const uint32_t misc = w.smt ? r.misc_smt : r.misc_nosmt;
    const uint32_t fg0 = (uint32_t)w.free_g[0], fg1 = (uint32_t)w.free_g[1], fc0 = (uint32_t)w.free_c[0], fc1 = (uint32_t)w.free_c[1];
    sg_mask = sc_mask = 0;
    for (uint32_t code = 0; code < nG; ++code) {
        uint32_t t1 = 0, all = 0;
#pragma unroll
        for (int g = 0; g < kMaxG; ++g) {                  // tup_digit(code, G, U, g) = bit G - 1 - g of the code
            all += gp[g];
            if (g < G && (code >> (G - 1 - g) & 1u)) t1 += gp[g];
        }
        if (all - t1 <= fg0 && t1 <= fg1) sg_mask |= 1u << code;
    }
    for (uint32_t code = 0; code < nC; ++code) {
        uint32_t t1 = (code & 1u) ? misc : 0u, all = misc;     // the misc cores are the tuple's last digit: bit 0
#pragma unroll
        for (int g = 0; g < kMaxG; ++g) {                  // tup_digit(code, G + 1, U, g) = bit G - g
            all += cd[g];
            if (g < G && (code >> (G - g) & 1u)) t1 += cd[g];
        }
        if (all - t1 <= fc0 && t1 <= fc1) sc_mask |= 1u << code;
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
