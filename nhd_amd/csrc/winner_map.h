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

// hash(tuple of `len` ints), digit i = (code / base^(len-1-i)) % base; hash(int k>=0) == k
NHD_HD uint64_t py_tuple_hash(uint32_t code, int len, int base) {
    uint64_t acc = kXX5;
    uint32_t div = 1;
    for (int i = 1; i < len; ++i) div *= (uint32_t)base;
    for (int i = 0; i < len; ++i) {
        const uint64_t lane = (code / div) % (uint32_t)base;
        div = div > 1 ? div / (uint32_t)base : 1;
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

// ---- the winner's resource state -------------------------------------------------------------------
struct WinnerState {
    int U;                        // Node.numa_nodes (1 or 2)
    bool smt;
    int free_c[2], free_g[2];
    nhdfit_detail d;
    const double* caps;           // capacity per class
};

// digit i of tuple code (length len, base U), i = 0 is the first element
NHD_HD int tup_digit(uint32_t code, int len, int U, int i) {
    uint32_t div = 1;
    for (int k = i + 1; k < len; ++k) div *= (uint32_t)U;
    return (int)((code / div) % (uint32_t)U);
}

NHD_HD uint32_t ipow(int b, int e) { uint32_t r = 1; while (e-- > 0) r *= (uint32_t)b; return r; }

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

// Full restatement of the winner-only tail of FindNode.  Returns false if the node is not feasible.
NHD_HD bool map_winner(const nhdfit_req& r, const WinnerState& w, nhdfit_mapping& out) {
    const int G = (int)r.n_groups, U = w.U;
    const bool pci = r.map_type == NHDFIT_MAP_PCI;
    const uint32_t nG = ipow(U, G), nC = ipow(U, G + 1);
    out.valid = 0;

    // candidate sets in product order (Matcher.py:116-141, 206-220, 239-268)
    PySet sg, sc, sn;
    ps_init(sg); ps_init(sc); ps_init(sn);
    uint32_t demand[kMaxG + 1];
    for (int g = 0; g < G; ++g) demand[g] = w.smt ? r.cpu_smt[g] : r.cpu_nosmt[g];
    demand[G] = w.smt ? r.misc_smt : r.misc_nosmt;
    for (uint32_t code = 0; code < nG; ++code) {
        uint32_t tot[2] = {0, 0};
        for (int g = 0; g < G; ++g) tot[tup_digit(code, G, U, g)] += r.gpus[g];
        bool ok = true;
        for (int u = 0; u < U; ++u) ok = ok && tot[u] <= (uint32_t)w.free_g[u];
        if (ok) ps_add(sg, (int16_t)code, py_tuple_hash(code, G, U));
    }
    for (uint32_t code = 0; code < nC; ++code) {
        uint32_t tot[2] = {0, 0};
        for (int g = 0; g <= G; ++g) tot[tup_digit(code, G + 1, U, g)] += demand[g];
        bool ok = true;
        for (int u = 0; u < U; ++u) ok = ok && tot[u] <= (uint32_t)w.free_c[u];
        if (ok) ps_add(sc, (int16_t)code, py_tuple_hash(code, G + 1, U));
    }
    int8_t first_nic[1 << kMaxG][kMaxG];
    bool nic_ok[1 << kMaxG];
    for (uint32_t code = 0; code < nG; ++code)
        nic_ok[code] = first_nic_choice(r, w, code, pci, first_nic[code]);
    if (sg.fill == 0 || sc.fill == 0) return false;

    // intersection of the three prefix sets (Matcher.py:342-346): set(list) re-inserts in list order
    int16_t lg[1 << kMaxG], lc[2 << kMaxG];
    const int ng = ps_list(sg, lg), nc = ps_list(sc, lc);
    PySet a, b, c, ab, abc;
    ps_init(a); ps_init(b); ps_init(c);
    for (int i = 0; i < ng; ++i) ps_add(a, lg[i], py_tuple_hash((uint32_t)lg[i], G, U));
    for (int i = 0; i < nc; ++i) {
        const int16_t pre = (int16_t)(lc[i] / U);                         // tuple[:-1]
        ps_add(b, pre, py_tuple_hash((uint32_t)pre, G, U));
    }
    bool any_nic = false;
    for (uint32_t code = 0; code < nG; ++code)
        if (nic_ok[code]) { ps_add(c, (int16_t)code, py_tuple_hash(code, G, U)); any_nic = true; }
    if (!any_nic) return false;
    ps_intersect(a, b, ab);
    ps_intersect(ab, c, abc);
    if (abc.fill == 0) return false;

    // GPU list: replaced by the intersection only if that drops something (Matcher.py:363-366)
    int16_t gl[1 << kMaxG];
    int ngl;
    if (abc.fill < sg.fill) ngl = ps_list(abc, gl);
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
        if ((uint32_t)(lc[i] / U) == gcode) { ccode = lc[i]; break; }
    if (ccode < 0 || !nic_ok[gcode]) return false;

    for (int g = 0; g < kMaxG; ++g) { out.gpu[g] = out.nic_numa[g] = out.nic_idx[g] = -1; }
    for (int g = 0; g <= kMaxG; ++g) out.cpu[g] = -1;
    for (int g = 0; g < G; ++g) {
        out.gpu[g] = (int8_t)tup_digit(gcode, G, U, g);
        out.nic_numa[g] = out.gpu[g];
        out.nic_idx[g] = first_nic[gcode][g];
    }
    for (int g = 0; g <= G; ++g) out.cpu[g] = (int8_t)tup_digit((uint32_t)ccode, G + 1, U, g);
    out.valid = 1;
    return true;
}

}  // namespace nhdfit
