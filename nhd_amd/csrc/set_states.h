// set_states.h - CPython sets of 3-tuples over {0,1} as a finite-state machine.
//
// For pods with three proc groups on a two-NUMA node the order-dependent core of the mapping (choose_tuples,
// winner_map.h) has 8 + 16 + 8 input bits - too many to tabulate outright as is done for G <= 2.  But everything it
// does to sets of GPU tuples (build, re-insert in list order, intersect) is a sequence of insertions of keys 0..7 into
// an initially empty set, and a CPython set's behaviour depends only on its LAYOUT (table size + which key sits in
// which slot).  The layouts reachable by inserting distinct keys 0..7 in any order are enumerated once (breadth
// first, with the register model of winner_map.h doing every insertion) into
//     info[state]    : keys in slot order (8 x 3 bits), key count, GetNumaGroupIdx's pick for that list, membership
//     next[state][k] : the state after set.add(k)
//     asc[subset]    : the state of the set filled with `subset` in ascending order
// after which choose_tuples for such a pod is ~40 dependent table look-ups instead of ~1 000 scalar instructions.
// Equivalence with choose_tuples is checked exhaustively-by-sampling on the host (tests/test_pyset_emulation.py).
#pragma once
#include "winner_map.h"

#include <unordered_map>
#include <vector>

namespace nhdfit {

struct SetStates {
    const uint64_t* info;      // [n]
    const uint32_t* next;      // [n][8]
    const uint32_t* asc;       // [256]
    uint32_t n;
};

NHD_HD uint32_t st_key(uint64_t w, uint32_t j) { return (uint32_t)(w >> (3 * j)) & 7u; }
NHD_HD uint32_t st_count(uint64_t w) { return (uint32_t)(w >> 24) & 15u; }
NHD_HD uint32_t st_pick(uint64_t w) { return (uint32_t)(w >> 28) & 7u; }
NHD_HD uint32_t st_present(uint64_t w) { return (uint32_t)(w >> 32) & 0xFFu; }

// out = iter-order walk of (a & b) exactly as set_intersection does it: iterate the smaller operand (b on ties) in
// slot order, keep the keys the other one holds, add them to an empty set in that order
NHD_HD uint32_t st_intersect(const SetStates& T, uint32_t a, uint32_t b) {
    const uint64_t wa = T.info[a], wb = T.info[b];
    const bool swap = st_count(wb) > st_count(wa);
    const uint64_t iter = swap ? wa : wb;
    const uint32_t probe = st_present(swap ? wb : wa);
    uint32_t out = 0;
    for (uint32_t j = 0; j < st_count(iter); ++j) {
        const uint32_t k = st_key(iter, j);
        if (probe >> k & 1) out = T.next[out * 8 + k];
    }
    return out;
}

// choose_tuples<SmallOps>(3, 2, sg_mask, sc_mask, nic_codes, ...) as a result word (ok << 8 | gcode << 4 | ccode)
NHD_HD uint32_t choose_g3(const SetStates& T, const AscEntry* asc, uint32_t sg_mask, uint32_t sc_mask, uint32_t nic_codes) {
    sg_mask &= 0xFFu; sc_mask &= 0xFFFFu; nic_codes &= 0xFFu;
    const uint32_t sgS = T.asc[sg_mask], cS = T.asc[nic_codes];
    const SmallSet sc = ss_from_asc(asc, 4, 2, sc_mask);
    // a = set(list(sg)): re-insertion in slot order;  b = set(t[:-1] for t in list(sc))
    const uint64_t wsg = T.info[sgS];
    uint32_t a = 0, b = 0;
    for (uint32_t j = 0; j < st_count(wsg); ++j) a = T.next[a * 8 + st_key(wsg, j)];
    for (int i = ss_next(sc, 0); i >= 0; i = ss_next(sc, i + 1)) b = T.next[b * 8 + ((uint32_t)ss_key(sc, i) >> 1)];
    const uint32_t abc = st_intersect(T, st_intersect(T, a, b), cS);
    const uint64_t wabc = T.info[abc];
    if (st_count(wabc) == 0) return 0;
    const uint32_t gcode = st_count(wabc) < st_count(wsg) ? st_pick(wabc) : st_pick(wsg);
    int ccode = -1;
    for (int i = ss_next(sc, 0); i >= 0 && ccode < 0; i = ss_next(sc, i + 1)) {
        const int k = ss_key(sc, i);
        if ((uint32_t)(k >> 1) == gcode) ccode = k;
    }
    return choose_result_word(ccode >= 0, gcode, ccode);
}

// Host-side enumeration (context creation / tests).  State 0 = the empty set.
inline void build_set_states(std::vector<uint64_t>& info, std::vector<uint32_t>& next, std::vector<uint32_t>& asc) {
    struct Key {
        uint32_t used; int mask; uint64_t k0, k1;
        bool operator==(const Key& o) const { return used == o.used && mask == o.mask && k0 == o.k0 && k1 == o.k1; }
    };
    struct Hash {
        size_t operator()(const Key& k) const {
            uint64_t h = k.used * 0x9E3779B97F4A7C15ull ^ (uint64_t)k.mask;
            h = (h ^ k.k0) * 0xC2B2AE3D27D4EB4Full;
            h = (h ^ k.k1) * 0x165667B19E3779F9ull;
            return (size_t)(h ^ (h >> 29));
        }
    };
    std::unordered_map<Key, uint32_t, Hash> ids;
    std::vector<SmallSet> sets;
    auto intern = [&](const SmallSet& s) -> uint32_t {
        const Key key{s.used, s.mask, s.k0, s.k1};
        auto it = ids.find(key);
        if (it != ids.end()) return it->second;
        const uint32_t id = (uint32_t)sets.size();
        ids.emplace(key, id);
        sets.push_back(s);
        return id;
    };
    intern(ss_make(3, 2));
    next.clear();
    for (uint32_t s = 0; s < sets.size(); ++s) {                 // breadth first: `sets` grows while it is walked
        for (int k = 0; k < 8; ++k) {
            const SmallSet cur = sets[s];
            const uint32_t t = (cur.present >> k & 1) ? s : intern(ss_add(cur, k));
            next.push_back(t);
        }
    }
    info.resize(sets.size());
    for (uint32_t s = 0; s < sets.size(); ++s) {
        const SmallSet& x = sets[s];
        uint64_t w = 0;
        uint32_t j = 0;
        for (int i = ss_next(x, 0); i >= 0; i = ss_next(x, i + 1)) w |= (uint64_t)ss_key(x, i) << (3 * j++);
        w |= (uint64_t)j << 24;
        if (j) w |= (uint64_t)(pick_gpu_tuple<SmallOps>(x, 3, 2) & 7) << 28;
        w |= (uint64_t)(x.present & 0xFFu) << 32;
        info[s] = w;
    }
    asc.assign(256, 0);
    for (uint32_t subset = 0; subset < 256; ++subset) {
        uint32_t s = 0;
        for (int k = 0; k < 8; ++k)
            if (subset >> k & 1) s = next[s * 8 + k];
        asc[subset] = s;
    }
}

}  // namespace nhdfit
