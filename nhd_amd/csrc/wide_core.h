// wide_core.h - the general path: nodes beyond the fast layout (3 or 4 sockets, 65..128 physical cores per socket),
// evaluated by explicit enumeration with the reference's own arithmetic.  Written once for the gfx950 kernels
// (wide_kernel.h) and the host twin of the tests.
//
// The table-driven pass (fit_core.h) is built around two sockets of at most 64 physical cores: an assignment is a bit
// pattern, a table row a mask over 2^G assignments.  The reference itself is general - it enumerates
// itertools.product(range(v.numa_nodes), repeat=len(req)) for whatever v.numa_nodes is (nhd/Matcher.py:118, 203, 242) over
// whatever cores_per_proc the labels say (nhd/Node.py:257, 336-350).  A node outside the fast shape is therefore carried
// as ONE self-contained record (nhdfit_wide_node, include/nhdfit.h) and every question the path asks about it is answered
// here the way the reference answers it, assignment by assignment:
//
//   wide_fits      FilterPodResources + the GPU / CPU / NIC stages + the PCI pruning + the intersection   Matcher.py:65-391
//   wide_map       GetNumaGroupIdx over the lists those stages leave, i.e. over CPython sets of int       Matcher.py:337-452
//                  tuples: the set model of winner_map.h for tuples over range(U), U <= 4, with tables of
//                  up to 4 096 slots held in caller-provided scratch (keys only: hashes are recomputed)
//   wide_commit    SetBusy + SetPhysicalIdsFromMapping + ClaimPodNICResources                             Node.py:663-841, 644
//
// Nothing here is fast; everything here is exact (f64 NIC arithmetic in the reference's order, -ffp-contract=off).
#pragma once
#include "winner_map.h"

namespace nhdfit {

constexpr int kWideU = NHDFIT_WIDE_MAX_NUMA;
NHD_HD uint32_t wide_ipow(uint32_t b, uint32_t e) { uint32_t r = 1; for (uint32_t i = 0; i < e; ++i) r *= b; return r; }
// the set model's tables for an ORDINARY request (nhdfit_req on a wide node): U^(G+1) <= 4^5 tuples, int16 keys
constexpr int kWideMaxTuples = 1024;
constexpr int kWideSetSlotsG = 1024;                 // a CPython set of <= 256 keys never outgrows 1 024 slots (growth: fill*5 >= mask*3 -> 4 x used)
constexpr int kWideSetSlotsC = 4096;                 // ... of <= 1 024 keys: 2 048; 4 096 leaves the model room to say so itself
// scratch of one mapping (keys): sg, a, b, c, ab, abc over G-tuples; sc over (G+1)-tuples; one resize buffer
constexpr int kWideScratchWords = 6 * kWideSetSlotsG + 2 * kWideSetSlotsC;
// ... for a BIG request the tables are sized per call: the slots a CPython set of `keys` distinct keys ends up with (the growth
// sequence depends on the count alone: fill*5 >= mask*3 -> smallest power of two > used*4, or > used*2 beyond 50 000 keys),
// int32 keys (tuple codes reach 4^9)
NHD_HD uint32_t wide_table_slots(uint32_t keys) {
    uint32_t size = 8;
    for (;;) {
        const uint32_t at = ((size - 1) * 3 + 4) / 5;                   // first fill with fill*5 >= mask*3
        if (at > keys) return size;
        const uint64_t minused = at > 50000 ? (uint64_t)at * 2 : (uint64_t)at * 4;
        uint32_t ns = 8;
        while ((uint64_t)ns <= minused) ns <<= 1;
        size = ns;
    }
}
NHD_HD size_t big_scratch_words(uint32_t U, uint32_t G) {             // of one mapping, int32 words (wide_map's carving)
    return 6 * (size_t)wide_table_slots(wide_ipow(U, G)) + 2 * (size_t)wide_table_slots(wide_ipow(U, G + 1));
}
static_assert(sizeof(nhdfit_wide_node) == 640 && sizeof(nhdfit_wide_placement) == 480, "record sizes of include/nhdfit.h");

// digit i (i = 0: first element) of tuple `code` of length `len` over range(U): first element most significant, so that
// ascending codes are itertools.product order
NHD_HD uint32_t wide_digit(uint32_t code, uint32_t len, uint32_t U, uint32_t i) {
    for (uint32_t k = i + 1; k < len; ++k) code /= U;
    return code % U;
}

// ---- the node's free resources ------------------------------------------------------------------------------------------
NHD_HD bool wide_bit(const uint64_t* w, uint32_t c) { return (w[c >> 6] >> (c & 63)) & 1u; }
NHD_HD void wide_clear(uint64_t* w, uint32_t c) { w[c >> 6] &= ~(1ull << (c & 63)); }

struct WideFree { uint32_t U; bool smt; uint32_t c[kWideU], g[kWideU]; };

NHD_HD bool wide_shape_ok(const nhdfit_wide_node& n) {
    return n.numa_nodes >= 1 && n.numa_nodes <= kWideU && n.cores_per_proc >= 1 && n.cores_per_proc <= NHDFIT_WIDE_MAX_CORES_PER_NUMA &&
           n.n_gpus <= NHDFIT_MAX_GPUS;
}

// GetFreeCpuCores (nhd/Node.py:250-264) / GetFreeNumaGPUs (456-462)
NHD_HD WideFree wide_free(const nhdfit_wide_node& n) {
    WideFree f;
    f.U = n.numa_nodes;
    f.smt = (n.flags & NHDFIT_NF_SMT) != 0;
    for (uint32_t u = 0; u < (uint32_t)kWideU; ++u) { f.c[u] = 0; f.g[u] = 0; }
    const uint32_t cpp = n.cores_per_proc;
    for (uint32_t u = 0; u < f.U; ++u) {             // cores [u * cpp, (u + 1) * cpp) of the flat bitmaps, word by word
        const uint32_t lo = u * cpp, hi = lo + cpp;
        for (uint32_t w = lo >> 6; w < (uint32_t)NHDFIT_WIDE_CORE_WORDS && w * 64u < hi; ++w) {
            uint64_t m = n.t0[w] & n.t1[w];
            const uint32_t s = w * 64u;
            if (lo > s) m &= ~0ull << (lo - s);
            if (hi < s + 64u) m &= ~0ull >> (s + 64u - hi);
            f.c[u] += (uint32_t)popc64(m);
        }
    }
    for (uint32_t x = 0; x < n.n_gpus; ++x)
        if ((n.gpu_free >> x & 1u) && n.gpu_numa[x] < f.U) f.g[n.gpu_numa[x]]++;
    return f;
}
// GetFreeGPUPCICount (nhd/Node.py:266-273) for one local switch id
NHD_HD uint32_t wide_sw_free(const nhdfit_wide_node& n, uint32_t sw) {
    uint32_t k = 0;
    for (uint32_t x = 0; x < n.n_gpus; ++x)
        if ((n.gpu_free >> x & 1u) && n.gpu_sw[x] == sw) ++k;
    return k;
}

// An ordinary node - five planes and its detail record - read through the wide record: what the general path needs to answer
// for it (fits, map; the commit step of an ordinary node stays on the planes, commit_core.h).  Socket u's cores sit in word u
// of the flat bitmaps (64 cores per socket: bits past the node's own core count are never set in t0).  A placeholder (a wide
// node's entry in the planes, a node no layout holds) comes out with numa_nodes == 0: wide_shape_ok says no.
NHD_HD void wide_view(const nhdfit_plane0& p0, const nhdfit_plane1& p1, const nhdfit_plane2& p2, const nhdfit_plane3& p3,
                      const nhdfit_plane4& p4, const nhdfit_detail& d, uint32_t index, nhdfit_wide_node& w) {
    for (int k = 0; k < NHDFIT_WIDE_CORE_WORDS; ++k) w.t0[k] = w.t1[k] = w.o0[k] = w.o1[k] = 0;
    for (int u = 0; u < NHDFIT_MAX_NUMA; ++u) { w.t0[u] = p0.t0[u]; w.t1[u] = p1.t1[u]; }
    w.groups = p3.groups;
    w.busy_time = p4.busy_time;
    w.gpu_free = p2.gpu_free;
    w.flags = p2.flags;
    w.hp_free = p2.hp_free; w.hp_total = 0;
    w.index = index;
    w.cores_per_proc = NHDFIT_MAX_CORES_PER_NUMA;
    w.numa_nodes = d.numa_nodes <= NHDFIT_MAX_NUMA ? d.numa_nodes : 0;
    w.n_gpus = d.n_gpus;
    for (int u = 0; u < NHDFIT_WIDE_MAX_NUMA; ++u) {
        w.nic_cnt[u] = u < NHDFIT_MAX_NUMA ? d.nic_cnt[u] : 0;
        for (int k = 0; k < NHDFIT_MAX_NICS_PER_NUMA; ++k) {
            w.nic_cls[u][k] = u < NHDFIT_MAX_NUMA ? d.nic_cls[u][k] : 0;
            w.nic_sw[u][k] = u < NHDFIT_MAX_NUMA ? d.nic_sw[u][k] : 0;
            w.nic_base[u][k] = 0;
            w.nic_pods[u][k] = 0;
        }
    }
    for (int x = 0; x < NHDFIT_MAX_GPUS; ++x) { w.gpu_numa[x] = (uint8_t)(p2.gpu_numa1 >> x & 1u); w.gpu_sw[x] = d.gpu_sw[x]; }
    for (int k = 0; k < (int)sizeof w.pad; ++k) w.pad[k] = 0;
}

// ---- stages ------------------------------------------------------------------------------------------------------------
// GPU stage, one assignment (Matcher.py:120-131)
template <class R> NHD_HD bool wide_gpu_ok(const R& r, const WideFree& f, uint32_t code) {
    uint32_t ttl[kWideU] = {0, 0, 0, 0};
    for (uint32_t g = 0; g < r.n_groups; ++g) ttl[wide_digit(code, r.n_groups, f.U, g)] += r.gpus[g];
    for (uint32_t u = 0; u < f.U; ++u)
        if (ttl[u] > f.g[u]) return false;
    return true;
}
// CPU stage, one (G+1)-tuple: the last element places the pod-level misc cores (Matcher.py:206-216)
template <class R> NHD_HD bool wide_cpu_ok(const R& r, const WideFree& f, uint32_t code) {
    uint32_t ttl[kWideU] = {0, 0, 0, 0};
    const uint32_t len = r.n_groups + 1;
    for (uint32_t g = 0; g < len; ++g) {
        const uint32_t d = g < r.n_groups ? (f.smt ? r.cpu_smt[g] : r.cpu_nosmt[g]) : (f.smt ? r.misc_smt : r.misc_nosmt);
        ttl[wide_digit(code, len, f.U, g)] += d;
    }
    for (uint32_t u = 0; u < f.U; ++u)
        if (ttl[u] > f.c[u]) return false;
    return true;
}

// First NIC choice, in the reference's enumeration order (Matcher.py:242-268: itertools.product over the NUMA nodes of
// itertools.product(range(K_u), repeat=#groups on u) - an odometer whose most significant digits are the groups of NUMA node
// 0 in ascending group index, then those of NUMA node 1, ...), that passes the bandwidth test (261-267: each NIC's [cap, cap]
// minus the requests of its groups in group order, nothing below zero) and, in PCI mode, the switch test (312-322: no more
// groups behind a switch than it has free GPUs).  Both tests only get harder as groups are added when every request is
// >= 0, so depth-first search in the same order with prefix pruning returns the same first combination (first_nic_choice,
// winner_map.h); requests that are negative or NaN take the plain odometer.
//
// Big requests (5..8 groups, req_traits<R>::kBig) search with two more things, neither of which changes the answer:
//   * among NICs of one NUMA node that are interchangeable - same capacity class (bit-identical f64), same switch - and that
//     no earlier group of the search order uses yet, only the lowest index is tried: a solution whose first use of such a NIC
//     k is ahead of its first use of an equal NIC k' < k becomes, with the roles of k and k' swapped throughout, a solution
//     that is earlier in the enumeration order (same per-NIC subtraction sequences, same switch counts), so the FIRST
//     solution is never in a pruned branch.  Eight VFs of one PF then cost set partitions, not 8^G.
//   * a budget of search steps per (pod, node) pair (NicSearch): when it runs out the pair is reported, the call fails
//     (NHDFIT_E_LIMIT) - the reference would be making K^G deepcopies there.
struct NicSearch { uint32_t left; bool exhausted; };        // a big request's budget of search steps per (pod, node) pair (see wide_nic_choice)
// What a NIC can still carry, per direction (nhd/Node.py:283-296).  Shipped arithmetic (ENABLE_SHARING = False): the capacity class
// of the NIC as it is now - 0 while a pod uses it, speed * 0.9 otherwise - for both directions.  ENABLE_SHARING = True (`sh` = the
// node's nhdfit_wide_share): `speed * 0.9 - speed_used[x]`, ONE f64 subtraction from the class value of the NIC's own capacity,
// as the reference writes it; pods_used plays no part.  A bare capacity table converts (every caller of the shipped arithmetic).
struct WideCaps {
    const double* cls;
    const nhdfit_wide_share* sh;
    NHD_HD WideCaps(const double* c, const nhdfit_wide_share* s = nullptr) : cls(c), sh(s) {}
    NHD_HD double free_of(const nhdfit_wide_node& n, uint32_t u, uint32_t k, uint32_t dir) const {
        return sh ? cls[n.nic_base[u][k]] - sh->used[u][k][dir] : cls[n.nic_cls[u][k]];
    }
    // NICs a search may treat as interchangeable: same price in both directions (and, the caller checks, the same switch)
    NHD_HD bool same_price(const nhdfit_wide_node& n, uint32_t u, uint32_t k, uint32_t k2) const {
        if (!sh) return n.nic_cls[u][k] == n.nic_cls[u][k2];
        return n.nic_base[u][k] == n.nic_base[u][k2] && sh->used[u][k][0] == sh->used[u][k2][0] && sh->used[u][k][1] == sh->used[u][k2][1];
    }
};
template <class R>
NHD_HD bool wide_nic_choice(const nhdfit_wide_node& n, const R& r, const WideCaps& caps, uint32_t gcode, int8_t* nic_idx, NicSearch* ns = nullptr) {
    constexpr int kG = req_traits<R>::kG;
    const uint32_t G = r.n_groups, U = n.numa_nodes;
    const bool pci = r.map_type == NHDFIT_MAP_PCI;
    uint32_t order[kG], numa[kG], pick[kG];
    uint32_t cnt = 0;
    for (uint32_t g = 0; g < G; ++g) { numa[g] = wide_digit(gcode, G, U, g); pick[g] = 0; }
    for (uint32_t u = 0; u < U; ++u)
        for (uint32_t g = 0; g < G; ++g)
            if (numa[g] == u) order[cnt++] = g;
    for (uint32_t g = 0; g < G; ++g)
        if (n.nic_cnt[numa[g]] == 0) return false;             // a NUMA node without NICs hosts no group (quirk Q3)
    bool prune = true;
    for (uint32_t g = 0; g < G; ++g)
        if (!(r.rx[g] >= 0) || !(r.tx[g] >= 0)) prune = false;
    // The reference looks for a negative remainder on EVERY NIC of the node, picked or not, AFTER the combination's demands were
    // subtracted (`any(x < 0 for y in nic_ttls ...)`, Matcher.py:262-267).  The shipped capacities are never negative; under
    // ENABLE_SHARING a NIC carrying more than its capacity (speed_used above speed * 0.9) makes every combination of non-negative
    // demands fail: the node offers no NIC candidates at all.  A negative (or NaN) demand can lift such a NIC back above zero - those
    // requests take the enumeration as the reference writes it, with the check over all NICs behind every combination.
    if (caps.sh && prune)
        for (uint32_t u = 0; u < U; ++u)
            for (uint32_t k = 0; k < n.nic_cnt[u]; ++k)
                if (caps.free_of(n, u, k, 0) < 0 || caps.free_of(n, u, k, 1) < 0) return false;
    // groups order[0..pos] assigned: does the NIC of the newest one still hold, and its switch?
    auto nic_holds = [&](uint32_t upto, uint32_t u, uint32_t k) {
        double rx = caps.free_of(n, u, k, 0), tx = caps.free_of(n, u, k, 1);
        for (uint32_t q = 0; q <= upto; ++q) {                 // same NUMA node => ascending group index along `order`
            const uint32_t h = order[q];
            if (numa[h] == u && pick[h] == k) { rx = rx - r.rx[h]; tx = tx - r.tx[h]; }
        }
        return !(rx < 0) && !(tx < 0);                         // Matcher.py:267
    };
    auto switch_holds = [&](uint32_t upto, uint32_t sw) {
        uint32_t c = 0;
        for (uint32_t q = 0; q <= upto; ++q) {
            const uint32_t h = order[q];
            if (n.nic_sw[numa[h]][pick[h]] == sw) ++c;
        }
        return c <= wide_sw_free(n, sw);                       // Matcher.py:318-322
    };
    // big requests: NIC (u, k) has an equal twin of lower index that, like itself, no group of order[0..pos) uses
    auto twin_skipped = [&](int pos, uint32_t u, uint32_t k) {
        for (int q = 0; q < pos; ++q)
            if (numa[order[q]] == u && pick[order[q]] == k) return false;                    // in use: its own state
        for (uint32_t k2 = 0; k2 < k; ++k2) {
            if (!caps.same_price(n, u, k, k2) || n.nic_sw[u][k2] != n.nic_sw[u][k]) continue;
            bool used = false;
            for (int q = 0; q < pos && !used; ++q) used = numa[order[q]] == u && pick[order[q]] == k2;
            if (!used) return true;
        }
        return false;
    };
    auto spend = [&]() {
        if (!ns) return true;
        if (ns->left == 0) { ns->exhausted = true; return false; }
        ns->left--;
        return true;
    };
    if (prune) {
        int pos = 0;
        for (;;) {
            const uint32_t g = order[pos];
            if (!spend()) return false;
            const bool ok = !(req_traits<R>::kBig && twin_skipped(pos, numa[g], pick[g])) &&
                            nic_holds((uint32_t)pos, numa[g], pick[g]) && (!pci || switch_holds((uint32_t)pos, n.nic_sw[numa[g]][pick[g]]));
            if (ok) {
                if (pos == (int)G - 1) { for (uint32_t q = 0; q < G; ++q) nic_idx[q] = (int8_t)pick[q]; return true; }
                ++pos;                                         // (the next group starts at its first NIC: its pick is 0)
                continue;
            }
            for (;;) {                                         // next candidate: advance this digit, or back up
                const uint32_t h = order[pos];
                if (pick[h] + 1 < n.nic_cnt[numa[h]]) { pick[h]++; break; }
                pick[h] = 0;
                if (--pos < 0) return false;
            }
        }
    }
    for (;;) {                                                 // the enumeration as the reference writes it
        if (!spend()) return false;
        bool ok = true;
        for (uint32_t q = 0; q < G && ok; ++q) {
            const uint32_t g = order[q];
            ok = nic_holds(G - 1, numa[g], pick[g]) && (!pci || switch_holds(G - 1, n.nic_sw[numa[g]][pick[g]]));
        }
        if (caps.sh)                                           // ... and no NIC of the node, picked or not, is left below zero (Matcher.py:267)
            for (uint32_t u = 0; u < U && ok; ++u)
                for (uint32_t k = 0; k < n.nic_cnt[u] && ok; ++k) ok = nic_holds(G - 1, u, k);
        if (ok) { for (uint32_t q = 0; q < G; ++q) nic_idx[q] = (int8_t)pick[q]; return true; }
        int pos = (int)G - 1;
        while (pos >= 0) {
            const uint32_t h = order[pos];
            if (pick[h] + 1 < n.nic_cnt[numa[h]]) { pick[h]++; break; }
            pick[h] = 0;
            --pos;
        }
        if (pos < 0) return false;
    }
}

// ---- big requests: the NIC stage asked per NUMA node ----------------------------------------------------------------------------
// Whether an assignment passes the NIC stage is, NUMA node by NUMA node, a question about the SET of groups it puts there: the groups
// of different NUMA nodes choose among disjoint NICs, and - unless one PCIe switch carries NICs of two NUMA nodes, which PCI mode
// then has to count jointly - behind disjoint switches.  U^G assignments share at most U * 2^G such sets, and a search over one NUMA
// node's groups is far shorter than one over all of them, so a big request's stage remembers each set's answer (1 KB per pair).
// The answer itself is the search's own: the same depth-first search (wide_nic_choice) over the request cut down to the set's groups,
// in ascending group index - per NIC the same subtractions in the same order, per switch the same counts.
struct NicMemo {
    uint8_t known[kWideU][1 << NHDFIT_BIG_MAX_GROUPS];         // 0 = not asked yet, 1 = no, 2 = yes
    bool separable;
};
template <class R> NHD_HD bool nic_separable(const nhdfit_wide_node& n, const R& r) {
    if (r.map_type != NHDFIT_MAP_PCI) return true;             // NUMA mode applies no switch test (Matcher.py:294-296)
    for (uint32_t u = 0; u < n.numa_nodes && u < (uint32_t)kWideU; ++u)
        for (uint32_t k = 0; k < n.nic_cnt[u]; ++k)
            for (uint32_t u2 = u + 1; u2 < n.numa_nodes && u2 < (uint32_t)kWideU; ++u2)
                for (uint32_t k2 = 0; k2 < n.nic_cnt[u2]; ++k2)
                    if (n.nic_sw[u][k] == n.nic_sw[u2][k2]) return false;
    return true;
}
template <class R> NHD_HD void nic_memo_init(NicMemo& m, const nhdfit_wide_node& n, const R& r) {
    for (int u = 0; u < kWideU; ++u)
        for (int s = 0; s < (1 << NHDFIT_BIG_MAX_GROUPS); ++s) m.known[u][s] = 0;
    m.separable = nic_separable(n, r);
}
// can the NICs of NUMA node u host the groups of the non-empty `set` (bit g = group g)?  The search itself:
template <class R>
NHD_HD bool nic_set_search(const nhdfit_wide_node& n, const R& r, const WideCaps& caps, uint32_t u, uint32_t set, NicSearch* ns) {
    R sub = r;
    uint32_t k = 0, code = 0;
    for (uint32_t g = 0; g < r.n_groups; ++g)
        if (set >> g & 1u) { sub.rx[k] = r.rx[g]; sub.tx[k] = r.tx[g]; ++k; code = code * n.numa_nodes + u; }   // every group of the cut-down request on NUMA node u
    sub.n_groups = k;
    int8_t nic[req_traits<R>::kG];
    return wide_nic_choice(n, sub, caps, code, nic, ns);
}
// ... and through the memo
template <class R>
NHD_HD bool nic_set_ok(NicMemo& m, const nhdfit_wide_node& n, const R& r, const WideCaps& caps, uint32_t u, uint32_t set, NicSearch* ns) {
    if (!set) return true;
    uint8_t& known = m.known[u][set];
    if (known) return known == 2;
    const bool ok = nic_set_search(n, r, caps, u, set, ns);
    if (ns && ns->exhausted) return false;                     // (no answer: nothing is remembered)
    known = ok ? 2 : 1;
    return ok;
}
// the NIC stage of assignment `gcode` through the memo (separable nodes) or by the joint search
template <class R>
NHD_HD bool nic_stage_ok(NicMemo& m, const nhdfit_wide_node& n, const R& r, const WideCaps& caps, uint32_t gcode, NicSearch* ns) {
    if (!m.separable) {
        int8_t nic[req_traits<R>::kG];
        return wide_nic_choice(n, r, caps, gcode, nic, ns);
    }
    uint32_t sets[kWideU] = {0, 0, 0, 0};
    for (uint32_t g = 0; g < r.n_groups; ++g) sets[wide_digit(gcode, r.n_groups, n.numa_nodes, g)] |= 1u << g;
    for (uint32_t u = 0; u < n.numa_nodes; ++u)
        if (!nic_set_ok(m, n, r, caps, u, sets[u], ns)) return false;
    return true;
}
// ... without a memo: the same searches in the same order.  On two NUMA nodes every assignment has its own pair of sets - nothing
// would be remembered twice, so the search steps taken over all assignments are the memo's (wide_map_wave, big_kernel.h: lane = tuple)
template <class R>
NHD_HD bool nic_stage_ok_plain(bool separable, const nhdfit_wide_node& n, const R& r, const WideCaps& caps, uint32_t gcode, NicSearch* ns) {
    if (!separable) {
        int8_t nic[req_traits<R>::kG];
        return wide_nic_choice(n, r, caps, gcode, nic, ns);
    }
    uint32_t sets[kWideU] = {0, 0, 0, 0};
    for (uint32_t g = 0; g < r.n_groups; ++g) sets[wide_digit(gcode, r.n_groups, n.numa_nodes, g)] |= 1u << g;
    for (uint32_t u = 0; u < n.numa_nodes; ++u)
        if (sets[u] && !nic_set_search(n, r, caps, u, sets[u], ns)) return false;
    return true;
}

// scalar predicates (Matcher.py:65-84, 107-111; InitialNodeFilter NHDScheduler.py:235-247 when the request asks for it)
template <class R> NHD_HD bool wide_scalar_ok(const nhdfit_wide_node& n, const R& r, bool busy) {
    if (!req_valid(r) || !wide_shape_ok(n)) return false;
    if (n.flags & NHDFIT_NF_MAINTENANCE) return false;
    if (r.hugepages_gb > n.hp_free) return false;
    if (r.flags & NHDFIT_RF_INITIAL_FILTER)
        if (!(n.flags & NHDFIT_NF_ACTIVE) || !(n.groups & r.groups)) return false;
    if (busy) {
        uint32_t want = 0;
        for (uint32_t g = 0; g < r.n_groups; ++g) want += r.gpus[g];
        if (want) return false;
    }
    return true;
}

// feasible(node, pod): some assignment passes all three stages (the set intersection of Matcher.py:346 is non-empty)
template <class R>
NHD_HD bool wide_fits(const nhdfit_wide_node& n, const R& r, bool busy, const WideCaps& caps, NicSearch* ns = nullptr) {
    if (!wide_scalar_ok(n, r, busy)) return false;
    const WideFree f = wide_free(n);
    const uint32_t G = r.n_groups, nG = wide_ipow(f.U, G);
    int8_t nic[req_traits<R>::kG];
    if constexpr (req_traits<R>::kBig) {
        // A big request walks the assignments as an odometer - the last group's digit turns fastest: ascending codes, i.e.
        // itertools.product order - carrying the per-NUMA-node GPU and core totals along instead of re-deriving every tuple's digits
        // (the stages above, wide_gpu_ok / wide_cpu_ok, asked per code, cost O(G^2) each: at 2^8 .. 4^8 assignments per pair that
        // was the whole pass).  Same integers, same comparisons.  Before any of it: a node whose free GPUs or free cores do not
        // cover the pod's totals passes no assignment at all (most nodes, for a pod of this size).
        constexpr int kG = req_traits<R>::kG;
        const uint32_t U = f.U;
        const uint32_t misc = f.smt ? r.misc_smt : r.misc_nosmt;
        uint32_t dg[kG], dc[kG], digit[kG];
        uint32_t need_g = 0, need_c = misc, have_g = 0, have_c = 0;
        for (uint32_t g = 0; g < G; ++g) {
            dg[g] = r.gpus[g];
            dc[g] = f.smt ? r.cpu_smt[g] : r.cpu_nosmt[g];
            digit[g] = 0;
            need_g += dg[g]; need_c += dc[g];
        }
        for (uint32_t u = 0; u < U; ++u) { have_g += f.g[u]; have_c += f.c[u]; }
        if (need_g > have_g || need_c > have_c) return false;
        uint32_t tg[kWideU] = {need_g, 0, 0, 0}, tc[kWideU] = {need_c - misc, 0, 0, 0};       // code 0: every group on NUMA node 0
        // (on two NUMA nodes every assignment has its own pair of sets: nothing to remember - the searches are asked directly, and the
        // kilobyte of answers is neither cleared nor carried)
        const bool remember = U > 2;
        const bool separable = remember ? true : nic_separable(n, r);
        NicMemo memo;
        if (remember) nic_memo_init(memo, n, r);
        for (uint32_t code = 0; code < nG; ++code) {
            bool ok = true;
            for (uint32_t u = 0; u < U; ++u) ok = ok && tg[u] <= f.g[u];                         // GPU stage, Matcher.py:120-131
            if (ok) {                                                                              // CPU stage: the misc cores on some NUMA node m, Matcher.py:206-216
                bool cpu = false;
                for (uint32_t m = 0; m < U && !cpu; ++m) {
                    bool fit = true;
                    for (uint32_t u = 0; u < U; ++u) fit = fit && tc[u] + (u == m ? misc : 0u) <= f.c[u];
                    cpu = fit;
                }
                ok = cpu;
            }
            if (ok) {
                if (remember ? nic_stage_ok(memo, n, r, caps, code, ns) : nic_stage_ok_plain(separable, n, r, caps, code, ns)) return true;
                if (ns && ns->exhausted) return false;
            }
            for (int g = (int)G - 1; g >= 0; --g) {                                               // next tuple
                const uint32_t u = digit[g];
                tg[u] -= dg[g]; tc[u] -= dc[g];
                if (u + 1 < U) { digit[g] = u + 1; tg[u + 1] += dg[g]; tc[u + 1] += dc[g]; break; }
                digit[g] = 0; tg[0] += dg[g]; tc[0] += dc[g];
            }
        }
        return false;
    }
    for (uint32_t code = 0; code < nG; ++code) {
        if (!wide_gpu_ok(r, f, code)) continue;
        bool cpu = false;
        for (uint32_t m = 0; m < f.U && !cpu; ++m) cpu = wide_cpu_ok(r, f, code * f.U + m);
        if (!cpu) continue;
        if (wide_nic_choice(n, r, caps, code, nic, ns)) return true;
        if (ns && ns->exhausted) return false;
    }
    return false;
}

// ---- CPython sets of int tuples over range(U) (winner_map.h's model, tables in scratch) --------------------------------------
// hash((d0, d1, ..)): Objects/tupleobject.c tuplehash() with hash(int k) == k
NHD_HD uint64_t wide_tuple_hash(uint32_t code, uint32_t len, uint32_t U) {
    uint64_t acc = kXX5;
    for (uint32_t i = 0; i < len; ++i) {
        const uint64_t lane = wide_digit(code, len, U, i);
        acc += lane * kXX2;
        acc = (acc << 31) | (acc >> 33);
        acc *= kXX1;
    }
    acc += (uint64_t)len ^ (kXX5 ^ 3527539ULL);
    if (acc == (uint64_t)-1) return 1546275796ULL;
    return acc;
}

template <class K> struct WideSetT {
    K* key;          // [cap] -1 = unused slot
    int32_t cap, mask, fill;
    uint32_t len, U;       // tuple length, digit base
    bool overflow;         // the table would have outgrown `cap` (cannot happen for the sizes above; reported, never silent)
    const uint64_t* htab;  // optional: hash of every tuple code, computed beforehand (wide_map_wave: lane = tuple); else recomputed per probe
};
template <class K> NHD_HD uint64_t ws_hash(const WideSetT<K>& s, K k) { return s.htab ? s.htab[(uint32_t)k] : wide_tuple_hash((uint32_t)k, s.len, s.U); }
using WideSet = WideSetT<int16_t>;      // keys of an ordinary request's tuples (< 1 024); a big request's reach 4^9: int32 (req_traits<R>::Key)
template <class K> NHD_HD void ws_init(WideSetT<K>& s, K* mem, int32_t cap, uint32_t len, uint32_t U) {
    s.key = mem; s.cap = cap; s.mask = 7; s.fill = 0; s.len = len; s.U = U; s.overflow = false; s.htab = nullptr;
    for (int32_t i = 0; i < 8; ++i) mem[i] = -1;
}
template <class K> NHD_HD void ws_insert_clean(K* key, int32_t mask, K k, uint64_t h) {      // set_insert_clean()
    uint64_t perturb = h;
    uint64_t i = h & (uint64_t)mask;
    for (;;) {
        if (key[i] < 0) break;
        bool found = false;
        if (i + 9 <= (uint64_t)mask)
            for (int j = 1; j <= 9; ++j)
                if (key[i + j] < 0) { i += j; found = true; break; }
        if (found) break;
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & (uint64_t)mask;
    }
    key[i] = k;
}
template <class K> NHD_HD void ws_resize(WideSetT<K>& s, int32_t minused, K* tmp) {                    // set_table_resize()
    int32_t newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    if (newsize > s.cap) { s.overflow = true; return; }
    const int32_t oldmask = s.mask;
    for (int32_t i = 0; i <= oldmask; ++i) tmp[i] = s.key[i];
    for (int32_t i = 0; i < newsize; ++i) s.key[i] = -1;
    s.mask = newsize - 1;
    for (int32_t i = 0; i <= oldmask; ++i)
        if (tmp[i] >= 0) ws_insert_clean(s.key, s.mask, tmp[i], ws_hash(s, tmp[i]));
}
template <class K> NHD_HD bool ws_has(const WideSetT<K>& s, K k) {
    const uint64_t h = ws_hash(s, k);
    uint64_t perturb = h;
    uint64_t i = h & (uint64_t)s.mask;
    for (;;) {
        const int probes = (i + 9 <= (uint64_t)s.mask) ? 9 : 0;
        for (int j = 0; j <= probes; ++j) {
            if (s.key[i + j] < 0) return false;
            if (s.key[i + j] == k) return true;
        }
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & (uint64_t)s.mask;
    }
}
template <class K> NHD_HD void ws_add(WideSetT<K>& s, K k, K* tmp) {                             // set_add_entry()
    if (s.overflow) return;
    const uint64_t h = ws_hash(s, k);
    uint64_t perturb = h;
    uint64_t i = h & (uint64_t)s.mask;
    for (;;) {
        const int probes = (i + 9 <= (uint64_t)s.mask) ? 9 : 0;
        for (int j = 0; j <= probes; ++j) {
            if (s.key[i + j] < 0) {
                s.key[i + j] = k;
                s.fill++;
                if (s.fill * 5 >= s.mask * 3) ws_resize(s, s.fill > 50000 ? s.fill * 2 : s.fill * 4, tmp);   // set_add_entry: used > 50000 ? used*2 : used*4
                return;
            }
            if (s.key[i + j] == k) return;
        }
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & (uint64_t)s.mask;
    }
}
template <class K> NHD_HD int32_t ws_next(const WideSetT<K>& s, int32_t from) {                              // iteration = slot order
    for (int32_t i = from; i <= s.mask; ++i)
        if (s.key[i] >= 0) return i;
    return -1;
}
// out = a & b: iterate the smaller operand (b on ties) in slot order, probe the other (set_intersection())
template <class K> NHD_HD void ws_intersect(const WideSetT<K>& a, const WideSetT<K>& b, WideSetT<K>& out, K* tmp) {
    const WideSetT<K>* probe = &a;
    const WideSetT<K>* iter = &b;
    if (b.fill > a.fill) { probe = &b; iter = &a; }
    for (int32_t i = ws_next(*iter, 0); i >= 0; i = ws_next(*iter, i + 1))
        if (ws_has(*probe, iter->key[i])) ws_add(out, iter->key[i], tmp);
}

// max - min over y in range(U) of tuple.count(y) (GetNumaGroupIdx's node_delta, Matcher.py:428-430)
NHD_HD int wide_spread(uint32_t code, uint32_t G, uint32_t U) {
    int cnt[kWideU] = {0, 0, 0, 0};
    for (uint32_t g = 0; g < G; ++g) cnt[wide_digit(code, G, U, g)]++;
    int mx = cnt[0], mn = cnt[0];
    for (uint32_t u = 1; u < U; ++u) { mx = cnt[u] > mx ? cnt[u] : mx; mn = cnt[u] < mn ? cnt[u] : mn; }
    return mx - mn;
}
template <class K> NHD_HD int32_t wide_pick_gpu_tuple(const WideSetT<K>& s, uint32_t G, uint32_t U) {        // first maximiser in list(set) order
    int32_t best = -1;
    int best_spread = -1;
    for (int32_t i = ws_next(s, 0); i >= 0; i = ws_next(s, i + 1)) {
        const int sp = wide_spread((uint32_t)s.key[i], G, U);
        if (sp > best_spread) { best_spread = sp; best = s.key[i]; }
    }
    return best;
}

// The winner-only tail of FindNode on a wide node (Matcher.py:337-391 + 423-452).  `scratch`: kWideScratchWords int16 (ordinary) / big_scratch_words(U, G) int32 (big).
// Returns 1 = mapped, 0 = the node does not take the pod, -1 = a set outgrew its table (never expected; the caller reports it).
// -2 = the NIC search budget of a big request ran out (reported like -1).
// slots_g / slots_c: table sizes of the sets over G- and (G+1)-tuples (ordinary requests: kWideSetSlotsG / kWideSetSlotsC).
//
// Two parts.  The STAGES answer, per tuple, whether the GPU / NIC / CPU stage lists it (Matcher.py:116-141, 206-220, 242-268 + 294-335):
// independent questions - asked on the spot by one thread (WideStagesScalar), or beforehand by a wavefront with lane = tuple
// (big_kernel.h wide_map_wave, which also hands over every tuple's hash).  The MODEL (wide_map_model) is the serial part: CPython
// sets filled in product order, their intersections, the picks in slot order.
template <bool> struct NicMemoBox { NicMemo m; };
template <> struct NicMemoBox<false> {};
template <class R> struct WideStagesScalar {
    const nhdfit_wide_node& n; const R& r; const WideCaps& caps; const WideFree& f;
    NicMemoBox<req_traits<R>::kBig> memo;
    NicSearch budget;
    NHD_HD WideStagesScalar(const nhdfit_wide_node& n_, const R& r_, const WideCaps& c_, const WideFree& f_)
        : n(n_), r(r_), caps(c_), f(f_), budget{req_traits<R>::kBig ? 8u * NHDFIT_BIG_NIC_BUDGET : 0u, false} {     // (every assignment is searched here, not only up to the first hit)
        if constexpr (req_traits<R>::kBig) nic_memo_init(memo.m, n, r);
    }
    NHD_HD bool gpu(uint32_t code) { return wide_gpu_ok(r, f, code); }
    NHD_HD bool nic(uint32_t code) {
        if constexpr (req_traits<R>::kBig) return nic_stage_ok(memo.m, n, r, caps, code, &budget);
        else { int8_t idx[req_traits<R>::kG]; return wide_nic_choice(n, r, caps, code, idx, nullptr); }
    }
    NHD_HD bool cpu(uint32_t code) { return wide_cpu_ok(r, f, code); }
    NHD_HD bool exhausted() const { return budget.exhausted; }
    NHD_HD const uint64_t* hash_g() const { return nullptr; }
    NHD_HD const uint64_t* hash_c() const { return nullptr; }
};
template <class R, class Stages>
NHD_HD int wide_map_model(const nhdfit_wide_node& n, const R& r, const WideCaps& caps, const WideFree& f, typename req_traits<R>::Key* scratch,
                          typename req_traits<R>::Mapping& out, int32_t kSlotsG, int32_t kSlotsC, Stages& st) {
    using K = typename req_traits<R>::Key;
    const uint32_t G = r.n_groups, U = f.U, nG = wide_ipow(U, G), nC = nG * U;
    K* mem = scratch;
    WideSetT<K> sg, sc, a, b, c, ab, abc;
    ws_init(sg, mem, kSlotsG, G, U); mem += kSlotsG;
    ws_init(a, mem, kSlotsG, G, U); mem += kSlotsG;
    ws_init(b, mem, kSlotsG, G, U); mem += kSlotsG;
    ws_init(c, mem, kSlotsG, G, U); mem += kSlotsG;
    ws_init(ab, mem, kSlotsG, G, U); mem += kSlotsG;
    ws_init(abc, mem, kSlotsG, G, U); mem += kSlotsG;
    ws_init(sc, mem, kSlotsC, G + 1, U); mem += kSlotsC;
    sg.htab = a.htab = b.htab = c.htab = ab.htab = abc.htab = st.hash_g();
    sc.htab = st.hash_c();
    K* tmp = mem;
    // candidate sets, filled in product order
    for (uint32_t code = 0; code < nG; ++code) {
        if (st.gpu(code)) ws_add(sg, (K)code, tmp);
        if (st.nic(code)) ws_add(c, (K)code, tmp);
        if (st.exhausted()) return -2;
    }
    for (uint32_t code = 0; code < nC; ++code)
        if (st.cpu(code)) ws_add(sc, (K)code, tmp);
    if (!sg.fill || !sc.fill || !c.fill) return 0;
    // set(gpu_tuples) & set(cpu_tuples) & set(nic_tuples): set(list) re-inserts in list (= slot) order
    for (int32_t i = ws_next(sg, 0); i >= 0; i = ws_next(sg, i + 1)) ws_add(a, sg.key[i], tmp);
    for (int32_t i = ws_next(sc, 0); i >= 0; i = ws_next(sc, i + 1)) ws_add(b, (K)(sc.key[i] / (K)U), tmp);     // tuple[:-1]
    ws_intersect(a, b, ab, tmp);
    ws_intersect(ab, c, abc, tmp);
    if (sg.overflow || sc.overflow || a.overflow || b.overflow || c.overflow || ab.overflow || abc.overflow) return -1;
    if (!abc.fill) return 0;
    // the GPU list is replaced by the intersection only if that drops something (Matcher.py:363-366)
    const int32_t gcode = abc.fill < sg.fill ? wide_pick_gpu_tuple(abc, G, U) : wide_pick_gpu_tuple(sg, G, U);
    int32_t ccode = -1;                                                              // Matcher.py:441-444
    for (int32_t i = ws_next(sc, 0); i >= 0 && ccode < 0; i = ws_next(sc, i + 1))
        if (sc.key[i] / (K)U == gcode) ccode = sc.key[i];
    if (gcode < 0 || ccode < 0) return 0;
    NicSearch last{req_traits<R>::kBig ? NHDFIT_BIG_NIC_BUDGET : 0u, false};
    if (!wide_nic_choice(n, r, caps, (uint32_t)gcode, out.nic_idx, req_traits<R>::kBig ? &last : nullptr)) return last.exhausted ? -2 : 0;
    for (uint32_t g = 0; g < G; ++g) { out.gpu[g] = (int8_t)wide_digit((uint32_t)gcode, G, U, g); out.nic_numa[g] = out.gpu[g]; }
    for (uint32_t g = 0; g <= G; ++g) out.cpu[g] = (int8_t)wide_digit((uint32_t)ccode, G + 1, U, g);
    out.valid = 1;
    return 1;
}
template <class M> NHD_HD void wide_map_clear(M& out, int kG) {
    for (int g = 0; g < kG; ++g) { out.gpu[g] = out.nic_numa[g] = out.nic_idx[g] = -1; }
    for (int g = 0; g <= kG; ++g) out.cpu[g] = -1;
    out.valid = 0; out.pad[0] = out.pad[1] = 0;
}
template <class R>
NHD_HD int wide_map(const nhdfit_wide_node& n, const R& r, const WideCaps& caps, typename req_traits<R>::Key* scratch,
                    typename req_traits<R>::Mapping& out, int32_t slots_g = kWideSetSlotsG, int32_t slots_c = kWideSetSlotsC) {
    wide_map_clear(out, req_traits<R>::kG);
    if (!req_valid(r) || !wide_shape_ok(n)) return 0;
    const WideFree f = wide_free(n);
    if ((uint64_t)wide_ipow(f.U, r.n_groups) * f.U > (uint64_t)req_traits<R>::kMaxTuples) return -1;
    WideStagesScalar<R> st(n, r, caps, f);
    return wide_map_model(n, r, caps, f, scratch, out, slots_g, slots_c, st);
}

// ---- the commit step (commit_core.h, on the wide record) -----------------------------------------------------------------
// GetFreeCpuBatch(numa, num, smt) (nhd/Node.py:502-519) over socket u's range of the flat bitmaps; masks are relative to the
// socket.  See take_batch (commit_core.h) for the walk, its run-on into the sibling range (`late`) and what "false" means.
NHD_HD bool wide_take_batch(nhdfit_wide_node& n, uint32_t u, uint32_t num, bool smt_requested, uint64_t take[2], uint64_t pair[2], uint64_t late[2]) {
    const bool smt_node = (n.flags & NHDFIT_NF_SMT) != 0;
    const bool pairs = smt_node && smt_requested;
    const uint32_t cpp = n.cores_per_proc, base = u * cpp;
    take[0] = take[1] = pair[0] = pair[1] = late[0] = late[1] = 0;
    uint32_t avail = 0;
    for (uint32_t b = 0; b < cpp; ++b) avail += wide_bit(n.t0, base + b) && wide_bit(n.t1, base + b);
    const uint32_t n_take = pairs ? (num + 1) / 2 : num, n_pair = pairs ? num / 2 : 0;
    const uint32_t n_late = (smt_node && !pairs && num > avail) ? num - avail : 0;
    uint32_t rank = 0, got_take = 0, got_late = 0;
    for (uint32_t b = 0; b < cpp; ++b) {
        const uint32_t c = base + b;
        if (!(wide_bit(n.t0, c) && wide_bit(n.t1, c))) continue;
        if (rank < n_take) { take[b >> 6] |= 1ull << (b & 63); ++got_take; }
        if (rank < n_pair) pair[b >> 6] |= 1ull << (b & 63);
        if (rank < n_late) { late[b >> 6] |= 1ull << (b & 63); ++got_late; }
        ++rank;
    }
    for (uint32_t b = 0; b < cpp; ++b) {
        const uint32_t c = base + b;
        if (take[b >> 6] >> (b & 63) & 1) wide_clear(n.t0, c);
        if (smt_node && ((pair[b >> 6] | late[b >> 6]) >> (b & 63) & 1)) wide_clear(n.t1, c);
    }
    return got_take + got_late == n_take && got_late == n_late;
}

template <class R>
NHD_HD int wide_commit(nhdfit_wide_node& n, const R& r, const typename req_traits<R>::Mapping& m, double busy_time,
                       typename req_traits<R>::WidePlacement& out, nhdfit_wide_share* sh = nullptr) {
    constexpr int kMaxG = req_traits<R>::kG;                                          // (shadows the table pass's constant: this body is per request form)
    const uint32_t G = r.n_groups, U = n.numa_nodes;
    int status = kCommitOk;
    for (int g = 0; g < kMaxG; ++g) {
        for (int w = 0; w < 2; ++w)
            out.proc_take[g][w] = out.proc_pair[g][w] = out.proc_late[g][w] = out.help_take[g][w] = out.help_pair[g][w] = out.help_late[g][w] = 0;
        for (int k = 0; k < NHDFIT_PLACEMENT_GPUS; ++k) out.gpu[g][k] = 0xFF;
        out.numa[g] = -1;
    }
    for (int w = 0; w < 2; ++w) out.misc_take[w] = out.misc_pair[w] = out.misc_late[w] = 0;
    out.numa[kMaxG] = -1;
    out.pad[0] = out.pad[1] = 0;
    n.busy_time = busy_time;                                                          // SetBusy, nhd/Node.py:843-845
    uint32_t claimed[kWideU] = {0, 0, 0, 0};
    for (uint32_t g = 0; g < G; ++g) {
        const uint32_t u = (uint32_t)m.gpu[g] % U;
        out.numa[g] = (int8_t)u;
        if (!wide_take_batch(n, u, r.n_proc[g], (r.smt_bits >> g & 1) != 0, out.proc_take[g], out.proc_pair[g], out.proc_late[g])) status = kCommitWouldRaise;
        const uint32_t nu = (uint32_t)m.nic_numa[g] % U, nk = (uint32_t)m.nic_idx[g] & 15u;
        if (nk >= n.nic_cnt[nu]) { status = kCommitWouldRaise; continue; }            // GetNicObjFromIndex finds nothing (Node.py:657-661)
        const uint32_t sw = n.nic_sw[nu][nk];
        for (uint32_t k = 0; k < r.gpus[g]; ++k) {
            int pick = -1;
            for (int x = 0; x < n.n_gpus && pick < 0; ++x)                            // GetFreePciGpuFromNic, Node.py:648-655
                if ((n.gpu_free >> x & 1) && n.gpu_sw[x] == sw) pick = x;
            if (pick < 0 && r.map_type != NHDFIT_MAP_PCI)
                for (int x = 0; x < n.n_gpus && pick < 0; ++x)                        // GetNextGpuFree, Node.py:495-500
                    if ((n.gpu_free >> x & 1) && n.gpu_numa[x] == u) pick = x;
            if (pick < 0) { status = kCommitWouldRaise; continue; }
            n.gpu_free &= ~(1u << pick);
            if (k < (uint32_t)NHDFIT_PLACEMENT_GPUS) out.gpu[g][k] = (uint8_t)pick;
        }
        if (!wide_take_batch(n, u, r.n_help[g], (r.smt_bits >> (kMaxG + g) & 1) != 0, out.help_take[g], out.help_pair[g], out.help_late[g])) status = kCommitWouldRaise;
        if (r.nic_use >> g & 1) {
            claimed[nu] |= 1u << nk;
            // speed_used[0 / 1] += the RX / TX core's speed (nhd/Node.py:754): a group's rx / tx is its one RX / TX core's speed (the
            // packer turns away groups with several under ENABLE_SHARING), a direction without a core adds 0.0
            if (sh) { sh->used[nu][nk][0] += r.rx[g]; sh->used[nu][nk][1] += r.tx[g]; }
        }
    }
    if (r.hugepages_gb > 0) n.hp_free -= r.hugepages_gb;                              // Node.py:794-796
    const uint32_t mu = (uint32_t)m.cpu[G] % U;
    out.numa[kMaxG] = (int8_t)mu;
    if (!wide_take_batch(n, mu, r.n_misc, r.misc_smt_enabled != 0, out.misc_take, out.misc_pair, out.misc_late)) status = kCommitWouldRaise;   // Node.py:799
    for (uint32_t u = 0; u < U; ++u)                                                  // ClaimPodNICResources, Node.py:644-646; capacity 0 while pods_used > 0 (292)
        for (uint32_t k = 0; k < (uint32_t)NHDFIT_MAX_NICS_PER_NUMA; ++k)
            if (claimed[u] >> k & 1) {
                if (n.nic_pods[u][k] < 127) n.nic_pods[u][k]++;
                n.nic_cls[u][k] = n.nic_pods[u][k] > 0 ? 0 : n.nic_base[u][k];
            }
    out.status = (uint8_t)status;
    return status;
}

}  // namespace nhdfit
