// fit_core.h - the arithmetic of the node filter-and-score path, written once and compiled both
// into the gfx950 kernels (nhdfit.hip) and, for CPU-only logic tests, into tests/harness.
//
// What the reference does per (pod, node) by enumerating NUMA assignments and NIC choices in Python
// (nhd/Matcher.py:86-391) is re-expressed as set algebra over the 2^G possible NUMA assignments of
// a pod's G proc groups on a 2-socket node:
//
//   assignment p : bit i of p = NUMA node of group i      (p in [0, 2^G))
//   S1(p) = p  = groups on NUMA 1,   S0(p) = ~p & full = groups on NUMA 0
//
//   GPU  ok(p) = sumG(S0) <= freeG[0]  and sumG(S1) <= freeG[1]                    Matcher.py:120-131
//   CPU  ok(p) = exists m in {0,1}: sumC(Su) + [m==u]*misc <= freeC[u] for u=0,1   Matcher.py:206-216
//   NIC  ok(p) = S0 in reach[0] and S1 in reach[1]                                  Matcher.py:239-268, 294-335
//
// where reach[u] is the family of group-sets the NICs of NUMA u can host (every group gets exactly
// one NIC of its NUMA node, several groups may share a NIC while the sequential f64 subtraction
// cap - rx_i1 - rx_i2 ... stays >= 0 for rx and tx; in PCI mode at most free_gpus(switch) groups per
// PCIe switch).  reach[u] depends on the node only through a small interned "NIC signature", so it
// is tabulated per pod by the request-digest kernel; the P x N kernel is then integer table
// look-ups only and every f64 operation is performed exactly as the reference performs it.
//
// A node is feasible for a pod iff some p passes all three (this is the set intersection of
// Matcher.py:346) plus the scalar predicates (maintenance, hugepages, busy, node groups), which are
// tabulated too: one 64-bit word per node-side value holds the verdict for all 64 pods of a tile.
//
// ---- table image of one 64-pod tile -----------------------------------------------------------------
//   Bit-sliced: every table row holds, for each NUMA assignment p, ONE 64-bit word whose bit j says
//   "assignment p of pod j passes this test".  A tile's rows have W = 2^(largest group count among its
//   pods) words (tiles are staged sorted by group count, so most tiles are narrow).
//
//   COLD section - global memory only; read for winners (mapping) and for committed nodes (mode B):
//     A0[f], A1[f]   f free GPUs on NUMA 0 / 1:      sumG(S0) <= f   /   sumG(S1) <= f
//     R0[sig]        NIC signature of NUMA 0:         S0(p) in reach(sig)
//     R1[sig]        NIC signature of NUMA 1:         S1(p) in reach(sig)
//   HOT section - staged in LDS by the fit role, rows 16-byte aligned for ds_read_b128:
//     X[class]       class = interned (NUMA u, f_u, sigNUMA_u, sigPCI_u) of the mirror's nodes (interned on the
//                    device, k_xkeys):  A_u[f] & (pod j in PCI mode ? R_u[sigPCI] : R_u[sigNUMA])
//     WC[u][smt][c]  record of two rows, c free physical cores on socket u:
//                      m=0: sumC(Su) <= c       m=1: sumC(Su) + misc <= c
//                    cpu_ok = (WC0[m=1] & WC1[m=0]) | (WC0[m=0] & WC1[m=1])
//     GX[g]          64-bit rows (bit j = pod j): row 0 = never (maintenance / lanes past the end), row
//                    1 + 2*gs + active: pod does not apply InitialNodeFilter || (node groups intersect && active)
//     HP[k]          k = clamp(free hugepages, -1, hp_max) + 1:  pod valid && hp_req <= free
//   feasible pods of a node = OR_p (cpu_ok & X0 & X1)[p]  &  GX & HP & (busy ? ~pods_needing_gpus : all)
//   - six 16-byte row fetches and 16 bit operations per pair of assignments serve all 64 pods at once.
#pragma once
#include <stdint.h>
#include "../../include/nhdfit.h"

#if defined(__HIPCC__)
#define NHD_HD __host__ __device__ __forceinline__
#else
#define NHD_HD inline
#endif

namespace nhdfit {

constexpr int kMaxG      = NHDFIT_MAX_GROUPS;
constexpr int kTile      = NHDFIT_TILE;
constexpr int kMaxHpRows = 1024;                     // hugepage table rows (larger requests are rejected at staging)
constexpr int kWClasses  = 4;                        // W = 2, 4, 8, 16
constexpr double kMinBusySecs = 30.0;                // Node.MIN_BUSY_SECS, nhd/Node.py:107
constexpr uint32_t kMinXCap = 32;                    // X rows are provisioned in powers of two (records stay valid while classes are appended)

NHD_HD uint32_t align16(uint32_t x) { return (x + 15u) & ~15u; }
// Row strides: multiples of 16 bytes (ds_read_b128) and an odd multiple of 16 for every W, so that the 16 lanes a
// b128 access is serviced for hit 16 different bank groups when their rows lie within a window of 16 rows.
NHD_HD uint32_t wc_stride_of(uint32_t W) { return 2 * W * 8 + 16; }
NHD_HD uint32_t x_stride_of(uint32_t W) { return W == 2 ? 16u : W * 8 + 16; }

struct Layout {
    uint32_t W;          // assignments per row: 2, 4, 8 or 16
    uint32_t fc_dim;     // 1 + max physical cores on one socket anywhere in the cluster (<= 65)
    uint32_t fg_dim;     // 1 + max GPUs installed on one NUMA node anywhere in the cluster (<= 9)
    uint32_t nsig, ngs;  // NIC signatures, node-group sets
    uint32_t hp_rows;    // 2 + largest hugepage request of the staged batch
    uint32_t x_cap;      // provisioned X rows (power of two >= interned classes)
    uint32_t row;        // W * 8: bytes of an unpadded row (cold section, m=1 row of a WC record)
    uint32_t wc_stride, x_stride;
    // cold section (byte offsets from the image start)
    uint32_t off_a0, off_a1, off_r0, off_r1;
    uint32_t off_hot;    // start of the hot section (multiple of 16)
    // hot section (byte offsets from off_hot): WC, GX, then X, then HP.  Everything up to the end of X depends on the
    // dictionary and the provisioned X rows only (node records hold these offsets); HP depends on the staged batch.
    // A fit block stages the section in LDS; when the node classes of a large heterogeneous cluster outgrow LDS it
    // stages a prefix (plus HP) and the X rows beyond it are read from global memory (L2).
    uint32_t hot_wc0, hot_wc1, hot_gx, hot_x, hot_hp;
    uint32_t hot_bytes;  // multiple of 16
    uint32_t bytes;      // whole image, multiple of 16
};

NHD_HD uint32_t x_capacity(uint32_t nx) {
    uint32_t c = kMinXCap;
    while (c < nx) c <<= 1;
    return c;
}

NHD_HD Layout make_layout(uint32_t W, uint32_t max_cores_per_numa, uint32_t max_gpus_per_numa, uint32_t nsig, uint32_t ngs,
                          uint32_t hp_rows, uint32_t x_cap) {
    Layout l;
    l.W = W;
    l.fc_dim = max_cores_per_numa + 1;
    l.fg_dim = max_gpus_per_numa + 1;
    l.nsig = nsig;
    l.ngs = ngs;
    l.hp_rows = hp_rows;
    l.x_cap = x_cap;
    l.row = W * 8;
    l.wc_stride = wc_stride_of(W);
    l.x_stride = x_stride_of(W);
    l.off_a0 = 0;
    l.off_a1 = l.off_a0 + l.fg_dim * l.row;
    l.off_r0 = l.off_a1 + l.fg_dim * l.row;
    l.off_r1 = l.off_r0 + nsig * l.row;
    l.off_hot = align16(l.off_r1 + nsig * l.row);
    l.hot_wc0 = 0;
    l.hot_wc1 = l.hot_wc0 + 2 * l.fc_dim * l.wc_stride;          // records [smt][c]
    l.hot_gx = l.hot_wc1 + 2 * l.fc_dim * l.wc_stride;
    l.hot_x = l.hot_gx + align16((1 + 2 * ngs) * 8);
    l.hot_hp = l.hot_x + x_cap * l.x_stride;
    l.hot_bytes = l.hot_hp + align16(hp_rows * 8);
    l.bytes = l.off_hot + l.hot_bytes;
    return l;
}

NHD_HD uint32_t wclass_of(uint32_t max_groups) {          // tile class 0..3 <-> W = 2 << class
    return max_groups <= 1 ? 0u : max_groups == 2 ? 1u : max_groups == 3 ? 2u : 3u;
}

// Request header (one per pod), kept for the lane-as-pod view of the fit kernel.
struct PodHeader {
    int32_t  hp_req;
    uint32_t flags;      // kPod*
    uint64_t groups;
};
constexpr uint32_t kPodValid   = 1u;   // map type NUMA or PCI, 1 <= G <= kMaxG
constexpr uint32_t kPodNeedGpu = 2u;   // sum(gpus) > 0  (== any group has GPUs, Matcher.py:403-407)
constexpr uint32_t kPodPci     = 4u;
constexpr uint32_t kPodFilter  = 8u;   // apply InitialNodeFilter
constexpr uint32_t kPodGroupsShift = 4;   // bits 4..6: n_groups

NHD_HD int popc64(uint64_t x) { return __builtin_popcountll(x); }
NHD_HD int popc32(uint32_t x) { return __builtin_popcount(x); }

// ---- subset sums of the per-group integer demands -------------------------------------------
struct PodSums {
    uint32_t G, W, full;                 // W = 2^G assignments, full = W-1
    uint32_t misc_smt, misc_nosmt;
    uint32_t gpu[1 << kMaxG];            // sum of gpus[i], i in S
    uint32_t cpu_smt[1 << kMaxG];        // sum of cpu_smt[i]
    uint32_t cpu_nosmt[1 << kMaxG];
};

NHD_HD bool req_valid(const nhdfit_req& r) {
    return (r.map_type == NHDFIT_MAP_NUMA || r.map_type == NHDFIT_MAP_PCI) && r.n_groups >= 1 &&
           r.n_groups <= (uint32_t)kMaxG;
}

// ---- the two request forms (include/nhdfit.h): nhdfit_req - the table-driven pass, G <= 4 - and nhdfit_big_req - 5..8
// processing groups, answered by the general path only (wide_core.h).  The general path is written once over both.
template <class R> struct req_traits;
template <> struct req_traits<nhdfit_req> {
    static constexpr int kG = NHDFIT_MAX_GROUPS;
    static constexpr bool kBig = false;
    using Key = int16_t;                                  // tuple codes of the general path's set model (wide_core.h): U^(G+1) <= 4^5
    static constexpr int kMaxTuples = 1024;
    using Mapping = nhdfit_mapping;
    using WidePlacement = nhdfit_wide_placement;
};
template <> struct req_traits<nhdfit_big_req> {
    static constexpr int kG = NHDFIT_BIG_MAX_GROUPS;
    static constexpr bool kBig = true;
    using Key = int32_t;                                  // ... <= 4^9
    static constexpr int kMaxTuples = NHDFIT_BIG_MAX_TUPLES;
    using Mapping = nhdfit_big_mapping;
    using WidePlacement = nhdfit_big_placement;
};
static_assert(sizeof(nhdfit_big_req) == 256 && sizeof(nhdfit_big_mapping) == 36 && sizeof(nhdfit_big_placement) == 904, "record sizes of include/nhdfit.h");
NHD_HD bool req_valid(const nhdfit_big_req& r) {
    return (r.map_type == NHDFIT_MAP_NUMA || r.map_type == NHDFIT_MAP_PCI) && r.n_groups >= 1 &&
           r.n_groups <= (uint32_t)NHDFIT_BIG_MAX_GROUPS;
}

NHD_HD void pod_sums(const nhdfit_req& r, PodSums& s) {
    s.G = r.n_groups;
    s.W = 1u << s.G;
    s.full = s.W - 1;
    s.misc_smt = r.misc_smt;
    s.misc_nosmt = r.misc_nosmt;
    for (uint32_t S = 0; S < s.W; ++S) {
        uint32_t g = 0, a = 0, b = 0;
        for (uint32_t i = 0; i < s.G; ++i)
            if (S >> i & 1) { g += r.gpus[i]; a += r.cpu_smt[i]; b += r.cpu_nosmt[i]; }
        s.gpu[S] = g; s.cpu_smt[S] = a; s.cpu_nosmt[S] = b;
    }
}

NHD_HD PodHeader pod_header(const nhdfit_req& r) {
    PodHeader h;
    h.hp_req = r.hugepages_gb;
    h.groups = r.groups;
    h.flags = 0;
    if (req_valid(r)) {
        h.flags |= kPodValid;
        uint32_t g = 0;
        for (uint32_t i = 0; i < r.n_groups; ++i) g += r.gpus[i];
        if (g) h.flags |= kPodNeedGpu;
        if (r.map_type == NHDFIT_MAP_PCI) h.flags |= kPodPci;
        if (r.flags & NHDFIT_RF_INITIAL_FILTER) h.flags |= kPodFilter;
        h.flags |= r.n_groups << kPodGroupsShift;
    }
    return h;
}

// ---- 16-bit table entries (bit p = assignment p of this pod passes) ----------------------------
// socket u in {0,1}; c free physical cores; m: 1 = the pod-level misc cores also land on this socket
NHD_HD uint32_t entry_w(const PodSums& s, uint32_t u, bool smt, uint32_t c, uint32_t m) {
    const uint32_t* sum = smt ? s.cpu_smt : s.cpu_nosmt;
    const uint32_t extra = m ? (smt ? s.misc_smt : s.misc_nosmt) : 0;
    uint32_t out = 0;
    for (uint32_t p = 0; p < s.W; ++p)
        if (sum[u ? p : (~p & s.full)] + extra <= c) out |= 1u << p;
    return out;
}

// f free GPUs on NUMA u: the groups assignment p puts there ask for no more
NHD_HD uint32_t entry_a(const PodSums& s, uint32_t u, uint32_t f) {
    uint32_t a = 0;
    for (uint32_t p = 0; p < s.W; ++p)
        if (s.gpu[u ? p : (~p & s.full)] <= f) a |= 1u << p;
    return a;
}

// ---- NIC reach families -----------------------------------------------------------------------
// A family is a bitmask over group-subsets S (bit S set = "these groups can be hosted together").
// dunion(A,B) = { R|S : R in A, S in B, R&S == 0 }.
// For disjoint R, S the union R|S equals R+S, so "every R of a that is disjoint from S, united with S" is
// (a & disj(S)) << S with disj(S) = the subsets of {0..3} that avoid S: 16 shift-and-mask terms, no inner loop.
static_assert(kMaxG == 4, "the disj() masks below enumerate subsets of four groups");
template <uint32_t TERMS>
NHD_HD uint32_t dunion_n(uint32_t a, uint32_t b) {       // members of b below TERMS only
    uint32_t out = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t S = 0; S < TERMS; ++S) {
        const uint32_t disj = ((S & 1) ? 0x5555u : 0xFFFFu) & ((S & 2) ? 0x3333u : 0xFFFFu) &
                              ((S & 4) ? 0x0F0Fu : 0xFFFFu) & ((S & 8) ? 0x00FFu : 0xFFFFu);
        if (b >> S & 1) out |= (a & disj) << S;
    }
    return out;
}
NHD_HD uint32_t dunion(uint32_t a, uint32_t b, uint32_t W) {
    (void)W;                                   // members of a and b are < W, and so is every R|S
    return dunion_n<1u << kMaxG>(a, b);
}

// Block S of groups sharing one NIC of capacity `cap`: the reference subtracts each group's rx / tx
// from the NIC's remaining [cap, cap] in group order and rejects if anything ends below zero
// (Matcher.py:261-267).  Same operations, same order, IEEE binary64, no contraction possible.
NHD_HD bool block_fits(const nhdfit_req& r, double cap, uint32_t S) {
    double rx = cap, tx = cap;
    for (uint32_t i = 0; i < r.n_groups; ++i)
        if (S >> i & 1) { rx = rx - r.rx[i]; tx = tx - r.tx[i]; }
    return !(rx < 0) && !(tx < 0);
}

// cover[n] = group-sets that n NICs of this capacity can host (n = 0..G)
// WW: compile-time bound >= W for the unions (the device instantiates the row width of the pod's tile: a two-group tile
// pays 4 terms per union, not 16)
template <uint32_t WW>
NHD_HD void class_cover_w(const nhdfit_req& r, double cap, uint32_t W, uint32_t G, uint16_t cover[kMaxG + 1]) {
    uint32_t fit = 1;                       // the empty block always "fits"
    for (uint32_t S = 1; S < W; ++S)
        if (block_fits(r, cap, S)) fit |= 1u << S;
    cover[0] = 1;
    for (uint32_t n = 1; n <= (uint32_t)kMaxG; ++n)
        cover[n] = (n <= G) ? (uint16_t)dunion_n<WW>(cover[n - 1], fit) : cover[G];
}
NHD_HD void class_cover(const nhdfit_req& r, double cap, uint32_t W, uint32_t G, uint16_t cover[kMaxG + 1]) {
    class_cover_w<1u << kMaxG>(r, cap, W, G, cover);
}

NHD_HD uint32_t size_le_mask(uint32_t W, uint32_t limit) {       // subsets of {0..3} below W with at most `limit` groups
    const uint32_t by_size = limit == 0 ? 0x0001u : limit == 1 ? 0x0117u : limit == 2 ? 0x177Fu : limit == 3 ? 0x7FFFu : 0xFFFFu;
    return by_size & ((1u << W) - 1u);
}

struct SigDict {
    const uint32_t* sig_off;   // [nsig+1] -> pools
    const uint32_t* pool_off;  // [npools+1] -> cc
    const uint8_t*  pool_glimit;
    const nhdfit_cc* cc;
    uint32_t nsig;
};

// reach family of one signature for one pod; cover = [ncls][kMaxG+1]
template <uint32_t WW>
NHD_HD uint32_t sig_reach_w(const SigDict& d, uint32_t sig, const uint16_t* cover, uint32_t W) {
    uint32_t reach = 1;
    for (uint32_t pl = d.sig_off[sig]; pl < d.sig_off[sig + 1]; ++pl) {
        uint32_t pool = 1;
        for (uint32_t k = d.pool_off[pl]; k < d.pool_off[pl + 1]; ++k) {
            uint32_t n = d.cc[k].cnt > kMaxG ? kMaxG : d.cc[k].cnt;
            pool = dunion_n<WW>(pool, cover[d.cc[k].cls * (kMaxG + 1) + n]);
        }
        if (d.pool_glimit[pl] != NHDFIT_GLIMIT_NONE) pool &= size_le_mask(W, d.pool_glimit[pl]);
        reach = dunion_n<WW>(reach, pool);
    }
    return reach;
}
NHD_HD uint32_t sig_reach(const SigDict& d, uint32_t sig, const uint16_t* cover, uint32_t W) {
    return sig_reach_w<1u << kMaxG>(d, sig, cover, W);
}

// R1[sig] bit p = reach bit S1(p) = reach bit p;  R0[sig] bit p = reach bit S0(p) = reach bit (W-1-p)
NHD_HD uint32_t entry_r(uint32_t reach, uint32_t W, uint32_t u) {
    if (u) return reach & 0xFFFFu;
    uint32_t r = reach & 0xFFFFu;                       // reverse the low 16 bits, then keep the top W of them
    r = (r & 0x5555u) << 1 | (r >> 1 & 0x5555u);
    r = (r & 0x3333u) << 2 | (r >> 2 & 0x3333u);
    r = (r & 0x0F0Fu) << 4 | (r >> 4 & 0x0F0Fu);
    r = (r & 0x00FFu) << 8 | (r >> 8 & 0x00FFu);
    return r >> (16 - W);
}

// 64-bit rows: one bit per pod of the tile.
// HP row k stands for "free hugepages = k-1"; the last row for "free >= hp_rows-2" (no pod of the batch asks for more).
NHD_HD bool hp_bit(const PodHeader& h, uint32_t k) {
    if (!(h.flags & kPodValid)) return false;
    return h.hp_req <= (int32_t)k - 1;                                         // Matcher.py:78
}
// GX row g: 0 = never; 1 + 2*gs + active
NHD_HD bool gx_bit(const PodHeader& h, uint32_t g, const uint64_t* group_sets) {
    if (g == 0) return false;
    const uint32_t gs = (g - 1) >> 1, active = (g - 1) & 1;
    if (!(h.flags & kPodFilter)) return true;                                  // the caller filtered already
    return active && (group_sets[gs] & h.groups) != 0;                         // NHDScheduler.py:240-242
}

// ---- node side -----------------------------------------------------------------------------------
struct NodeIdx {            // what the five planes of a node say about which table rows apply to it
    uint32_t w0, w1;         // WC record index smt * fc_dim + free physical cores, sockets 0 / 1
    uint32_t f0, f1;         // free GPUs per NUMA node (clamped to the table)
    uint32_t hp;             // HP row before the clamp to the batch's hp_rows (< kMaxHpRows)
    uint32_t gx;             // GX row
    uint32_t nogpu;          // no GPU installed (SelectNode's preference, Matcher.py:401-413)
};

NHD_HD NodeIdx node_index(const nhdfit_plane0& a, const nhdfit_plane1& b, const nhdfit_plane2& c, const nhdfit_plane4& e,
                          uint32_t fc_dim, uint32_t fg_dim, uint32_t ngs) {
    NodeIdx n;
    const uint32_t smt = (c.flags & NHDFIT_NF_SMT) ? fc_dim : 0;
    uint32_t c0 = popc64(a.t0[0] & b.t1[0]), c1 = popc64(a.t0[1] & b.t1[1]);       // free physical cores, nhd/Node.py:250-264
    c0 = c0 < fc_dim ? c0 : fc_dim - 1;
    c1 = c1 < fc_dim ? c1 : fc_dim - 1;
    n.w0 = smt + c0;
    n.w1 = smt + c1;
    uint32_t f0 = popc32(c.gpu_free & ~c.gpu_numa1), f1 = popc32(c.gpu_free & c.gpu_numa1);   // nhd/Node.py:456-462
    n.f0 = f0 < fg_dim ? f0 : fg_dim - 1;
    n.f1 = f1 < fg_dim ? f1 : fg_dim - 1;
    int32_t hp = c.hp_free;
    hp = hp < -1 ? -1 : hp;
    hp = hp > kMaxHpRows - 2 ? kMaxHpRows - 2 : hp;
    n.hp = (uint32_t)(hp + 1);
    const uint32_t gs = e.group_set < ngs ? e.group_set : 0;
    n.gx = (c.flags & NHDFIT_NF_MAINTENANCE) ? 0u : 1u + 2u * gs + ((c.flags & NHDFIT_NF_ACTIVE) ? 1u : 0u);   // Matcher.py:71
    n.nogpu = (c.flags & NHDFIT_NF_HAS_GPU) ? 0u : 1u;
    return n;
}

// class key of one NUMA node of a node: everything the GPU and NIC tests of that NUMA node depend on
NHD_HD uint64_t xkey(uint32_t u, uint32_t f, uint32_t sig_numa, uint32_t sig_pci) {
    return (1ull << 63) | ((uint64_t)u << 62) | ((uint64_t)f << 32) | ((uint64_t)sig_numa << 16) | (uint64_t)sig_pci;
}
NHD_HD uint32_t xkey_u(uint64_t k) { return (uint32_t)(k >> 62) & 1u; }
NHD_HD uint32_t xkey_f(uint64_t k) { return (uint32_t)(k >> 32) & 0xFFu; }
NHD_HD uint32_t xkey_sig_numa(uint64_t k) { return (uint32_t)(k >> 16) & 0xFFFFu; }
NHD_HD uint32_t xkey_sig_pci(uint64_t k) { return (uint32_t)k & 0xFFFFu; }

// The 16-byte node record the fit role streams (one array per row width W): hot-section offsets in units of
// 8 bytes, so that a node costs two loads (record + busy time) and one shift per row address.  The last field carries
// the free cores per socket and the SMT flag unscaled: the pair form of the sweep (below) indexes the table it derives in LDS
// with them.
// LANE ORDER (round 5).  The 64 records of a chunk need not sit in node order: the verdict word is stored by node index and a
// winner carries its index, so which lane of the wavefront works on which node of the chunk is free - and it decides how the
// lanes' row fetches meet in the LDS banks (chunk_lane_order below).  A record therefore names its node's position in the
// chunk (`pos`, the top six bits of `hp`); position = lane is the order k_xrecords writes and every consumer accepts.
struct alignas(16) NodeRec {
    uint16_t w0, w1, x0, x1;          // WC records of socket 0 / 1, X rows of NUMA 0 / 1   (offset / 8)
    uint16_t gx;                      // GX row (offset / 8)
    uint16_t hp;                      // bits 0..9: HP row INDEX (clamped to the batch's rows in the kernel); bits 10..15: the node's position in its chunk
    uint16_t flags;                   // bit 0: kRecNoGpu; bits 1..15 (round 6): the node's row of the pair table C for the dimension the records were
                                      // last written for (pair_c_row; 0 = not written for any) - the pipelined sweep reads its C row's address
                                      // with one AND and one shift instead of ten instructions of min / multiply-add on `cc`, when the launch's
                                      // dimension is that one (FitArgs::crow_ok); `cc` stays what every other reader uses
    uint16_t cc;                      // free physical cores socket 0 | socket 1 << 7 | SMT << 14
};
static_assert(sizeof(NodeRec) == 16, "one 16-byte load per node");
static_assert(kMaxHpRows <= 1024, "NodeRec::hp keeps ten bits for the row index");
constexpr uint16_t kRecNoGpu = 1;
NHD_HD uint32_t rec_hp(const NodeRec& r) { return r.hp & 1023u; }
NHD_HD uint32_t rec_pos(const NodeRec& r) { return r.hp >> 10; }

NHD_HD uint32_t pair_c_row(uint32_t cc, uint32_t D);
NHD_HD NodeRec make_record(const NodeIdx& n, uint32_t x0, uint32_t x1, const Layout& L, uint32_t pos, uint32_t crow_D = 0) {
    NodeRec r;
    r.w0 = (uint16_t)((L.hot_wc0 + n.w0 * L.wc_stride) >> 3);
    r.w1 = (uint16_t)((L.hot_wc1 + n.w1 * L.wc_stride) >> 3);
    r.x0 = (uint16_t)((L.hot_x + x0 * L.x_stride) >> 3);
    r.x1 = (uint16_t)((L.hot_x + x1 * L.x_stride) >> 3);
    r.gx = (uint16_t)((L.hot_gx + n.gx * 8) >> 3);
    r.hp = (uint16_t)(n.hp | (pos & 63u) << 10);
    r.flags = (uint16_t)(n.nogpu ? kRecNoGpu : 0);
    const uint32_t smt = n.w0 >= L.fc_dim ? 1u : 0u;                       // node_index: w = smt * fc_dim + free cores
    r.cc = (uint16_t)((n.w0 - smt * L.fc_dim) | (n.w1 - smt * L.fc_dim) << 7 | smt << 14);
    if (crow_D) r.flags = (uint16_t)(r.flags | pair_c_row(r.cc, crow_D) << 1);           // (2 D^2 <= 2 * 65^2 rows: 14 bits)
    return r;
}
NHD_HD NodeRec dead_record(const Layout& L, uint32_t pos) {      // lanes past the end of the mirror: GX row 0 = never
    NodeRec r;
    r.w0 = (uint16_t)(L.hot_wc0 >> 3); r.w1 = (uint16_t)(L.hot_wc1 >> 3);
    r.x0 = r.x1 = (uint16_t)(L.hot_x >> 3);
    r.gx = (uint16_t)(L.hot_gx >> 3);
    r.hp = (uint16_t)((pos & 63u) << 10); r.flags = 0; r.cc = 0;
    return r;
}

// ---- pair rows ---------------------------------------------------------------------------------------------------------
// The sweep of a node fetches, per pair of assignments, four CPU rows (both sockets, m = 0 / 1) and two class rows.  What
// the CPU rows are ANDed into depends on the node through (SMT, free cores 0, free cores 1) only, so a fit block can tabulate
// the product once, in LDS, when it stages the tile:
//     C[smt][c0][c1]  = (WC0[smt][c0].m1 & WC1[smt][c1].m0) | (WC0[smt][c0].m0 & WC1[smt][c1].m1)         one fetch for four
// C would be 2 x 65 x 65 rows over all free-core counts; but a CPU row only says "demand <= c", so every row at or above
// the tile's largest demand (all groups on one socket plus the misc cores, the larger of the SMT / non-SMT counts) is the
// same row: with D = 1 + that demand, counts are clamped to D - 1 and the table is 2 x D x D rows (two-group tiles of
// BASELINE config 4: D = 24, 37 KB).  The table is stored as W / 2 PLANES of 16-byte pieces, piece q of row r at
// (q * rows + r) * 16 (round 5): rows of W * 8 = 32 bytes put two neighbouring rows' pieces 32 bytes apart - the 16 lanes a
// ds_read_b128 is serviced for then share 8 of the 16 bank groups and every fetch of a four-assignment tile cost twice its
// cycles (16.3 LDS cycles per wavefront fetch against 8.2 on config 4's cluster, tools/lds_bank_model.py).  (A table of the
// products of two class rows, XX[k0][k1], was built and measured in round 4 and is gone: its row index put 32 rows on one
// bank group - 13.6 cycles per fetch against 4 + 4 for the two class rows it replaced.)  The launch decides per
// row width whether the table fits its LDS budget (nhdfit.hip refresh_layouts); the verdicts are the same bits either way
// (host twin: node_word_pair == node_word_hot on every node of every CPU test).
NHD_HD uint32_t req_max_demand(const nhdfit_req& r) {
    uint32_t a = r.misc_smt, b = r.misc_nosmt;
    for (uint32_t i = 0; i < r.n_groups && i < (uint32_t)kMaxG; ++i) { a += r.cpu_smt[i]; b += r.cpu_nosmt[i]; }
    return a > b ? a : b;
}
NHD_HD uint32_t pair_dim(uint32_t max_demand, uint32_t fc_dim) { return max_demand + 1 < fc_dim ? max_demand + 1 : fc_dim; }
NHD_HD uint32_t pair_c_row(uint32_t cc, uint32_t D) {              // row of C for a record's `cc` field
    const uint32_t c0 = cc & 127u, c1 = (cc >> 7) & 127u, smt = cc >> 14;
    return ((smt * D + (c0 < D ? c0 : D - 1)) * D + (c1 < D ? c1 : D - 1));
}

// Node.IsBusy (nhd/Node.py:847-850) is `(now - busy_time) < 30.0` in binary64.  fl(now - t) never increases with t,
// so the busy nodes are exactly those with busy_time >= busy_threshold(now): one comparison per node instead of a
// subtraction and a comparison, and still the reference's own arithmetic (the threshold is found with it).
inline double busy_threshold(double now) {
    double b = now - kMinBusySecs;
    auto up = [](double x) { return __builtin_nextafter(x, __builtin_inf()); };
    auto down = [](double x) { return __builtin_nextafter(x, -__builtin_inf()); };
    for (int i = 0; i < 64 && !((now - b) < kMinBusySecs); ++i) b = up(b);
    for (int i = 0; i < 64 && (now - down(b)) < kMinBusySecs; ++i) b = down(b);
    return b;
}

NHD_HD uint64_t ld64(const uint8_t* img, uint32_t off) { return *reinterpret_cast<const uint64_t*>(img + off); }

// Feasible pods (bit j) of one node from the HOT section of a tile image - the fit role's arithmetic, restated
// with plain loads (host twin, tests).  Wt = the tile's row width, m_need = pods of the tile that request GPUs.
NHD_HD uint64_t node_word_hot(const uint8_t* hot, const Layout& L, const NodeRec& r, bool busy, uint64_t m_need) {
    const uint32_t w0 = (uint32_t)r.w0 << 3, w1 = (uint32_t)r.w1 << 3, x0 = (uint32_t)r.x0 << 3, x1 = (uint32_t)r.x1 << 3;
    uint64_t acc = 0;
    for (uint32_t p = 0; p < L.W; ++p) {
        const uint32_t o = p * 8;
        const uint64_t cpu = (ld64(hot, w0 + L.row + o) & ld64(hot, w1 + o)) | (ld64(hot, w0 + o) & ld64(hot, w1 + L.row + o));
        acc |= cpu & ld64(hot, x0 + o) & ld64(hot, x1 + o);
    }
    const uint32_t hp = rec_hp(r) < L.hp_rows ? rec_hp(r) : L.hp_rows - 1;
    uint64_t pred = ld64(hot, (uint32_t)r.gx << 3) & ld64(hot, L.hot_hp + hp * 8);
    if (busy) pred &= ~m_need;                                           // Matcher.py:107-111
    return acc & pred;
}

// The pair form of the same verdict (what the fit role computes when its block tabulated C): the row is derived here on the
// fly from the hot section, addressed through the record's unscaled fields exactly as the kernel does.
NHD_HD uint64_t node_word_pair(const uint8_t* hot, const Layout& L, const NodeRec& r, uint32_t D, bool busy, uint64_t m_need) {
    const uint32_t row = pair_c_row(r.cc, D), smt = r.cc >> 14, c0 = (row / D) % D, c1 = row % D;
    const uint32_t w0 = L.hot_wc0 + (smt * L.fc_dim + c0) * L.wc_stride, w1 = L.hot_wc1 + (smt * L.fc_dim + c1) * L.wc_stride;
    const uint32_t x0 = (uint32_t)r.x0 << 3, x1 = (uint32_t)r.x1 << 3;
    uint64_t acc = 0;
    for (uint32_t p = 0; p < L.W; ++p) {
        const uint32_t o = p * 8;
        const uint64_t cpu = (ld64(hot, w0 + L.row + o) & ld64(hot, w1 + o)) | (ld64(hot, w0 + o) & ld64(hot, w1 + L.row + o));
        acc |= cpu & ld64(hot, x0 + o) & ld64(hot, x1 + o);
    }
    const uint32_t hp = rec_hp(r) < L.hp_rows ? rec_hp(r) : L.hp_rows - 1;
    uint64_t pred = ld64(hot, (uint32_t)r.gx << 3) & ld64(hot, L.hot_hp + hp * 8);
    if (busy) pred &= ~m_need;
    return acc & pred;
}

// The same verdict from the COLD rows (A / R per signature) and the class-independent hot rows (WC, GX, HP): for
// nodes whose (f, signature) class has no X row yet - a node a pod of the running batch was just committed to.
// one assignment word p of the W the verdict ORs together / the node-wide predicates - split out so that the sequential
// kernel can give every p a lane
NHD_HD uint64_t node_term_cold(const uint8_t* img, const Layout& L, const NodeIdx& n, const nhdfit_plane3& q3, uint64_t m_pci, uint32_t p) {
    const uint8_t* hot = img + L.off_hot;
    const uint32_t w0 = L.hot_wc0 + n.w0 * L.wc_stride, w1 = L.hot_wc1 + n.w1 * L.wc_stride;
    const uint32_t o = p * 8;
    const uint64_t cpu = (ld64(hot, w0 + L.row + o) & ld64(hot, w1 + o)) | (ld64(hot, w0 + o) & ld64(hot, w1 + L.row + o));
    const uint64_t r0 = (ld64(img, L.off_r0 + q3.sig_pci[0] * L.row + o) & m_pci) | (ld64(img, L.off_r0 + q3.sig_numa[0] * L.row + o) & ~m_pci);
    const uint64_t r1 = (ld64(img, L.off_r1 + q3.sig_pci[1] * L.row + o) & m_pci) | (ld64(img, L.off_r1 + q3.sig_numa[1] * L.row + o) & ~m_pci);
    return cpu & ld64(img, L.off_a0 + n.f0 * L.row + o) & ld64(img, L.off_a1 + n.f1 * L.row + o) & r0 & r1;
}
NHD_HD uint64_t node_pred_cold(const uint8_t* img, const Layout& L, const NodeIdx& n, bool busy, uint64_t m_need) {
    const uint8_t* hot = img + L.off_hot;
    const uint32_t hp = n.hp < L.hp_rows ? n.hp : L.hp_rows - 1;
    uint64_t pred = ld64(hot, L.hot_gx + n.gx * 8) & ld64(hot, L.hot_hp + hp * 8);
    if (busy) pred &= ~m_need;
    return pred;
}
NHD_HD uint64_t node_word_cold(const uint8_t* img, const Layout& L, const NodeIdx& n, const nhdfit_plane3& q3, bool busy,
                               uint64_t m_need, uint64_t m_pci) {
    uint64_t acc = 0;
    for (uint32_t p = 0; p < L.W; ++p) acc |= node_term_cold(img, L, n, q3, m_pci, p);
    return acc & node_pred_cold(img, L, n, busy, m_need);
}

// NIC-feasible assignment bits (bit p) of one (pod, node) pair, for the winner mapping (cold R rows).
// ColdView: the four layout words that addressing needs (kernel arguments carry these instead of whole Layouts).
struct ColdView { uint32_t off_r0, off_r1, row, W; };
NHD_HD ColdView cold_view(const Layout& L) { return ColdView{L.off_r0, L.off_r1, L.row, L.W}; }
// (every word of both rows is requested before the first is looked at: with the row width as a run-time loop bound and a branch per
//  assignment the 2 W loads were W dependent round trips to L2 - 8 us of a drain launch's 34 for a three-group tile, round 6)
template <uint32_t W>
NHD_HD uint32_t nic_assignment_bits_w(const uint8_t* img, uint32_t o0, uint32_t o1, uint32_t col) {
    uint64_t x[W];
#pragma unroll
    for (uint32_t p = 0; p < W; ++p) x[p] = ld64(img, o0 + p * 8) & ld64(img, o1 + p * 8);
    uint32_t bits = 0;
#pragma unroll
    for (uint32_t p = 0; p < W; ++p) bits |= (uint32_t)(x[p] >> col & 1) << p;
    return bits;
}
NHD_HD uint32_t nic_assignment_bits(const uint8_t* img, const ColdView& L, uint32_t col, bool pci, const nhdfit_plane3& q3) {
    const uint32_t o0 = L.off_r0 + (pci ? q3.sig_pci[0] : q3.sig_numa[0]) * L.row;
    const uint32_t o1 = L.off_r1 + (pci ? q3.sig_pci[1] : q3.sig_numa[1]) * L.row;
    switch (L.W) {
        case 2: return nic_assignment_bits_w<2>(img, o0, o1, col);
        case 4: return nic_assignment_bits_w<4>(img, o0, o1, col);
        case 8: return nic_assignment_bits_w<8>(img, o0, o1, col);
        default: break;
    }
    uint32_t bits = 0;
    for (uint32_t p = 0; p < L.W; ++p)
        if ((ld64(img, o0 + p * 8) & ld64(img, o1 + p * 8)) >> col & 1) bits |= 1u << p;
    return bits;
}
NHD_HD uint32_t nic_assignment_bits(const uint8_t* img, const Layout& L, uint32_t col, bool pci, const nhdfit_plane3& q3) {
    return nic_assignment_bits(img, cold_view(L), col, pci, q3);
}

// ---- one pod against the nodes directly --------------------------------------------------------------
// For a LONE pod (the scheduler's pod-at-a-time FindNode) the bit-sliced tile image is overhead: 63 of its 64 columns are
// empty and every row is a ballot.  The same verdict comes from the pod's own masks over its 2^G assignments (bit p =
// assignment p passes) - the 16-bit entries the digest would have bit-sliced - looked up per node: node_word_cold for one pod.
//   a0 / a1 [fg_dim]            entry_a(u, f)
//   w0 / w1 [2 * fc_dim][2]     entry_w(u, smt, c, m) at ((smt * fc_dim + c) * 2 + m): NodeIdx::w0 / w1 index the record
//   r0 / r1 [nsig]              entry_r(reach(sig), u)
struct LoneMasks {
    const uint16_t* a0; const uint16_t* a1;
    const uint16_t* w0; const uint16_t* w1;
    const uint16_t* r0; const uint16_t* r1;
};
NHD_HD bool lone_pod_fits(const LoneMasks& t, const PodHeader& h, const NodeIdx& n, const nhdfit_plane3& q3, bool busy,
                          const uint64_t* group_sets) {
    const bool pci = (h.flags & kPodPci) != 0;
    const uint32_t s0 = pci ? q3.sig_pci[0] : q3.sig_numa[0], s1 = pci ? q3.sig_pci[1] : q3.sig_numa[1];
    const uint32_t cpu = ((uint32_t)t.w0[n.w0 * 2 + 1] & t.w1[n.w1 * 2]) | ((uint32_t)t.w0[n.w0 * 2] & t.w1[n.w1 * 2 + 1]);
    if (!(cpu & t.a0[n.f0] & t.a1[n.f1] & t.r0[s0] & t.r1[s1])) return false;
    if (!gx_bit(h, n.gx, group_sets) || !hp_bit(h, n.hp)) return false;          // (hp_bit: the row clamp of a batch changes nothing for its own pods)
    return !(busy && (h.flags & kPodNeedGpu));                                   // Matcher.py:107-111
}
// NIC-feasible assignment bits of the pod on a node (nic_assignment_bits for the lone pod)
NHD_HD uint32_t lone_nic_bits(const LoneMasks& t, bool pci, const nhdfit_plane3& q3) {
    return (uint32_t)t.r0[pci ? q3.sig_pci[0] : q3.sig_numa[0]] & t.r1[pci ? q3.sig_pci[1] : q3.sig_numa[1]];
}
// reach family of one signature from the dictionary's 16-bit stream (DictView::flat: [0, nsig] word offset of each
// signature's record behind the table; record = { #pools, per pool: glimit << 8 | #cc, then #cc x (cls << 8 | cnt) }) -
// sig_reach_w with every lane on a signature of its own; cover = [ncls][kMaxG+1]
NHD_HD uint32_t sig_reach_flat(const uint16_t* flat, uint32_t nsig, uint32_t sig, const uint16_t* cover, uint32_t W) {
    uint32_t at = nsig + 1 + flat[sig];
    const uint32_t npools = flat[at++];
    uint32_t reach = 1;
    for (uint32_t pl = 0; pl < npools; ++pl) {
        const uint32_t head = flat[at++], ncc = head & 0xFFu, glimit = head >> 8;
        uint32_t pool = 1;
        for (uint32_t k = 0; k < ncc; ++k) {
            const uint32_t e = flat[at++], cnt = e & 0xFFu, cls = e >> 8;
            pool = dunion_n<1u << kMaxG>(pool, cover[cls * (kMaxG + 1) + (cnt > (uint32_t)kMaxG ? (uint32_t)kMaxG : cnt)]);
        }
        if (glimit != NHDFIT_GLIMIT_NONE) pool &= size_le_mask(W, glimit);
        reach = dunion_n<1u << kMaxG>(reach, pool);
    }
    return reach;
}

// ---- selection (Matcher.py:393-421) ----------------------------------------------------------
// word = feasibility of 64 consecutive nodes for one pod, nogpu = nodes with no GPU installed.
NHD_HD uint64_t chunk_score(uint64_t word, uint64_t nogpu, bool pod_needs_gpu, uint64_t first_global_index) {
    if (!word) return 0;
    const uint64_t pref = pod_needs_gpu ? 0 : (word & nogpu);
    const uint64_t pick = pref ? pref : word;
    const uint64_t idx = first_global_index + (uint64_t)__builtin_ctzll(pick);
    return (pref ? (1ull << 63) : 0ull) | (0x7FFFFFFFFFFFFFFFull - idx);
}
NHD_HD uint64_t score_of(bool pref, uint64_t global_index) {
    return (pref ? (1ull << 63) : 0ull) | (0x7FFFFFFFFFFFFFFFull - global_index);
}

}  // namespace nhdfit
