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
// Matcher.py:346) plus the scalar predicates (maintenance, hugepages, busy, node groups).
#pragma once
#include <stdint.h>
#include "../../include/nhdfit.h"

#if defined(__HIPCC__)
#define NHD_HD __host__ __device__ inline
#else
#define NHD_HD inline
#endif

namespace nhdfit {

constexpr int kMaxG       = NHDFIT_MAX_GROUPS;
constexpr int kTile       = NHDFIT_TILE;
constexpr int kRowStride  = kTile + 2;               // words per table row: even (8-byte aligned pod pairs for ds_read_b64)
                                                     // and = 2 mod 64 so different rows land on different LDS banks
// rows of a tile's table image: W0[2][fc_dim] W1[2][fc_dim] A[fg_dim][fg_dim] R[nsig]
struct Layout {
    uint32_t fc_dim;     // 1 + max physical cores on one socket anywhere in the cluster (<= 65)
    uint32_t fg_dim;     // 1 + max GPUs installed on one NUMA node anywhere in the cluster (<= 9)
    uint32_t row_w1;     // first row of W1   (W0 starts at row 0)
    uint32_t row_a;      // first row of A    (+ f0 * fg_dim + f1)
    uint32_t row_r;      // first NIC-signature row
    uint32_t rows;
};
NHD_HD Layout make_layout(uint32_t max_cores_per_numa, uint32_t max_gpus_per_numa, uint32_t nsig) {
    Layout l;
    l.fc_dim = max_cores_per_numa + 1;
    l.fg_dim = max_gpus_per_numa + 1;
    l.row_w1 = 2 * l.fc_dim;
    l.row_a = 4 * l.fc_dim;
    l.row_r = l.row_a + l.fg_dim * l.fg_dim;
    l.rows = l.row_r + nsig;
    return l;
}
constexpr double kMinBusySecs = 30.0;                // Node.MIN_BUSY_SECS, nhd/Node.py:107

// Request header consumed in the wave-uniform part of the fit kernel (one per pod).
struct PodHeader {
    int32_t  hp_req;
    uint32_t flags;      // kPod*
    uint64_t groups;
};
constexpr uint32_t kPodValid   = 1u;   // map type NUMA or PCI, 1 <= G <= kMaxG
constexpr uint32_t kPodNeedGpu = 2u;   // sum(gpus) > 0  (== any group has GPUs, Matcher.py:403-407)
constexpr uint32_t kPodPci     = 4u;
constexpr uint32_t kPodFilter  = 8u;   // apply InitialNodeFilter

NHD_HD int popc64(uint64_t x) { return __builtin_popcountll(x); }
NHD_HD int popc32(uint32_t x) { return __builtin_popcount(x); }

// ---- subset sums of the per-group integer demands -------------------------------------------
struct PodSums {
    uint32_t G, W, full;                 // W = 2^G assignments, full = W-1
    uint32_t gpu[1 << kMaxG];            // sum of gpus[i], i in S
    uint32_t cpu_smt[1 << kMaxG];        // sum of cpu_smt[i]
    uint32_t cpu_nosmt[1 << kMaxG];
};

NHD_HD bool req_valid(const nhdfit_req& r) {
    return (r.map_type == NHDFIT_MAP_NUMA || r.map_type == NHDFIT_MAP_PCI) && r.n_groups >= 1 &&
           r.n_groups <= (uint32_t)kMaxG;
}

NHD_HD void pod_sums(const nhdfit_req& r, PodSums& s) {
    s.G = r.n_groups;
    s.W = 1u << s.G;
    s.full = s.W - 1;
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
    }
    return h;
}

// ---- CPU tables: row = smt * fc_dim + free_cores ---------------------------------------------
//   W0[e] = C0 | C0m<<16,  C0 bit p:  sumC(S0(p))        <= f ,  C0m: sumC(S0(p)) + misc <= f
//   W1[e] = C1m | C1<<16,  C1 bit p:  sumC(S1(p))        <= f ,  C1m: sumC(S1(p)) + misc <= f
//   x = W0[e0] & W1[e1]  ->  cpu_ok = (x | x>>16) & 0xFFFF
NHD_HD uint32_t entry_w0(const nhdfit_req& r, const PodSums& s, bool smt, uint32_t f) {
    const uint32_t* sum = smt ? s.cpu_smt : s.cpu_nosmt;
    const uint32_t misc = smt ? r.misc_smt : r.misc_nosmt;
    uint32_t c0 = 0, c0m = 0;
    for (uint32_t p = 0; p < s.W; ++p) {
        const uint32_t d = sum[~p & s.full];
        if (d <= f) c0 |= 1u << p;
        if (d + misc <= f) c0m |= 1u << p;
    }
    return c0 | (c0m << 16);
}

NHD_HD uint32_t entry_w1(const nhdfit_req& r, const PodSums& s, bool smt, uint32_t f) {
    const uint32_t* sum = smt ? s.cpu_smt : s.cpu_nosmt;
    const uint32_t misc = smt ? r.misc_smt : r.misc_nosmt;
    uint32_t c1 = 0, c1m = 0;
    for (uint32_t p = 0; p < s.W; ++p) {
        const uint32_t d = sum[p];
        if (d <= f) c1 |= 1u << p;
        if (d + misc <= f) c1m |= 1u << p;
    }
    return c1m | (c1 << 16);
}

// ---- GPU table: row (f0, f1) = free GPUs on NUMA 0 / NUMA 1;  bit p: both sums fit ---------------
NHD_HD uint32_t entry_a(const PodSums& s, uint32_t f0, uint32_t f1) {
    uint32_t a = 0;
    for (uint32_t p = 0; p < s.W; ++p)
        if (s.gpu[~p & s.full] <= f0 && s.gpu[p] <= f1) a |= 1u << p;
    return a;
}

// ---- NIC reach families -----------------------------------------------------------------------
// A family is a bitmask over group-subsets S (bit S set = "these groups can be hosted together").
// dunion(A,B) = { R|S : R in A, S in B, R&S == 0 }.
NHD_HD uint32_t dunion(uint32_t a, uint32_t b, uint32_t W) {
    uint32_t out = 0;
    for (uint32_t S = 0; S < W; ++S) {
        if (!(b >> S & 1)) continue;
        for (uint32_t R = 0; R < W; ++R)
            if ((a >> R & 1) && !(R & S)) out |= 1u << (R | S);
    }
    return out;
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
NHD_HD void class_cover(const nhdfit_req& r, double cap, uint32_t W, uint32_t G, uint16_t cover[kMaxG + 1]) {
    uint32_t fit = 1;                       // the empty block always "fits"
    for (uint32_t S = 1; S < W; ++S)
        if (block_fits(r, cap, S)) fit |= 1u << S;
    cover[0] = 1;
    for (uint32_t n = 1; n <= (uint32_t)kMaxG; ++n)
        cover[n] = (n <= G) ? (uint16_t)dunion(cover[n - 1], fit, W) : cover[G];
}

NHD_HD uint32_t size_le_mask(uint32_t W, uint32_t limit) {       // subsets with at most `limit` groups
    uint32_t m = 0;
    for (uint32_t S = 0; S < W; ++S)
        if ((uint32_t)popc32(S) <= limit) m |= 1u << S;
    return m;
}

struct SigDict {
    const uint32_t* sig_off;   // [nsig+1] -> pools
    const uint32_t* pool_off;  // [npools+1] -> cc
    const uint8_t*  pool_glimit;
    const nhdfit_cc* cc;
    uint32_t nsig;
};

// reach family of one signature for one pod; cover = [ncls][kMaxG+1]
NHD_HD uint32_t sig_reach(const SigDict& d, uint32_t sig, const uint16_t* cover, uint32_t W) {
    uint32_t reach = 1;
    for (uint32_t pl = d.sig_off[sig]; pl < d.sig_off[sig + 1]; ++pl) {
        uint32_t pool = 1;
        for (uint32_t k = d.pool_off[pl]; k < d.pool_off[pl + 1]; ++k) {
            uint32_t n = d.cc[k].cnt > kMaxG ? kMaxG : d.cc[k].cnt;
            pool = dunion(pool, cover[d.cc[k].cls * (kMaxG + 1) + n], W);
        }
        if (d.pool_glimit[pl] != NHDFIT_GLIMIT_NONE) pool &= size_le_mask(W, d.pool_glimit[pl]);
        reach = dunion(reach, pool, W);
    }
    return reach;
}

// R[sig] = reach | rev_W(reach)<<16, rev_W(x) bit p = x bit (W-1-p) = x bit S0(p)
//   nic_ok = (R[sig0] >> 16) & R[sig1] & 0xFFFF
NHD_HD uint32_t entry_r(uint32_t reach, uint32_t W) {
    uint32_t rev = 0;
    for (uint32_t p = 0; p < W; ++p)
        if (reach >> (W - 1 - p) & 1) rev |= 1u << p;
    return (reach & 0xFFFFu) | (rev << 16);
}

// ---- node side -----------------------------------------------------------------------------------
struct NodeLane {            // what one lane keeps for its node while it sweeps a tile of pods
    uint32_t off_w0, off_w1; // word offsets of the node's table rows (row * kRowStride)
    uint32_t off_a;
    uint32_t off_rn0, off_rn1, off_rp0, off_rp1;
    int32_t  hp_free;
    uint32_t flags;
    uint64_t groups;
    bool     busy;
};

NHD_HD NodeLane node_lane(const nhdfit_plane0& a, const nhdfit_plane1& b, const nhdfit_plane2& c,
                          const nhdfit_plane3& d, const nhdfit_plane4& e, double now, const Layout& L) {
    NodeLane n;
    const uint32_t smt = (c.flags & NHDFIT_NF_SMT) ? L.fc_dim : 0;
    uint32_t c0 = popc64(a.t0[0] & b.t1[0]), c1 = popc64(a.t0[1] & b.t1[1]);       // free physical cores, nhd/Node.py:250-264
    c0 = c0 < L.fc_dim ? c0 : L.fc_dim - 1;
    c1 = c1 < L.fc_dim ? c1 : L.fc_dim - 1;
    n.off_w0 = (smt + c0) * kRowStride;
    n.off_w1 = (L.row_w1 + smt + c1) * kRowStride;
    uint32_t f0 = popc32(c.gpu_free & ~c.gpu_numa1), f1 = popc32(c.gpu_free & c.gpu_numa1);   // nhd/Node.py:456-462
    f0 = f0 < L.fg_dim ? f0 : L.fg_dim - 1;
    f1 = f1 < L.fg_dim ? f1 : L.fg_dim - 1;
    n.off_a = (L.row_a + f0 * L.fg_dim + f1) * kRowStride;
    n.off_rn0 = (L.row_r + d.sig_numa[0]) * kRowStride;
    n.off_rn1 = (L.row_r + d.sig_numa[1]) * kRowStride;
    n.off_rp0 = (L.row_r + d.sig_pci[0]) * kRowStride;
    n.off_rp1 = (L.row_r + d.sig_pci[1]) * kRowStride;
    n.hp_free = c.hp_free;
    n.flags = c.flags;
    n.groups = d.groups;
    n.busy = (now - e.busy_time) < kMinBusySecs;              // Node.IsBusy, nhd/Node.py:847-850
    return n;
}

// One (pod, node) evaluation against the pod's table column `col` (= pod index inside its tile).
// `tab` is the tile's table image: word [row * kRowStride + col].  Conditions on the pod header are
// wave-uniform on the GPU (every lane of a wavefront works on the same pod), conditions on the node
// are folded into one predicate so lanes never diverge.
NHD_HD bool eval_pair(const NodeLane& n, const PodHeader& h, const uint32_t* tab, uint32_t col) {
    bool pass = (h.flags & kPodValid) != 0;
    pass &= !(n.flags & NHDFIT_NF_MAINTENANCE);                              // Matcher.py:71
    pass &= h.hp_req <= n.hp_free;                                           // Matcher.py:78
    if (h.flags & kPodFilter)                                                // NHDScheduler.py:240-242
        pass &= ((n.flags & NHDFIT_NF_ACTIVE) != 0) & ((n.groups & h.groups) != 0);
    const uint32_t x = tab[n.off_w0 + col] & tab[n.off_w1 + col];
    uint32_t ok = (x | (x >> 16)) & 0xFFFFu;
    if (h.flags & kPodNeedGpu) {
        pass &= !n.busy;                                                     // Matcher.py:107-111
        ok &= tab[n.off_a + col];
    }
    uint32_t r0, r1;
    if (h.flags & kPodPci) { r0 = tab[n.off_rp0 + col]; r1 = tab[n.off_rp1 + col]; }
    else                   { r0 = tab[n.off_rn0 + col]; r1 = tab[n.off_rn1 + col]; }
    ok &= (r0 >> 16) & r1;
    return pass & (ok != 0);
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

}  // namespace nhdfit
