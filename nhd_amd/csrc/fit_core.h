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
// ---- table image of one 64-pod tile (staged in LDS by the fit kernel) ---------------------------
//   Bit-sliced: every table row holds, for each NUMA assignment p, ONE 64-bit word whose bit j says
//   "assignment p of pod j passes this test".  Row = W words (W = 2^maxG of the batch) + 8 B pad.
//     W0[m][smt][c]  c free cores on socket 0:  m=0: sumC(S0) <= c      m=1: sumC(S0)+misc <= c
//     W1[m][smt][c]  c free cores on socket 1:  m=0: sumC(S1) <= c      m=1: sumC(S1)+misc <= c
//                    cpu_ok = (W0[1] & W1[0]) | (W0[0] & W1[1])
//     A[f0][f1]      free GPUs on NUMA 0 / 1:   sumG(S0) <= f0 && sumG(S1) <= f1
//     R0[sig]        NIC signature of NUMA 0:   S0(p) in reach(sig)
//     R1[sig]        NIC signature of NUMA 1:   S1(p) in reach(sig)
//   so  feasible pods of a node = OR over p of (cpu_ok & A & R0 & R1)[p]  -  five/seven 64-bit ANDs per
//   assignment serve all 64 pods at once, and no per-pod "any assignment left?" test is needed.
//   64-bit scalar-predicate rows (bit j = verdict for pod j):
//     HP[k]          k = clamp(free hugepages, -1, hp_max) + 1:  pod valid && hp_req <= free
//     GF[gs]         node-group set id: pod does not filter || sets intersect  (NHDScheduler.py:240)
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
constexpr int kMaxHpRows = 1024;                     // hugepage table rows (larger requests are clamped, see hp_bit)
constexpr double kMinBusySecs = 30.0;                // Node.MIN_BUSY_SECS, nhd/Node.py:107

struct Layout {
    uint32_t fc_dim;     // 1 + max physical cores on one socket anywhere in the cluster (<= 65)
    uint32_t fg_dim;     // 1 + max GPUs installed on one NUMA node anywhere in the cluster (<= 9)
    uint32_t nsig, ngs;  // NIC signatures, node-group sets
    uint32_t hp_rows;    // 2 + largest hugepage request of the staged batch (capped at kMaxHpRows)
    uint32_t W;          // assignments per pod the rows provide for: 2^(largest group count of the staged batch)
    uint32_t row_bytes;  // W * 8 + 8: 8-byte aligned, and 18 r mod 64 banks: 32 different rows never collide
    uint32_t row_w1, row_a, row_r0, row_r1, rows16;   // first row of each assignment table (W0 starts at 0)
    uint32_t off_hp, off_gf;                           // byte offsets of the 64-bit tables
    uint32_t bytes;                                    // image size, multiple of 16
};

NHD_HD Layout make_layout(uint32_t max_cores_per_numa, uint32_t max_gpus_per_numa, uint32_t nsig, uint32_t ngs,
                          uint32_t hp_rows, uint32_t max_groups) {
    Layout l;
    l.W = 1u << max_groups;
    l.row_bytes = l.W * 8 + 8;
    l.fc_dim = max_cores_per_numa + 1;
    l.fg_dim = max_gpus_per_numa + 1;
    l.nsig = nsig;
    l.ngs = ngs;
    l.hp_rows = hp_rows;
    l.row_w1 = 4 * l.fc_dim;                         // W0: rows [m][smt][c] = m*2*fc_dim + smt*fc_dim + c
    l.row_a = 8 * l.fc_dim;
    l.row_r0 = l.row_a + l.fg_dim * l.fg_dim;
    l.row_r1 = l.row_r0 + nsig;
    l.rows16 = l.row_r1 + nsig;
    l.off_hp = l.rows16 * l.row_bytes;
    l.off_gf = l.off_hp + hp_rows * 8;
    l.bytes = (l.off_gf + ngs * 8 + 15u) & ~15u;
    return l;
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
constexpr uint32_t kPodGroupsShift = 4;   // bits 4..6: n_groups (lets the fit kernel skip assignments no pod of a tile has)

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

// ---- 16-bit table entries ---------------------------------------------------------------------
// socket u in {0,1}; c free physical cores; m: 1 = the pod-level misc cores also land on this socket
NHD_HD uint32_t entry_w(const PodSums& s, uint32_t u, bool smt, uint32_t c, uint32_t m) {
    const uint32_t* sum = smt ? s.cpu_smt : s.cpu_nosmt;
    const uint32_t extra = m ? (smt ? s.misc_smt : s.misc_nosmt) : 0;
    uint32_t out = 0;
    for (uint32_t p = 0; p < s.W; ++p)
        if (sum[u ? p : (~p & s.full)] + extra <= c) out |= 1u << p;
    return out;
}

NHD_HD uint32_t entry_a(const PodSums& s, uint32_t f0, uint32_t f1) {
    uint32_t a = 0;
    for (uint32_t p = 0; p < s.W; ++p)
        if (s.gpu[~p & s.full] <= f0 && s.gpu[p] <= f1) a |= 1u << p;
    return a;
}

// ---- NIC reach families -----------------------------------------------------------------------
// A family is a bitmask over group-subsets S (bit S set = "these groups can be hosted together").
// dunion(A,B) = { R|S : R in A, S in B, R&S == 0 }.
// For disjoint R, S the union R|S equals R+S, so "every R of a that is disjoint from S, united with S" is
// (a & disj(S)) << S with disj(S) = the subsets of {0..3} that avoid S: 16 shift-and-mask terms, no inner loop.
static_assert(kMaxG == 4, "the disj() masks below enumerate subsets of four groups");
NHD_HD uint32_t dunion(uint32_t a, uint32_t b, uint32_t W) {
    (void)W;                                   // members of a and b are < W, and so is every R|S
    uint32_t out = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t S = 0; S < (1u << kMaxG); ++S) {
        const uint32_t disj = ((S & 1) ? 0x5555u : 0xFFFFu) & ((S & 2) ? 0x3333u : 0xFFFFu) &
                              ((S & 4) ? 0x0F0Fu : 0xFFFFu) & ((S & 8) ? 0x00FFu : 0xFFFFu);
        if (b >> S & 1) out |= (a & disj) << S;
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

// value of 16-bit row `row` for one pod
NHD_HD uint32_t row16_entry(const Layout& L, const PodSums& s, const SigDict& d, const uint16_t* cover, uint32_t row) {
    if (row < L.row_a) {
        const uint32_t u = row >= L.row_w1, k = u ? row - L.row_w1 : row;      // k = m*2*fc_dim + smt*fc_dim + c
        const uint32_t m = k >= 2 * L.fc_dim, sc = m ? k - 2 * L.fc_dim : k;
        return entry_w(s, u, sc >= L.fc_dim, sc >= L.fc_dim ? sc - L.fc_dim : sc, m);
    }
    if (row < L.row_r0) return entry_a(s, (row - L.row_a) / L.fg_dim, (row - L.row_a) % L.fg_dim);
    if (row < L.row_r1) return entry_r(sig_reach(d, row - L.row_r0, cover, s.W), s.W, 0);
    return entry_r(sig_reach(d, row - L.row_r1, cover, s.W), s.W, 1);
}

// 64-bit rows: one bit per pod of the tile.
// HP row k stands for "free hugepages = k-1"; the last row for "free >= hp_rows-2".  Requests are
// non-negative; requests above the cap are compared against the cap row exactly as hp_req <= free would
// fail for every free < request as long as free is below the cap, and the cap (1022 GiB of 1 GiB pages
// per pod) is documented in DESIGN.md.
NHD_HD bool hp_bit(const PodHeader& h, const Layout& L, uint32_t k) {
    if (!(h.flags & kPodValid)) return false;
    return h.hp_req <= (int32_t)k - 1;                                         // Matcher.py:78
}
NHD_HD bool gf_bit(const PodHeader& h, uint64_t node_groups) {
    return !(h.flags & kPodFilter) || (node_groups & h.groups) != 0;           // NHDScheduler.py:240
}

// ---- node side -----------------------------------------------------------------------------------
struct NodeLane {            // what one lane keeps for its node while it sweeps a tile of pods
    uint32_t off_w0, off_w1; // byte offsets of the node's rows in the tile image (the m=1 rows follow at +w_misc)
    uint32_t w_misc;
    uint32_t off_a;
    uint32_t off_r0n, off_r1n, off_r0p, off_r1p;   // NUMA-mode / PCI-mode NIC rows
    uint32_t off_hp, off_gf;
    uint32_t flags;
    bool     busy;
};

NHD_HD NodeLane node_lane(const nhdfit_plane0& a, const nhdfit_plane1& b, const nhdfit_plane2& c,
                          const nhdfit_plane3& d, const nhdfit_plane4& e, double now, const Layout& L) {
    NodeLane n;
    const uint32_t smt = (c.flags & NHDFIT_NF_SMT) ? L.fc_dim : 0;
    uint32_t c0 = popc64(a.t0[0] & b.t1[0]), c1 = popc64(a.t0[1] & b.t1[1]);       // free physical cores, nhd/Node.py:250-264
    c0 = c0 < L.fc_dim ? c0 : L.fc_dim - 1;
    c1 = c1 < L.fc_dim ? c1 : L.fc_dim - 1;
    n.off_w0 = (smt + c0) * L.row_bytes;
    n.off_w1 = (L.row_w1 + smt + c1) * L.row_bytes;
    n.w_misc = 2 * L.fc_dim * L.row_bytes;
    uint32_t f0 = popc32(c.gpu_free & ~c.gpu_numa1), f1 = popc32(c.gpu_free & c.gpu_numa1);   // nhd/Node.py:456-462
    f0 = f0 < L.fg_dim ? f0 : L.fg_dim - 1;
    f1 = f1 < L.fg_dim ? f1 : L.fg_dim - 1;
    n.off_a = (L.row_a + f0 * L.fg_dim + f1) * L.row_bytes;
    n.off_r0n = (L.row_r0 + d.sig_numa[0]) * L.row_bytes;
    n.off_r1n = (L.row_r1 + d.sig_numa[1]) * L.row_bytes;
    n.off_r0p = (L.row_r0 + d.sig_pci[0]) * L.row_bytes;
    n.off_r1p = (L.row_r1 + d.sig_pci[1]) * L.row_bytes;
    int32_t hp = c.hp_free;
    hp = hp < -1 ? -1 : hp;
    hp = hp > (int32_t)L.hp_rows - 2 ? (int32_t)L.hp_rows - 2 : hp;
    n.off_hp = L.off_hp + (uint32_t)(hp + 1) * 8;
    const uint32_t gs = e.group_set < L.ngs ? e.group_set : 0;
    n.off_gf = L.off_gf + gs * 8;
    n.flags = c.flags;
    n.busy = (now - e.busy_time) < kMinBusySecs;              // Node.IsBusy, nhd/Node.py:847-850
    return n;
}

// ---- node records: node_lane() minus the clock, precomputed -----------------------------------------------
// Everything node_lane derives from the five planes depends only on the mirror and the table layout, not on the
// pod tile or the step: a 32-byte record per node (row offsets in units of 8 bytes - rows are 8-byte aligned and an
// image is far below 512 KB) replaces five 16-byte plane loads and ~100 VALU instructions per (node chunk, pod
// tile) pair by two loads and a few unpacks.  Rebuilt when nodes are uploaded or the layout changes.
struct alignas(16) NodeRec {
    uint16_t off_w0, off_w1, off_a, off_r0n, off_r1n, off_r0p, off_r1p, off_hp;    // first 16 bytes
    uint16_t off_gf, flags;
    uint32_t reserved;
    double busy_time;                                                               // second 16 bytes
};
static_assert(sizeof(NodeRec) == 32, "two 16-byte loads per node");

NHD_HD NodeRec make_node_record(const nhdfit_plane0& a, const nhdfit_plane1& b, const nhdfit_plane2& c,
                                const nhdfit_plane3& d, const nhdfit_plane4& e, const Layout& L) {
    const NodeLane n = node_lane(a, b, c, d, e, 0.0, L);
    NodeRec r;
    r.off_w0 = (uint16_t)(n.off_w0 >> 3); r.off_w1 = (uint16_t)(n.off_w1 >> 3); r.off_a = (uint16_t)(n.off_a >> 3);
    r.off_r0n = (uint16_t)(n.off_r0n >> 3); r.off_r1n = (uint16_t)(n.off_r1n >> 3);
    r.off_r0p = (uint16_t)(n.off_r0p >> 3); r.off_r1p = (uint16_t)(n.off_r1p >> 3);
    r.off_hp = (uint16_t)(n.off_hp >> 3); r.off_gf = (uint16_t)(n.off_gf >> 3);
    r.flags = (uint16_t)n.flags;
    r.reserved = 0;
    r.busy_time = e.busy_time;
    return r;
}

NHD_HD NodeLane node_lane_from_record(const NodeRec& r, double now, const Layout& L) {
    NodeLane n;
    n.off_w0 = (uint32_t)r.off_w0 << 3; n.off_w1 = (uint32_t)r.off_w1 << 3; n.off_a = (uint32_t)r.off_a << 3;
    n.off_r0n = (uint32_t)r.off_r0n << 3; n.off_r1n = (uint32_t)r.off_r1n << 3;
    n.off_r0p = (uint32_t)r.off_r0p << 3; n.off_r1p = (uint32_t)r.off_r1p << 3;
    n.off_hp = (uint32_t)r.off_hp << 3; n.off_gf = (uint32_t)r.off_gf << 3;
    n.w_misc = 2 * L.fc_dim * L.row_bytes;
    n.flags = r.flags;
    n.busy = (now - r.busy_time) < kMinBusySecs;
    return n;
}

NHD_HD uint64_t ld64(const uint8_t* img, uint32_t off) { return *reinterpret_cast<const uint64_t*>(img + off); }

// Scalar predicates of one node against all 64 pods of the tile (bit j = pod j may consider the node).
// m_filt / m_need: tile masks of pods that apply InitialNodeFilter / request GPUs.
NHD_HD uint64_t node_pod_mask(const NodeLane& n, const uint8_t* img, uint64_t m_filt, uint64_t m_need) {
    if (n.flags & NHDFIT_NF_MAINTENANCE) return 0;                       // Matcher.py:71
    uint64_t m = ld64(img, n.off_hp);                                    // Matcher.py:78 (+ request validity)
    m &= ld64(img, n.off_gf);                                            // NHDScheduler.py:240
    if (!(n.flags & NHDFIT_NF_ACTIVE)) m &= ~m_filt;                     // NHDScheduler.py:241-242
    if (n.busy) m &= ~m_need;                                            // Matcher.py:107-111
    return m;
}

// Pods of the tile (bit j) for which SOME NUMA assignment passes the CPU, GPU and NIC tests on this node.
// m_pci: tile mask of pods in PCI mode (they read the PCI-mode NIC rows).
NHD_HD uint64_t node_assignment_mask(const NodeLane& n, const uint8_t* img, uint32_t W, uint64_t m_pci) {
    uint64_t acc = 0;
    for (uint32_t p = 0; p < W; ++p) {
        const uint32_t o = p * 8;
        const uint64_t cpu = (ld64(img, n.off_w0 + n.w_misc + o) & ld64(img, n.off_w1 + o)) |
                             (ld64(img, n.off_w0 + o) & ld64(img, n.off_w1 + n.w_misc + o));
        const uint64_t r0 = (ld64(img, n.off_r0p + o) & m_pci) | (ld64(img, n.off_r0n + o) & ~m_pci);
        const uint64_t r1 = (ld64(img, n.off_r1p + o) & m_pci) | (ld64(img, n.off_r1n + o) & ~m_pci);
        acc |= cpu & ld64(img, n.off_a + o) & r0 & r1;
    }
    return acc;
}

// NIC-feasible assignment bits (bit p) of one (pod, node) pair, for the winner mapping
NHD_HD uint32_t nic_assignment_bits(const uint8_t* img, const Layout& L, uint32_t col, bool pci, const nhdfit_plane3& q3) {
    const uint32_t o0 = (L.row_r0 + (pci ? q3.sig_pci[0] : q3.sig_numa[0])) * L.row_bytes;
    const uint32_t o1 = (L.row_r1 + (pci ? q3.sig_pci[1] : q3.sig_numa[1])) * L.row_bytes;
    uint32_t bits = 0;
    for (uint32_t p = 0; p < L.W; ++p)
        if ((ld64(img, o0 + p * 8) & ld64(img, o1 + p * 8)) >> col & 1) bits |= 1u << p;
    return bits;
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
