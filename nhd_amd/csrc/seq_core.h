// seq_core.h - mode B: sequential-commit semantics of a batch.
//
// The real scheduler commits pod k's winner before it matches pod k+1 (nhd/NHDScheduler.py:289-304, 425-437).
// A commit only ever takes resources away from ONE node, so it can only turn that node's column of the
// snapshot feasibility bitmap from 1 to 0 for later pods.  The resolver therefore walks the pods in order
// over the bitmap the fit kernel produced for the snapshot and re-examines a candidate only if an earlier
// pod of the batch landed on it ("dirty" node); dirty nodes are kept in an overlay (free-core counts, GPU
// mask, per-NIC capacity class, hugepages, busy flag) and evaluated directly - by enumeration over their
// NICs, exactly like the winner mapping does - because their NIC signature may not have a table row.
//
// Count-level restatement of the commit (nhd/Node.py:663-841, 502-519, 644-655): which logical cores are
// picked does not influence any later decision, only how many physical cores disappear per socket; WHICH
// GPU is taken does (per-switch free counts) and is reproduced exactly (first unused GPU on the NIC's PCIe
// switch, else - NUMA mode only - first unused GPU of the group's NUMA node).
#pragma once
#include "winner_map.h"

namespace nhdfit {

struct SeqStatic {               // read-only views of the device mirror
    const nhdfit_plane0* p0;
    const nhdfit_plane1* p1;
    const nhdfit_plane2* p2;
    const nhdfit_plane3* p3;
    const nhdfit_plane4* p4;
    const nhdfit_detail* det;
    const double* caps;
    uint32_t n;
    uint64_t global_base;
    double now;
};

struct OverlayNode {             // a node some earlier pod of the batch was committed to
    uint32_t node;               // local index
    int32_t  free_c[2];
    uint32_t gpu_free;
    int32_t  hp_free;
    uint32_t busy;
    uint32_t unclaimed;          // NICs whose capacity class is not 0 (not yet claimed by a pod)
    nhdfit_detail d;             // nic_cls / sw_free are kept current
};

NHD_HD void overlay_init(OverlayNode& o, const SeqStatic& s, uint32_t node) {
    o.node = node;
    o.free_c[0] = popc64(s.p0[node].t0[0] & s.p1[node].t1[0]);
    o.free_c[1] = popc64(s.p0[node].t0[1] & s.p1[node].t1[1]);
    o.gpu_free = s.p2[node].gpu_free;
    o.hp_free = s.p2[node].hp_free;
    o.busy = (s.now - s.p4[node].busy_time) < kMinBusySecs;
    o.d = s.det[node];
    o.unclaimed = 0;
    for (int u = 0; u < 2; ++u)
        for (int k = 0; k < o.d.nic_cnt[u]; ++k) o.unclaimed += o.d.nic_cls[u][k] != 0;
}

NHD_HD WinnerState overlay_state(const OverlayNode& o, const SeqStatic& s) {
    WinnerState w;
    const nhdfit_plane2& q2 = s.p2[o.node];
    w.U = o.d.numa_nodes;
    w.smt = (q2.flags & NHDFIT_NF_SMT) != 0;
    w.free_c[0] = o.free_c[0];
    w.free_c[1] = o.free_c[1];
    w.free_g[0] = popc32(o.gpu_free & ~q2.gpu_numa1);
    w.free_g[1] = popc32(o.gpu_free & q2.gpu_numa1);
    w.d = &o.d;
    w.caps = s.caps;
    return w;
}

// NIC-feasible assignments (tuple codes) of a node evaluated directly from its per-NIC records
NHD_HD uint32_t nic_codes_direct(const nhdfit_req& r, const WinnerState& w) {
    const int G = (int)r.n_groups;
    const uint32_t nG = ipow(w.U, G);
    uint32_t bits = 0;
    int8_t scratch_idx[kMaxG];
    for (uint32_t code = 0; code < nG; ++code)
        if (first_nic_choice(r, w, code, r.map_type == NHDFIT_MAP_PCI, scratch_idx)) bits |= 1u << code;
    return bits;
}

// FindNode's verdict for one (pod, dirty node) pair, by enumeration (nhd/Matcher.py:65-391 for one node).
NHD_HD bool eval_direct(const OverlayNode& o, const SeqStatic& s, const nhdfit_req& r, const PodHeader& h,
                        uint32_t* nic_codes_out) {
    const nhdfit_plane2& q2 = s.p2[o.node];
    if (!(h.flags & kPodValid)) return false;
    if (q2.flags & NHDFIT_NF_MAINTENANCE) return false;
    if (h.hp_req > o.hp_free) return false;
    if (h.flags & kPodFilter)
        if (!(q2.flags & NHDFIT_NF_ACTIVE) || !(s.p3[o.node].groups & h.groups)) return false;
    if ((h.flags & kPodNeedGpu) && o.busy) return false;
    const WinnerState w = overlay_state(o, s);
    const int G = (int)r.n_groups, U = w.U;
    const uint32_t nG = ipow(U, G);
    const uint32_t codes = nic_codes_direct(r, w);
    *nic_codes_out = codes;
    for (uint32_t code = 0; code < nG; ++code) {
        if (!(codes >> code & 1)) continue;
        uint32_t g0 = 0, g1 = 0, c0 = 0, c1 = 0;
        for (int g = 0; g < G; ++g) {
            const uint32_t d = w.smt ? r.cpu_smt[g] : r.cpu_nosmt[g];
            if (tup_digit(code, G, U, g)) { g1 += r.gpus[g]; c1 += d; } else { g0 += r.gpus[g]; c0 += d; }
        }
        if (g0 > (uint32_t)w.free_g[0] || g1 > (uint32_t)w.free_g[1]) continue;
        const uint32_t misc = w.smt ? r.misc_smt : r.misc_nosmt;
        const bool m0 = (int32_t)(c0 + misc) <= w.free_c[0] && (int32_t)c1 <= w.free_c[1];
        const bool m1 = U > 1 && (int32_t)c0 <= w.free_c[0] && (int32_t)(c1 + misc) <= w.free_c[1];
        if (m0 || m1) return true;
    }
    return false;
}

// Cheap necessary conditions (no enumeration): prunes the typical "an earlier pod of the batch filled this
// node" case before eval_direct is tried.  Never rejects a feasible pair.
NHD_HD bool quick_maybe(const OverlayNode& o, const SeqStatic& s, const nhdfit_req& r, const PodHeader& h) {
    const nhdfit_plane2& q2 = s.p2[o.node];
    if (!(h.flags & kPodValid) || (q2.flags & NHDFIT_NF_MAINTENANCE) || h.hp_req > o.hp_free) return false;
    if ((h.flags & kPodNeedGpu) && o.busy) return false;
    const bool smt = (q2.flags & NHDFIT_NF_SMT) != 0;
    uint32_t cores = smt ? r.misc_smt : r.misc_nosmt, gpus = 0, need_bw = 0, biggest = 0;
    for (uint32_t g = 0; g < r.n_groups; ++g) {
        const uint32_t d = smt ? r.cpu_smt[g] : r.cpu_nosmt[g];
        cores += d;
        biggest = d > biggest ? d : biggest;
        gpus += r.gpus[g];
        need_bw |= (r.rx[g] > 0 || r.tx[g] > 0) ? 1u : 0u;
    }
    const int32_t f0 = o.free_c[0], f1 = o.free_c[1];
    if ((int32_t)cores > f0 + f1 || (int32_t)biggest > (f0 > f1 ? f0 : f1)) return false;
    if ((int32_t)gpus > popc32(o.gpu_free)) return false;
    if (need_bw && !o.unclaimed) return false;               // every NIC left has capacity 0
    return true;
}

NHD_HD uint32_t phys_cores(uint32_t n, bool smt_requested, bool smt_node) {
    return (smt_node && smt_requested) ? (n + 1) / 2 : n;     // GetFreeCpuBatch, nhd/Node.py:502-519
}

// Applies one placement to the overlay.  Returns 0, or 1 where the reference's commit would raise.
NHD_HD int apply_commit(OverlayNode& o, const SeqStatic& s, const nhdfit_req& r, const nhdfit_mapping& m) {
    const nhdfit_plane2& q2 = s.p2[o.node];
    const bool smt = (q2.flags & NHDFIT_NF_SMT) != 0;
    const int G = (int)r.n_groups;
    int bad = 0;
    uint32_t claimed0 = 0, claimed1 = 0;                       // NIC ordinals to claim, per NUMA node
    o.busy = 1;                                                // SetBusy, nhd/Node.py:843-845
    for (int g = 0; g < G; ++g) {
        const int u = m.gpu[g];
        o.free_c[u] -= (int32_t)(phys_cores(r.n_proc[g], r.smt_bits >> g & 1, smt) +
                                 phys_cores(r.n_help[g], r.smt_bits >> (4 + g) & 1, smt));
        const uint32_t sw = o.d.nic_sw[(int)m.nic_numa[g]][(int)m.nic_idx[g]];
        for (uint32_t k = 0; k < r.gpus[g]; ++k) {
            int pickg = -1;
            for (int x = 0; x < o.d.n_gpus && pickg < 0; ++x)                      // Node.py:648-655
                if ((o.gpu_free >> x & 1) && o.d.gpu_sw[x] == sw) pickg = x;
            if (pickg < 0 && r.map_type != NHDFIT_MAP_PCI)
                for (int x = 0; x < o.d.n_gpus && pickg < 0; ++x)                  // Node.py:495-500
                    if ((o.gpu_free >> x & 1) && (int)(q2.gpu_numa1 >> x & 1) == u) pickg = x;
            if (pickg < 0) { bad = 1; continue; }
            o.gpu_free &= ~(1u << pickg);
            if (o.d.sw_free[o.d.gpu_sw[pickg]]) o.d.sw_free[o.d.gpu_sw[pickg]]--;
        }
        if (r.nic_use >> g & 1) {
            if (m.nic_numa[g]) claimed1 |= 1u << m.nic_idx[g]; else claimed0 |= 1u << m.nic_idx[g];
        }
    }
    if (r.hugepages_gb > 0) o.hp_free -= r.hugepages_gb;                           // Node.py:794-796
    o.free_c[(int)m.cpu[G]] -= (int32_t)phys_cores(r.n_misc, r.misc_smt_enabled, smt);   // Node.py:799
    if (o.free_c[0] < 0 || o.free_c[1] < 0) { bad = 1; o.free_c[0] = o.free_c[0] < 0 ? 0 : o.free_c[0]; o.free_c[1] = o.free_c[1] < 0 ? 0 : o.free_c[1]; }
    for (int k = 0; k < NHDFIT_MAX_NICS_PER_NUMA; ++k) {                            // ClaimPodNICResources, Node.py:644-646
        if (claimed0 >> k & 1) o.d.nic_cls[0][k] = 0;                               // capacity class 0 = 0.0
        if (claimed1 >> k & 1) o.d.nic_cls[1][k] = 0;
    }
    o.unclaimed = 0;
    for (int u = 0; u < 2; ++u)
        for (int k = 0; k < o.d.nic_cnt[u]; ++k) o.unclaimed += o.d.nic_cls[u][k] != 0;
    return bad;
}

// Result of one pod in mode B
struct SeqResult {
    int64_t node;                // global node index or -1
    nhdfit_mapping map;
    int32_t status;              // 0 ok / not placed, 1 = the reference's commit step would have failed
};

// Is the snapshot candidate `nd` still feasible for this pod after the commits so far?
struct StillFeasible {
    const SeqStatic& s;
    const nhdfit_req& r;
    const PodHeader& h;
    const int32_t* slot_of;
    const OverlayNode* overlay;
    NHD_HD bool operator()(int64_t nd) const {
        const int32_t slot = slot_of[nd];
        if (slot < 0) return true;                            // untouched: the snapshot verdict stands
        uint32_t codes;
        return quick_maybe(overlay[slot], s, r, h) && eval_direct(overlay[slot], s, r, h, &codes);
    }
};

// One pod of the sequential batch.
//   score_a / map_a : the snapshot (mode A) result of this pod
//   scan.find_first(pref, from, ok) : first node >= from whose snapshot feasibility bit is set for this pod
//                           (and, if `pref`, that has no GPU installed) and for which ok(node) holds, or -1
//   slot_of[node]   : overlay slot of a dirty node, -1 if no earlier pod of the batch touched it
// Selection rule = SelectNode (nhd/Matcher.py:393-421) over the up-to-date candidate list.
template <class Scan, bool SMALL_ONLY = false>
NHD_HD void resolve_pod(const SeqStatic& s, const nhdfit_req& r, const PodHeader& h, unsigned long long score_a,
                        const nhdfit_mapping& map_a, Scan& scan, int32_t* slot_of, OverlayNode* overlay,
                        uint32_t* n_overlay, SeqResult& out) {
    out.node = -1;
    out.status = 0;
    out.map = nhdfit_mapping{};
    if (!score_a) return;                                     // infeasible everywhere even before any commit
    const int64_t winner_a = (int64_t)(NHDFIT_SCORE_INDEX(score_a) - s.global_base);
    const StillFeasible ok{s, r, h, slot_of, overlay};
    int64_t nd = -1;
    if ((score_a >> 63) != 0) nd = scan.find_first(true, winner_a, ok);    // GPU-less nodes first for a GPU-less pod
    if (nd < 0) nd = scan.find_first(false, (score_a >> 63) ? 0 : winner_a, ok);
    if (nd < 0) return;

    const int32_t slot = slot_of[nd];
    nhdfit_mapping m = nhdfit_mapping{};
    if (slot < 0 && nd == winner_a) {
        m = map_a;                                            // untouched snapshot winner: its mapping is already known
    } else {
        OverlayNode tmp;
        if (slot < 0) overlay_init(tmp, s, (uint32_t)nd);
        const OverlayNode& o = slot < 0 ? tmp : overlay[slot];
        // SMALL_ONLY: the batch holds no pod with more than 3 groups -> only the register-resident set model
        // is instantiated and the resolver kernel needs no scratch memory
        if (SMALL_ONLY) map_winner_t<SmallOps>(r, overlay_state(o, s), nic_codes_direct(r, overlay_state(o, s)), m);
        else map_winner(r, overlay_state(o, s), nic_codes_direct(r, overlay_state(o, s)), m);
    }
    int32_t use = slot;
    if (use < 0) {
        use = (int32_t)(*n_overlay);
        *n_overlay = (uint32_t)use + 1;
        overlay_init(overlay[use], s, (uint32_t)nd);
        slot_of[nd] = use;
    }
    out.status = m.valid ? apply_commit(overlay[use], s, r, m) : 1;
    out.map = m;
    out.node = (int64_t)s.global_base + nd;
}

}  // namespace nhdfit
