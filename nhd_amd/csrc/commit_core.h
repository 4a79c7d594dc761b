// commit_core.h - the commit step on the packed node state.
//
// After FindNode has picked a node and a mapping, the scheduler turns the mapping into physical ids and marks the
// resources used (nhd/NHDScheduler.py:289-304):
//     SetBusy()                                   nhd/Node.py:843-845
//     SetPhysicalIdsFromMapping(mapping, top)      nhd/Node.py:663-841  (GetFreeCpuBatch 502-519, GetNicObjFromIndex
//                                                  657-661, GetFreePciGpuFromNic 648-655, GetNextGpuFree 495-500)
//     ClaimPodNICResources(nidx)                   nhd/Node.py:644-646
// The same on the five planes + detail record of the device mirror, written once for the gfx950 kernels and the host
// twin of the tests.  What is picked, in the reference's order:
//   * per group g on NUMA node u = mapping['gpu'][g]: one GetFreeCpuBatch(u, n_proc[g], proc_smt): the scan walks
//     Node.cores in index order, i.e. the thread-0 cores of socket u ascending (they precede every sibling), and takes
//     a core when both of its threads are unused - [core, sibling] while the request is SMT and >= 2 cores are still
//     wanted, else [core].  On the bitmaps: the lowest set bits of t0[u] & t1[u].
//   * each GPU of the group: first unused GPU on the NIC's PCIe switch, else (NUMA mode) first unused GPU of NUMA u
//   * a second batch for the group's helper cores (helper_smt), a last one for the pod's misc cores on NUMA
//     mapping['cpu'][-1] with the REAL misc_cores_smt flag
//   * hugepages, busy time, and pods_used += 1 on every NIC that carries an RX / TX core: its capacity becomes 0
// A batch that cannot be filled from thread-0 cores makes the reference scan on into the sibling range and hand out
// cores twice or raise: status 1, parity undefined from there on, as in the reference.
//
// The NIC signatures of the node change with claims and GPU picks; their ids are looked up in a table the host
// derives from the dictionary (sig_key below is the canonical key of a NUMA node's pool set).  A state the dictionary
// does not hold yet is reported (status kCommitNewSig): the host interns it and patches the node.
#pragma once
#include "fit_core.h"

namespace nhdfit {

constexpr int kCommitOk = 0, kCommitWouldRaise = 1, kCommitNewSig = 2;
static_assert(sizeof(nhdfit_detail) == 128 && sizeof(nhdfit_origin) == 80 && sizeof(nhdfit_delta) == 96, "record sizes of include/nhdfit.h");

struct NodeState {                   // registers / LDS copy of one node while it is modified
    nhdfit_plane0 p0; nhdfit_plane1 p1; nhdfit_plane2 p2; nhdfit_plane3 p3; nhdfit_plane4 p4;
};

// lowest k set bits of x
NHD_HD uint64_t lowest_bits(uint64_t x, uint32_t k) {
    uint64_t out = 0;
    for (uint32_t i = 0; i < k && x; ++i) { const uint64_t b = x & (0 - x); out |= b; x ^= b; }
    return out;
}

// GetFreeCpuBatch(numa, num, smt) on the bitmaps of socket u (nhd/Node.py:502-519).  The reference walks ALL of Node.cores
// in index order and marks nothing while it walks: first the thread-0 cores of the socket whose both threads are free
// (`take`; `pair` = those whose sibling went out with them: SMT request and >= 2 cores still wanted), and - if the batch
// is still short - on into the sibling range, where the very same physical cores qualify again (their thread 0 is still
// "unused"): a request WITHOUT the SMT flag gets their second threads as cores of their own (`late`, ascending).  That
// is what happens to the pod-level misc cores of quirk Q1 (the filter halved them, Matcher.py:198; the commit does not,
// Node.py:799) - defined behaviour, no exception.  An SMT request that runs on is handed its own cores a second time;
// a batch that is short after both passes raises IndexError (Node.py:686): both return false - the reference's state is
// garbage or its unwind path (itself broken) runs, parity is undefined from there on.
NHD_HD bool take_batch(NodeState& s, uint32_t u, uint32_t num, bool smt_requested, uint64_t& take, uint64_t& pair, uint64_t& late) {
    const bool smt_node = (s.p2.flags & NHDFIT_NF_SMT) != 0;
    const uint64_t free = s.p0.t0[u] & s.p1.t1[u];
    const bool pairs = smt_node && smt_requested;
    const uint32_t avail = (uint32_t)popc64(free);
    const uint32_t n_take = pairs ? (num + 1) / 2 : num, n_pair = pairs ? num / 2 : 0;
    take = lowest_bits(free, n_take);
    pair = lowest_bits(free, n_pair);
    late = 0;
    uint32_t n_late = 0;
    if (smt_node && !pairs && num > avail) {                 // the walk runs on into the sibling range
        n_late = num - avail;
        late = lowest_bits(free, n_late);
    }
    s.p0.t0[u] &= ~take;
    if (smt_node) s.p1.t1[u] &= ~(pair | late);
    return (uint32_t)popc64(take) + (uint32_t)popc64(late) == n_take && (uint32_t)popc64(late) == n_late;
}

// canonical key of one NIC pool: free-GPU limit (NHDFIT_GLIMIT_NONE for the NUMA-mode pool) and the number of NICs
// per capacity class, both capped at NHDFIT_MAX_GROUPS exactly as the host packer caps them (more never matters).
// Counts travel as sixteen 4-bit saturating counters in one 64-bit word: no local arrays (dynamically indexed local
// arrays live in scratch memory on the GPU - a dependent memory round trip per access).
NHD_HD uint64_t count_add(uint64_t counts, uint32_t cls) {
    const uint32_t sh = 4 * (cls & 15u);
    return ((counts >> sh) & 15u) < 15u ? counts + (1ull << sh) : counts;
}
NHD_HD uint64_t pool_key_packed(uint32_t glimit, uint64_t counts) {
    uint64_t k = (uint64_t)(glimit & 0xFFu) << 48;
    for (uint32_t c = 0; c < NHDFIT_MAX_CLASSES; ++c) {
        const uint64_t n = (counts >> (4 * c)) & 15u;
        k |= (n > (uint64_t)kMaxG ? (uint64_t)kMaxG : n) << (3 * c);
    }
    return k;
}
NHD_HD uint64_t pool_key(uint32_t glimit, const uint8_t cnt[NHDFIT_MAX_CLASSES]) {       // host side: dictionary entries
    uint64_t k = (uint64_t)(glimit & 0xFFu) << 48;
    for (uint32_t c = 0; c < NHDFIT_MAX_CLASSES; ++c) k |= (uint64_t)(cnt[c] > kMaxG ? kMaxG : cnt[c]) << (3 * c);
    return k;
}
NHD_HD uint64_t mix64(uint64_t k) { k ^= k >> 33; k *= 0xFF51AFD7ED558CCDull; k ^= k >> 33; k *= 0xC4CEB9FE1A85EC53ull; k ^= k >> 33; return k; }
// key of a pool SET (order-free): sum of mixed pool keys, tagged with the pool count
NHD_HD uint64_t sig_key_add(uint64_t acc, uint64_t pk) { return acc + mix64(pk) + 0x9E3779B97F4A7C15ull; }

// keys of the NUMA-mode and PCI-mode signature of NUMA node u from a detail record (host packer: pack_node_into)
NHD_HD void sig_keys_of(const nhdfit_detail& d, uint32_t u, uint64_t& key_numa, uint64_t& key_pci) {
    key_numa = key_pci = 0;
    const uint32_t n = d.nic_cnt[u];
    uint64_t all = 0;
    for (uint32_t k = 0; k < n; ++k) all = count_add(all, d.nic_cls[u][k]);
    if (n) key_numa = sig_key_add(0, pool_key_packed(NHDFIT_GLIMIT_NONE, all));
    uint32_t seen = 0;                                           // local switch ids already turned into a pool
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t sw = d.nic_sw[u][k];
        if (seen >> sw & 1) continue;
        seen |= 1u << sw;
        const uint32_t gl = d.sw_free[sw] > kMaxG ? kMaxG : d.sw_free[sw];
        if (!gl) continue;                                       // no free GPU behind it: the pool hosts nothing
        uint64_t cnt = 0;
        for (uint32_t j = k; j < n; ++j)
            if (d.nic_sw[u][j] == sw) cnt = count_add(cnt, d.nic_cls[u][j]);
        key_pci = sig_key_add(key_pci, pool_key_packed(gl, cnt));
    }
}

struct SigTable {                    // open addressing, built by the host from the dictionary (nhdfit_set_dictionary)
    const uint64_t* key;             // [slots]: 0 = empty (the empty signature, key 0, is id 0 by convention)
    const uint32_t* id;
    uint32_t mask;                   // slots - 1
};
NHD_HD bool sig_lookup(const SigTable& t, uint64_t key, uint32_t& id) {
    if (key == 0) { id = 0; return true; }
    for (uint32_t s = (uint32_t)mix64(key) & t.mask, probes = 0; probes <= t.mask; ++probes, s = (s + 1) & t.mask) {
        if (t.key[s] == key) { id = t.id[s]; return true; }
        if (t.key[s] == 0) return false;
    }
    return false;
}

// Node.nics[].pods_used next to the capacity classes: 32 three-bit counters in the detail record (include/nhdfit.h),
// two's complement -3 .. 3 (an all-zero record = no pod on any NIC), the bit pattern 4 = out of range (sticky).
// pods_add returns +1 (pods_used > 0 afterwards: capacity 0), 0 (pods_used <= 0: the NIC's own capacity, nhd/Node.py:292)
// or -1 (out of range: only the Node object knows - treated as used here; the delta path reports NHDFIT_DELTA_REPACK
// and the host re-packs the node).
constexpr uint32_t kPodsLost = 4;
NHD_HD uint32_t pods_get(const nhdfit_detail& d, uint32_t u, uint32_t k) {
    const uint32_t bit = 3 * (u * NHDFIT_MAX_NICS_PER_NUMA + k), by = bit >> 3, sh = bit & 7;
    const uint32_t w = (uint32_t)d.nic_pods[by] | (by + 1 < sizeof d.nic_pods ? (uint32_t)d.nic_pods[by + 1] << 8 : 0u);
    return (w >> sh) & 7u;
}
NHD_HD void pods_set(nhdfit_detail& d, uint32_t u, uint32_t k, uint32_t v) {
    const uint32_t bit = 3 * (u * NHDFIT_MAX_NICS_PER_NUMA + k), by = bit >> 3, sh = bit & 7;
    const uint32_t m = 7u << sh, x = (v & 7u) << sh;
    d.nic_pods[by] = (uint8_t)((d.nic_pods[by] & ~m) | x);
    if (sh > 5 && by + 1 < sizeof d.nic_pods) d.nic_pods[by + 1] = (uint8_t)((d.nic_pods[by + 1] & ~(m >> 8)) | (x >> 8));
}
NHD_HD int pods_add(nhdfit_detail& d, uint32_t u, uint32_t k, int delta) {
    const uint32_t cur = pods_get(d, u, k);
    if (cur == kPodsLost) return -1;
    const int nv = (cur < 4 ? (int)cur : (int)cur - 8) + delta;
    if (nv < -3 || nv > 3) { pods_set(d, u, k, kPodsLost); return -1; }
    pods_set(d, u, k, (uint32_t)nv & 7u);
    return nv > 0 ? 1 : 0;
}

// One placement on one node.  `s` / `d` are modified in place; `out` receives the physical ids.
// Written over both request forms (fit_core.h req_traits): a big request's placement record carries two mask words per batch
// (nhdfit_big_placement, shared with the wide nodes) of which a node of the planes fills word 0.
NHD_HD uint64_t& batch_mask(uint64_t& m) { return m; }
NHD_HD uint64_t& batch_mask(uint64_t (&m)[2]) { m[1] = 0; return m[0]; }
template <class R, class PL>
NHD_HD int commit_node_t(NodeState& s, nhdfit_detail& d, const R& r, const typename req_traits<R>::Mapping& m, double busy_time,
                         const SigTable& sigs, PL& out) {
    constexpr int kMaxG = req_traits<R>::kG;                                        // (per request form; shadows the table pass's constant)
    const int G = (int)r.n_groups;
    int status = kCommitOk;
    for (int g = 0; g < kMaxG; ++g) {
        batch_mask(out.proc_take[g]) = batch_mask(out.proc_pair[g]) = batch_mask(out.help_take[g]) = batch_mask(out.help_pair[g]) =
            batch_mask(out.proc_late[g]) = batch_mask(out.help_late[g]) = 0;
        for (int k = 0; k < NHDFIT_PLACEMENT_GPUS; ++k) out.gpu[g][k] = 0xFF;
        out.numa[g] = -1;
    }
    batch_mask(out.misc_take) = batch_mask(out.misc_pair) = batch_mask(out.misc_late) = 0;
    out.numa[kMaxG] = -1;
    out.pad[0] = out.pad[1] = 0;
    s.p4.busy_time = busy_time;                                                     // SetBusy, nhd/Node.py:843-845
    uint32_t claimed0 = 0, claimed1 = 0;                                            // NIC ordinals to claim, per NUMA node
    bool gpu_taken = false;
    for (int g = 0; g < G; ++g) {
        const uint32_t u = (uint32_t)m.gpu[g] & 1u;
        out.numa[g] = (int8_t)u;
        if (!take_batch(s, u, r.n_proc[g], (r.smt_bits >> g & 1) != 0, batch_mask(out.proc_take[g]), batch_mask(out.proc_pair[g]), batch_mask(out.proc_late[g]))) status = kCommitWouldRaise;
        const uint32_t nu = (uint32_t)m.nic_numa[g] & 1u, nk = (uint32_t)m.nic_idx[g] & 15u;
        const uint32_t sw = d.nic_sw[nu][nk];
        for (uint32_t k = 0; k < r.gpus[g]; ++k) {
            int pick = -1;
            for (int x = 0; x < d.n_gpus && pick < 0; ++x)                          // GetFreePciGpuFromNic, Node.py:648-655
                if ((s.p2.gpu_free >> x & 1) && d.gpu_sw[x] == sw) pick = x;
            if (pick < 0 && r.map_type != NHDFIT_MAP_PCI)
                for (int x = 0; x < d.n_gpus && pick < 0; ++x)                      // GetNextGpuFree, Node.py:495-500
                    if ((s.p2.gpu_free >> x & 1) && (s.p2.gpu_numa1 >> x & 1) == u) pick = x;
            if (pick < 0) { status = kCommitWouldRaise; continue; }
            s.p2.gpu_free &= ~(1u << pick);
            if (d.sw_free[d.gpu_sw[pick]]) d.sw_free[d.gpu_sw[pick]]--;
            gpu_taken = true;
            if (k < (uint32_t)NHDFIT_PLACEMENT_GPUS) out.gpu[g][k] = (uint8_t)pick;
        }
        if (!take_batch(s, u, r.n_help[g], (r.smt_bits >> (kMaxG + g) & 1) != 0, batch_mask(out.help_take[g]), batch_mask(out.help_pair[g]), batch_mask(out.help_late[g]))) status = kCommitWouldRaise;
        if (r.nic_use >> g & 1) { if (nu) claimed1 |= 1u << nk; else claimed0 |= 1u << nk; }
    }
    if (r.hugepages_gb > 0) s.p2.hp_free -= r.hugepages_gb;                        // Node.py:794-796
    const uint32_t mu = (uint32_t)m.cpu[G] & 1u;
    out.numa[kMaxG] = (int8_t)mu;
    if (!take_batch(s, mu, r.n_misc, r.misc_smt_enabled != 0, batch_mask(out.misc_take), batch_mask(out.misc_pair), batch_mask(out.misc_late))) status = kCommitWouldRaise;   // Node.py:799
    // ClaimPodNICResources: pods_used += 1; the capacity class is 0 (= 0.0) while pods_used > 0 (Node.py:292, 644-646).
    // A counter that is already out of the tracked range keeps its class: it left the range on the side its class
    // says (free: pods_used < -3, used: > 3), and one more pod does not bring it back across zero.
    for (uint32_t k = 0; k < (uint32_t)NHDFIT_MAX_NICS_PER_NUMA; ++k)
        for (uint32_t u = 0; u < 2; ++u)
            if ((u ? claimed1 : claimed0) >> k & 1)
                if (pods_get(d, u, k) != kPodsLost && pods_add(d, u, k, 1) != 0) d.nic_cls[u][k] = 0;
    // the node's NIC signatures under the new NIC / GPU state (a NUMA node without a claim keeps its ids unless a GPU
    // was taken: the free-GPU count behind a switch enters the PCI-mode pools)
    for (uint32_t u = 0; u < 2; ++u) {
        if (!(u ? claimed1 : claimed0) && !gpu_taken) continue;
        uint64_t kn, kp;
        uint32_t idn = 0, idp = 0;
        sig_keys_of(d, u, kn, kp);
        if (!sig_lookup(sigs, kn, idn) || !sig_lookup(sigs, kp, idp)) { if (status == kCommitOk) status = kCommitNewSig; }
        s.p3.sig_numa[u] = (uint16_t)idn;
        s.p3.sig_pci[u] = (uint16_t)idp;
    }
    out.status = (uint8_t)status;
    return status;
}
NHD_HD int commit_node(NodeState& s, nhdfit_detail& d, const nhdfit_req& r, const nhdfit_mapping& m, double busy_time,
                       const SigTable& sigs, nhdfit_placement& out) {
    return commit_node_t<nhdfit_req, nhdfit_placement>(s, d, r, m, busy_time, sigs, out);
}

// ---- K3: one delta on one node ---------------------------------------------------------
// free GPUs behind every local switch, from the free mask (what pack_node_into counts, nhd/Node.py:266-273)
NHD_HD void recount_sw_free(const NodeState& s, nhdfit_detail& d) {
    for (uint32_t k = 0; k < (uint32_t)NHDFIT_MAX_SWITCHES; ++k) d.sw_free[k] = 0;
    for (uint32_t x = 0; x < d.n_gpus && x < (uint32_t)NHDFIT_MAX_GPUS; ++x)
        if (s.p2.gpu_free >> x & 1) d.sw_free[d.gpu_sw[x]]++;
}
// signature ids of both NUMA nodes from the detail record; false = a state the dictionary does not hold
NHD_HD bool resign(NodeState& s, const nhdfit_detail& d, const SigTable& sigs) {
    bool ok = true;
    for (uint32_t u = 0; u < 2; ++u) {
        uint64_t kn, kp;
        uint32_t idn = 0, idp = 0;
        sig_keys_of(d, u, kn, kp);
        if (!sig_lookup(sigs, kn, idn) || !sig_lookup(sigs, kp, idp)) ok = false;
        s.p3.sig_numa[u] = (uint16_t)idn;
        s.p3.sig_pci[u] = (uint16_t)idp;
    }
    return ok;
}
// RemoveResourcesFromTopology / AddResourcesFromTopology / ResetResources / the scalar setters on the packed state
// (nhd/Node.py:530-636, 144-161, 308-310, 489-493, 843-845; nhd/NHDScheduler.py:533-570).  `o` may be written by
// SET_HUGEPAGES (ttl_hugepages_gb).  Returns NHDFIT_DELTA_*.
NHD_HD int apply_delta(NodeState& s, nhdfit_detail& d, nhdfit_origin& o, const nhdfit_delta& q, const SigTable& sigs) {
    int status = NHDFIT_DELTA_OK;
    bool nic_or_gpu = false;
    switch (q.op) {
    case NHDFIT_DELTA_TAKE:
    case NHDFIT_DELTA_GIVE: {
        const bool take = q.op == NHDFIT_DELTA_TAKE;
        const bool smt = (s.p2.flags & NHDFIT_NF_SMT) != 0;
        for (uint32_t u = 0; u < 2; ++u) {                           // cores[..].used = True / False
            if (take) { s.p0.t0[u] &= ~q.t0[u]; if (smt) s.p1.t1[u] &= ~q.t1[u]; }
            else      { s.p0.t0[u] |= q.t0[u];  if (smt) s.p1.t1[u] |= q.t1[u]; }
        }
        const uint32_t all = d.n_gpus >= 32 ? 0xFFFFFFFFu : (1u << d.n_gpus) - 1u;
        const uint32_t g = q.gpus & all;
        if (g) {                                                     // GetGPU(device_id).used = True / False
            s.p2.gpu_free = take ? s.p2.gpu_free & ~g : s.p2.gpu_free | g;
            recount_sw_free(s, d);
            nic_or_gpu = true;
        }
        for (uint32_t i = 0; i < q.nic_n && i < (uint32_t)NHDFIT_DELTA_MAX_NICS; ++i) {   // nic_core_pairing: pods_used +- 1
            const uint32_t u = q.nic[i] >> 4 & 1u, k = q.nic[i] & 15u;
            if (k >= d.nic_cnt[u]) continue;
            const int r = pods_add(d, u, k, take ? 1 : -1);
            if (r < 0) status = NHDFIT_DELTA_REPACK;
            d.nic_cls[u][k] = r == 0 ? o.nic_base[u][k] : 0;
            nic_or_gpu = true;
        }
        if (q.hugepages_gb > 0) s.p2.hp_free = take ? s.p2.hp_free - q.hugepages_gb : s.p2.hp_free + q.hugepages_gb;
        break;
    }
    case NHDFIT_DELTA_RESET: {
        const bool smt = (s.p2.flags & NHDFIT_NF_SMT) != 0;
        for (uint32_t u = 0; u < 2; ++u) { s.p0.t0[u] = o.t0[u]; if (smt) s.p1.t1[u] = o.t1[u]; }
        s.p2.gpu_free = d.n_gpus >= 32 ? 0xFFFFFFFFu : (1u << d.n_gpus) - 1u;
        recount_sw_free(s, d);
        for (uint32_t u = 0; u < 2; ++u)
            for (uint32_t k = 0; k < d.nic_cnt[u]; ++k) { pods_set(d, u, k, 0); d.nic_cls[u][k] = o.nic_base[u][k]; }
        s.p2.hp_free = o.hp_total;
        nic_or_gpu = true;
        break;
    }
    case NHDFIT_DELTA_SET_FLAGS: {
        const uint32_t m = q.flags_mask & (NHDFIT_NF_MAINTENANCE | NHDFIT_NF_ACTIVE);
        s.p2.flags = (s.p2.flags & ~m) | (q.flags_value & m);
        break;
    }
    case NHDFIT_DELTA_SET_GROUPS: s.p3.groups = q.groups; s.p4.group_set = q.group_set; break;
    case NHDFIT_DELTA_SET_BUSY: s.p4.busy_time = q.busy_time; break;
    case NHDFIT_DELTA_SET_HUGEPAGES: s.p2.hp_free = q.hugepages_gb; o.hp_total = q.hp_total; break;
    default: status = NHDFIT_DELTA_REPACK; break;
    }
    if (nic_or_gpu && !resign(s, d, sigs) && status == NHDFIT_DELTA_OK) status = NHDFIT_DELTA_NEW_SIG;
    return status;
}

}  // namespace nhdfit
