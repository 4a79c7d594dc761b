// seq2_kernel.h - mode B, the decision-engine form: decisions under the scheduler's commit semantics with the commits spread
// over the chip.  Device code of libnhdfit.so; included by nhdfit.hip inside its anonymous namespace after seq_kernel.h.
// gfx950 only.
//
// What the sequential loop (nhd/NHDScheduler.py:425-437: FindNode, then SetBusy / SetPhysicalIdsFromMapping /
// ClaimPodNICResources, nhd/NHDScheduler.py:289-304) really chains, pod by pod:
//   * for a pod that requests GPUs a commit changes exactly one thing: the node is busy from then on (SetBusy,
//     nhd/Node.py:843-850; nhd/Matcher.py:107-111 drops busy nodes for such pods), i.e. gone - every other node still is
//     what the snapshot's verdict row says.  Its DECISION is the first bit of its row that no earlier pod took: a bitmap
//     operation.  Its mapping and commit touch a node no other pod with GPUs will ever look at: they can happen any
//     time later, anywhere on the chip;
//   * a pod without GPUs is not stopped by busy nodes: it takes the first node (GPU-less nodes first, SelectNode,
//     nhd/Matcher.py:401-413) that still has the resources, and so needs the state the earlier commits left on the nodes
//     it looks at - a true chain: consecutive such pods pile onto the same node until it is full.
// k_decide (below): block 0 decides - a sequencer wavefront walks the pods in the caller's order, speculator wavefronts run ahead
// of it for the pods without GPUs (verification against the node's current version, commit computed while the sequencer
// validates the version), fetcher wavefronts park the windows of the pods with GPUs; blocks 1.. are workers: they take queue
// entries - map + commit a pod with GPUs on its node, publish the node (pub[v] = 1), re-evaluate a committed node's column for
// the tiles that hold GPU-less pods and clear the bits of the pods that lost it (hints).
// A commit that leaves a node in a NIC state without a signature id poisons the node and is reported: the host undoes the
// batch and runs the general kernel (k_seq), whose stop / intern / resume protocol the caller knows.
struct DecideArgs {
    SeqArgs s;
    unsigned long long* queue;   // [2 P] work items, 0 = not written yet (pre-zeroed): bit 63 valid, bit 62 kind (0 commit, 1 patch),
                                 // bits 32..61 caller's pod index (commit), bits 0..31 node
    uint32_t* ctrl;              // [0] tickets handed out to workers, [1] 1 + items pushed, once the driver is through
    uint32_t* mat;               // [n] pub[v]: commits of this batch whose state is in the mirror (0 = as the snapshot left it); bit 31 = poisoned
    uint32_t* flags;             // [1] a commit met a NIC state without a signature, [3] a wait ran out (never expected)
    uint32_t lds_sigs, lds_states, lds_choose;   // block 0: stage the signature hash table / the set-state tables / the G <= 2 choose table in LDS (they fit)
    const uint32_t* list_n; uint32_t n_n;   // the pods without GPUs (valid requests), caller's indices ascending
    const uint32_t* list_g; uint32_t n_g;   // every other pod
    const uint4* ent_n; const uint4* ent_g; // per list entry (k_decide_prep): caller's index, staged position, the snapshot winner's local index
                                            // (kNoNode: none), 1 = that winner is a node without GPUs
    uint32_t queue_len;          // entries of `queue`
    uint32_t ncls;               // NIC capacity classes of the dictionary
    uint32_t hash_slots;         // block 0's multiset of GPU-less commits (decide_hash_slots)
    uint32_t span;               // chunks block 0's two node bit maps cover (<= s.chunks).  Shorter than the mirror (config 5's whole cluster: 4 096
                                 // chunks = 2 x 32 KB of LDS on their own): a decision on a node past the span ends the pass with flags[3] and the
                                 // host falls back on the general kernel - first fit fills a cluster from the front, a batch seldom gets that far
    uint32_t dbg;                // tuning aid (NHDFIT_SEQ_SKIP, tuning build; results are wrong with it): 1 no first-touch copy, 2 no commit, 4 no result / node stores
};
constexpr uint32_t kNoNode = 0xFFFFFFFFu;
constexpr unsigned long long kItemValid = 1ull << 63, kItemPatch = 1ull << 62;
// tiles (of those that hold GPU-less pods) one patch item covers: a committed node's column is re-evaluated by up to three
// worker wavefronts at once
__host__ __device__ inline uint32_t patch_span(uint32_t ngl) { return ngl <= 24u ? 8u : ngl; }   // (many tiles: one item - the workers are the bottleneck then)

// first-touch copy of a node for apply = 0 (whole wavefront), as in k_seq
// known_fresh: the caller knows that nothing touched the node in this batch (no look-up needed)
__device__ __forceinline__ void note_first_touch(const SeqArgs& a, uint32_t v, const NodeState& st, const nhdfit_detail& dd, uint32_t lane, bool known_fresh = false) {
    if (!known_fresh) {
        int32_t seen = 0;             // (read and written past the CU's vector cache: another CU may have touched the node)
        if (lane == 0) seen = __hip_atomic_load(&a.touched[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seen = __builtin_amdgcn_readfirstlane(seen);
        if (seen >= 0) return;
    }
    uint32_t slot = 0;
    if (lane == 0) { slot = atomicAdd(&a.counters[0], 1u); __hip_atomic_store(&a.touched[v], (int32_t)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
    if (a.keep_undo) {
        uint32_t* dst = reinterpret_cast<uint32_t*>(&a.undo[slot]);
        if (lane == 0) dst[0] = v;
        if (lane < sizeof(NodeState) / 4) dst[4 + lane] = reinterpret_cast<const uint32_t*>(&st)[lane];
        if (lane < sizeof(nhdfit_detail) / 4) dst[4 + sizeof(NodeState) / 4 + lane] = reinterpret_cast<const uint32_t*>(&dd)[lane];
    }
}

__device__ __forceinline__ void load_node_lds(const SeqArgs& a, uint32_t v, NodeState* st_out, nhdfit_detail* det_out, uint32_t lane) {
    uint32_t* st = reinterpret_cast<uint32_t*>(st_out);
    if (lane < 5) {
        const uint4 q = lane == 0 ? *reinterpret_cast<const uint4*>(a.p0 + v) : lane == 1 ? *reinterpret_cast<const uint4*>(a.p1 + v) :
                        lane == 2 ? *reinterpret_cast<const uint4*>(a.p2 + v) : lane == 3 ? *reinterpret_cast<const uint4*>(a.p3 + v) :
                                    *reinterpret_cast<const uint4*>(a.p4 + v);
        st[lane * 4 + 0] = q.x; st[lane * 4 + 1] = q.y; st[lane * 4 + 2] = q.z; st[lane * 4 + 3] = q.w;
    }
    if (lane >= 8 && lane < 16) {
        const uint4 q = reinterpret_cast<const uint4*>(a.det + v)[lane - 8];
        uint32_t* dd = reinterpret_cast<uint32_t*>(det_out) + (lane - 8) * 4;
        dd[0] = q.x; dd[1] = q.y; dd[2] = q.z; dd[3] = q.w;
    }
}
// the same past the CU's vector cache: for a node this kernel itself stored earlier (its line may sit stale in L1)
__device__ __forceinline__ void load_node_lds_coherent(const SeqArgs& a, uint32_t v, NodeState* st_out, nhdfit_detail* det_out, uint32_t lane) {
    uint64_t* st = reinterpret_cast<uint64_t*>(st_out);
    if (lane < 10) {
        const uint32_t pl = lane >> 1, h = lane & 1;
        const uint64_t* src = pl == 0 ? reinterpret_cast<const uint64_t*>(a.p0 + v) : pl == 1 ? reinterpret_cast<const uint64_t*>(a.p1 + v) :
                              pl == 2 ? reinterpret_cast<const uint64_t*>(a.p2 + v) : pl == 3 ? reinterpret_cast<const uint64_t*>(a.p3 + v) :
                                        reinterpret_cast<const uint64_t*>(a.p4 + v);
        st[lane] = __hip_atomic_load(src + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane >= 16 && lane < 32)
        reinterpret_cast<uint64_t*>(det_out)[lane - 16] = __hip_atomic_load(reinterpret_cast<const uint64_t*>(a.det + v) + (lane - 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_node_lds(const SeqArgs& a, uint32_t v, const NodeState* st_in, const nhdfit_detail* det_in, uint32_t lane) {
    const uint32_t* st = reinterpret_cast<const uint32_t*>(st_in);
    if (lane < 5) {
        const uint4 q = make_uint4(st[lane * 4], st[lane * 4 + 1], st[lane * 4 + 2], st[lane * 4 + 3]);
        if (lane == 0) *reinterpret_cast<uint4*>(a.p0 + v) = q;
        else if (lane == 1) *reinterpret_cast<uint4*>(a.p1 + v) = q;
        else if (lane == 2) *reinterpret_cast<uint4*>(a.p2 + v) = q;
        else if (lane == 3) *reinterpret_cast<uint4*>(a.p3 + v) = q;
        else *reinterpret_cast<uint4*>(a.p4 + v) = q;
    }
    if (lane >= 8 && lane < 16) {
        const uint32_t* dd = reinterpret_cast<const uint32_t*>(det_in) + (lane - 8) * 4;
        reinterpret_cast<uint4*>(a.det + v)[lane - 8] = make_uint4(dd[0], dd[1], dd[2], dd[3]);
    }
}

// a speculator's publication (two-stage commit): every plane but thread 1's whole; thread 1 loses the bits stage 2 found - an AND,
// because the copy in LDS holds a superset of them and the stage 2 of a later pod on the same node may already have cleared its own
__device__ __forceinline__ void store_node_lds_chain(const SeqArgs& a, uint32_t v, const NodeState* st_in, const nhdfit_detail* det_in, uint32_t lane,
                                                     uint64_t clear0, uint64_t clear1) {
    const uint32_t* st = reinterpret_cast<const uint32_t*>(st_in);
    if (lane < 5 && lane != 1) {
        const uint4 q = make_uint4(st[lane * 4], st[lane * 4 + 1], st[lane * 4 + 2], st[lane * 4 + 3]);
        if (lane == 0) *reinterpret_cast<uint4*>(a.p0 + v) = q;
        else if (lane == 2) *reinterpret_cast<uint4*>(a.p2 + v) = q;
        else if (lane == 3) *reinterpret_cast<uint4*>(a.p3 + v) = q;
        else *reinterpret_cast<uint4*>(a.p4 + v) = q;
    }
    if (lane == 1 && clear0) atomicAnd(reinterpret_cast<unsigned long long*>(&a.p1[v].t1[0]), ~(unsigned long long)clear0);
    if (lane == 5 && clear1) atomicAnd(reinterpret_cast<unsigned long long*>(&a.p1[v].t1[1]), ~(unsigned long long)clear1);
    if (lane >= 8 && lane < 16) {
        const uint32_t* dd = reinterpret_cast<const uint32_t*>(det_in) + (lane - 8) * 4;
        reinterpret_cast<uint4*>(a.det + v)[lane - 8] = make_uint4(dd[0], dd[1], dd[2], dd[3]);
    }
}

// the committed node against the tiles that hold pods without GPUs: sixteen lanes per (node, tile) evaluate the W assignment
// words, OR them together and clear the node's bit in the rows of the pods that lost it (hints for the pods to come)
__device__ __forceinline__ void patch_columns(const SeqArgs& a, uint32_t v, const NodeState& st, const uint16_t* gl_tiles, uint32_t k_begin, uint32_t ngl,
                                              const Layout* L4, uint32_t lane) {
    const uint32_t p = lane & 15u, grp = lane >> 4;
    const NodeIdx ni = node_index(st.p0, st.p1, st.p2, st.p4, a.fc_dim, a.fg_dim, a.ngs);
    const bool busy = (a.now - st.p4.busy_time) < kMinBusySecs;
    for (uint32_t k0 = k_begin; k0 < ngl; k0 += 4) {
        const uint32_t k = k0 + grp;
        uint64_t lost = 0;
        uint32_t t = 0;
        if (k < ngl) {
            t = gl_tiles[k];
            const uint64_t need = a.tile_masks[2 * t];
            const uint8_t* img = a.tabs + (size_t)t * a.pitch;
            const Layout& L = L4[a.tile_wcls[t]];
            uint64_t term = p < L.W ? node_term_cold(img, L, ni, st.p3, a.tile_masks[2 * t + 1], p) : 0ull;
            for (int m = 1; m < 16; m <<= 1) term |= __shfl_xor(term, m, 16);
            lost = ~(term & node_pred_cold(img, L, ni, busy, need)) & ~need;        // pods with GPUs go by the taken bits
        }
        for (uint32_t qq = 0; qq < 4; ++qq) {
            const uint32_t j = qq * 16 + p;
            if ((lost >> j & 1) && (size_t)t * 64 + j < a.P)
                atomicAnd(reinterpret_cast<unsigned long long*>(&NHDFIT_ROW(a, v >> 6, (size_t)t * 64 + j)), ~(1ull << (v & 63)));
        }
    }
}

// ---- the commit step with the wavefront's lanes (commit_core.h commit_node, same arithmetic) -----------------------------
// One lane walking GetFreeCpuBatch's picks bit by bit, the GPU lists and the NIC pools of a node costs ~5 us per placement -
// the chain of the pods without GPUs pays it pod after pod.  Here lane i answers for core i / GPU i / NIC i and a ballot
// collects the answer: lowest k set bits = "bit set and fewer than k set bits below it".  Every lane ends up with the same
// (uniform) values; lane 0 writes them to the LDS copies.  The host twin keeps the scalar form; the GPU parity tests compare
// the two through their results (tests/golden/commit, the oracle's loop).
__device__ __forceinline__ uint64_t lowest_bits_wave(uint64_t x, uint32_t k, uint32_t lane) {
    const uint64_t below = lane ? x & ((1ull << lane) - 1ull) : 0ull;
    return __ballot((x >> lane & 1) && (uint32_t)__popcll(below) < k);
}
struct WaveBatch { uint64_t take, pair, late; bool ok; };
__device__ __forceinline__ WaveBatch take_batch_wave(uint64_t& t0, uint64_t& t1, bool smt_node, uint32_t num, bool smt_requested, uint32_t lane) {
    const uint64_t free = t0 & t1;
    const bool pairs = smt_node && smt_requested;
    const uint32_t avail = (uint32_t)__popcll(free);
    const uint32_t n_take = pairs ? (num + 1) / 2 : num, n_pair = pairs ? num / 2 : 0;
    WaveBatch b;
    b.take = lowest_bits_wave(free, n_take, lane);
    b.pair = n_pair ? lowest_bits_wave(free, n_pair, lane) : 0ull;
    b.late = 0;
    uint32_t n_late = 0;
    if (smt_node && !pairs && num > avail) { n_late = num - avail; b.late = lowest_bits_wave(free, n_late, lane); }   // the walk runs on into the sibling range
    t0 &= ~b.take;
    if (smt_node) t1 &= ~(b.pair | b.late);
    b.ok = (uint32_t)__popcll(b.take) + (uint32_t)__popcll(b.late) == n_take && (uint32_t)__popcll(b.late) == n_late;
    return b;
}
// key of the NIC pool the lanes flagged `in` form (pool_key_packed over their classes)
__device__ __forceinline__ uint64_t pool_key_wave(uint32_t glimit, bool in, uint32_t my_cls, uint32_t ncls) {
    uint64_t k = (uint64_t)(glimit & 0xFFu) << 48;
    for (uint32_t c = 0; c < ncls; ++c) {                                 // (classes the dictionary does not hold count zero)
        const uint32_t n = (uint32_t)__popcll(__ballot(in && my_cls == c));
        k |= (uint64_t)(n > (uint32_t)kMaxG ? (uint32_t)kMaxG : n) << (3 * c);
    }
    return k;
}
__device__ __forceinline__ void sig_keys_wave(const nhdfit_detail& d, uint32_t u, uint32_t lane, uint32_t ncls, uint64_t& key_numa, uint64_t& key_pci) {
    key_numa = key_pci = 0;
    const uint32_t n = d.nic_cnt[u];
    const bool valid = lane < n;
    const uint32_t my_cls = valid ? d.nic_cls[u][lane & 15u] : 0xFFu, my_sw = valid ? d.nic_sw[u][lane & 15u] : 0xFFu;
    if (n) key_numa = sig_key_add(0, pool_key_wave(NHDFIT_GLIMIT_NONE, valid, my_cls, ncls));
    uint64_t todo = __ballot(valid);                                      // NICs whose switch has not been turned into a pool yet
    while (todo) {
        const uint32_t k = (uint32_t)__builtin_ctzll(todo);
        const uint32_t sw = d.nic_sw[u][k];
        const uint64_t same = __ballot(valid && my_sw == sw);
        todo &= ~same;
        const uint32_t gl = d.sw_free[sw] > kMaxG ? kMaxG : d.sw_free[sw];
        if (!gl) continue;                                                // no free GPU behind it: the pool hosts nothing
        key_pci = sig_key_add(key_pci, pool_key_wave(gl, valid && my_sw == sw, my_cls, ncls));
    }
}
// `s` / `d` / `out` live in LDS (one copy per wavefront); every lane returns the same status
__device__ __forceinline__ int commit_node_wave(NodeState& s, nhdfit_detail& d, const nhdfit_req& r, const nhdfit_mapping& m, double busy_time,
                                                const SigTable& sigs, uint32_t ncls, nhdfit_placement& out, uint32_t lane) {
    const int G = (int)r.n_groups;
    int status = kCommitOk;
    {   // the placement record: zeros, 0xFF for the GPU list and the NUMA entries (bytes 144 .. 180 of the 256)
        const uint32_t o = lane * 4u;
        reinterpret_cast<uint32_t*>(&out)[lane] = o >= 144u && o < 180u ? 0xFFFFFFFFu : o == 180u ? 0x000000FFu : 0u;
    }
    uint64_t t0[2] = {s.p0.t0[0], s.p0.t0[1]}, t1[2] = {s.p1.t1[0], s.p1.t1[1]};
    const bool smt_node = (s.p2.flags & NHDFIT_NF_SMT) != 0;
    uint32_t gpu_free = s.p2.gpu_free;
    const uint32_t gpu_numa1 = s.p2.gpu_numa1, n_gpus = d.n_gpus;
    const uint32_t my_gsw = lane < n_gpus && lane < (uint32_t)NHDFIT_MAX_GPUS ? d.gpu_sw[lane & 31u] : 0xFFu;
    uint32_t claimed0 = 0, claimed1 = 0;
    bool gpu_taken = false;
    // the mapping bit / nibble-packed under compile-time indices: indexed by the run-time group number it would live in scratch memory
    uint32_t m_gpu = 0, m_nnuma = 0, m_nidx = 0, mu = 0;
#pragma unroll
    for (int g = 0; g < kMaxG; ++g) {
        m_gpu |= ((uint32_t)m.gpu[g] & 1u) << g; m_nnuma |= ((uint32_t)m.nic_numa[g] & 1u) << g; m_nidx |= ((uint32_t)m.nic_idx[g] & 15u) << (4 * g);
    }
#pragma unroll
    for (int g = 0; g <= kMaxG; ++g) if (g == G) mu = (uint32_t)m.cpu[g] & 1u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int g = 0; g < G; ++g) {
        const uint32_t u = (m_gpu >> g) & 1u;
        const WaveBatch pb = take_batch_wave(t0[u], t1[u], smt_node, r.n_proc[g], (r.smt_bits >> g & 1) != 0, lane);
        if (!pb.ok) status = kCommitWouldRaise;
        const uint32_t nu = (m_nnuma >> g) & 1u, nk = (m_nidx >> (4 * g)) & 15u;
        const uint32_t sw = d.nic_sw[nu][nk];
        uint32_t picks = 0xFFFFFFFFu;                                     // up to four picks travel in a register (byte k), the rest through lane 0
        for (uint32_t k = 0; k < r.gpus[g]; ++k) {
            const bool mine_free = lane < n_gpus && (gpu_free >> lane & 1);
            uint64_t cand = __ballot(mine_free && my_gsw == sw);          // GetFreePciGpuFromNic, Node.py:648-655
            if (!cand && r.map_type != NHDFIT_MAP_PCI) cand = __ballot(mine_free && (gpu_numa1 >> lane & 1) == u);   // GetNextGpuFree, Node.py:495-500
            if (!cand) { status = kCommitWouldRaise; continue; }
            const uint32_t pick = (uint32_t)__builtin_ctzll(cand);
            gpu_free &= ~(1u << pick);
            gpu_taken = true;
            if (lane == 0) {
                const uint32_t psw = d.gpu_sw[pick];
                if (d.sw_free[psw]) d.sw_free[psw]--;
                if (k < (uint32_t)NHDFIT_PLACEMENT_GPUS) out.gpu[g][k] = (uint8_t)pick;
            }
            (void)picks;
        }
        const WaveBatch hb = take_batch_wave(t0[u], t1[u], smt_node, r.n_help[g], (r.smt_bits >> (4 + g) & 1) != 0, lane);
        if (!hb.ok) status = kCommitWouldRaise;
        if (r.nic_use >> g & 1) { if (nu) claimed1 |= 1u << nk; else claimed0 |= 1u << nk; }
        if (lane == 0) {
            out.numa[g] = (int8_t)u;
            out.proc_take[g] = pb.take; out.proc_pair[g] = pb.pair; out.proc_late[g] = pb.late;
            out.help_take[g] = hb.take; out.help_pair[g] = hb.pair; out.help_late[g] = hb.late;
        }
    }
    const WaveBatch mb = take_batch_wave(t0[mu], t1[mu], smt_node, r.n_misc, r.misc_smt_enabled != 0, lane);     // Node.py:799
    if (!mb.ok) status = kCommitWouldRaise;
    if (lane == 0) {
        out.numa[kMaxG] = (int8_t)mu;
        out.misc_take = mb.take; out.misc_pair = mb.pair; out.misc_late = mb.late;
        s.p0.t0[0] = t0[0]; s.p0.t0[1] = t0[1]; s.p1.t1[0] = t1[0]; s.p1.t1[1] = t1[1];
        s.p2.gpu_free = gpu_free;
        if (r.hugepages_gb > 0) s.p2.hp_free -= r.hugepages_gb;              // Node.py:794-796
        s.p4.busy_time = busy_time;                                          // SetBusy, nhd/Node.py:843-845
        for (uint32_t cl = claimed0 | (claimed1 << 16); cl; cl &= cl - 1u) {  // ClaimPodNICResources (commit_core.h: the same rule; every NIC touched once)
            const uint32_t b = (uint32_t)__builtin_ctz(cl), u = b >> 4, k = b & 15u;
            if (pods_get(d, u, k) != kPodsLost && pods_add(d, u, k, 1) != 0) d.nic_cls[u][k] = 0;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t u = 0; u < 2; ++u) {
        if (!(u ? claimed1 : claimed0) && !gpu_taken) continue;
        uint64_t kn, kp;
        uint32_t idn = 0, idp = 0;
        sig_keys_wave(d, u, lane, ncls, kn, kp);
        if (!sig_lookup(sigs, kn, idn) || !sig_lookup(sigs, kp, idp)) { if (status == kCommitOk) status = kCommitNewSig; }
        if (lane == 0) { s.p3.sig_numa[u] = (uint16_t)idn; s.p3.sig_pci[u] = (uint16_t)idp; }
    }
    if (lane == 0) out.status = (uint8_t)status;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return status;
}

// The commit of a pod WITHOUT GPUs in two stages (round 5).  Consecutive GPU-less pods pile onto the same node, so
// pod k + 1's verification waits for pod k's commit - and what it needs of it is little: the free-core COUNTS, hugepages, the NIC
// classes and the signature ids.  A batch of cores takes the lowest free bits of its socket, batch after batch, so the socket's free
// set after ALL of a pod's batches is the free set minus its lowest N bits, N = the sum of what the batches take - one ballot per
// socket instead of one to three per batch.  Stage 1 (commit_summary_wave, on the chain) does that, the scalar writes, the NIC
// claims and the signature ids, and leaves thread 1's bitmap alone: the sibling bits of cores handed out in pairs (or by the run-on
// walk, quirk Q1) are cleared by stage 2 - until then the copy holds a SUPERSET of thread 1's final bits, all of them inside the N
// bits just cleared in thread 0, so `t0 & t1` (the only form the verification and every later pick read the bitmaps in) is exact.
// Stage 2 (commit_picks_wave, off the chain) walks the batches on the socket's free sets as they were BEFORE the pod - the placement
// record's masks - and returns the thread-1 bits to clear; the caller ANDs them into the mirror (commutative: later pods' stage 2
// may overtake).  Together: commit_node_wave's state, record and status (tests/harness/wave_emul.cpp runs both on emulated lanes).
__device__ __forceinline__ int commit_summary_wave(NodeState& s, nhdfit_detail& d, const nhdfit_req& r, const nhdfit_mapping& m, double busy_time,
                                                   const SigTable& sigs, uint32_t ncls, uint32_t lane, uint64_t& free0, uint64_t& free1) {
    // the request's commit counts read ONCE, as independent dword loads: n_proc[0..3], n_help[0..3], (n_misc, smt_bits,
    // misc_smt_enabled, nic_use) - byte by byte inside the group loop every one was a round trip with a wait on the chain
    static_assert(kMaxG == 4 && offsetof(nhdfit_req, n_proc) % 4 == 0 && offsetof(nhdfit_req, n_help) % 4 == 0 && offsetof(nhdfit_req, n_misc) % 4 == 0 &&
                  offsetof(nhdfit_req, smt_bits) == offsetof(nhdfit_req, n_misc) + 1 && offsetof(nhdfit_req, misc_smt_enabled) == offsetof(nhdfit_req, n_misc) + 2 &&
                  offsetof(nhdfit_req, nic_use) == offsetof(nhdfit_req, n_misc) + 3, "packed reads of the request");
    const uint32_t w_proc = *reinterpret_cast<const uint32_t*>(r.n_proc), w_help = *reinterpret_cast<const uint32_t*>(r.n_help);
    const uint32_t w_tail = *reinterpret_cast<const uint32_t*>(&r.n_misc);
    const int32_t hugepages = r.hugepages_gb;
    const uint32_t smt_bits = (w_tail >> 8) & 0xFFu, nic_use = w_tail >> 24;
    const int G = (int)r.n_groups;
    free0 = s.p0.t0[0] & s.p1.t1[0];
    free1 = s.p0.t0[1] & s.p1.t1[1];
    const bool smt_node = (s.p2.flags & NHDFIT_NF_SMT) != 0;
    uint32_t m_gpu = 0, m_nnuma = 0, m_nidx = 0, mu = 0;
#pragma unroll
    for (int g = 0; g < kMaxG; ++g) {
        m_gpu |= ((uint32_t)m.gpu[g] & 1u) << g; m_nnuma |= ((uint32_t)m.nic_numa[g] & 1u) << g; m_nidx |= ((uint32_t)m.nic_idx[g] & 15u) << (4 * g);
    }
#pragma unroll
    for (int g = 0; g <= kMaxG; ++g) if (g == G) mu = (uint32_t)m.cpu[g] & 1u;
    // cores each socket loses: batch by batch min(asked, left) - a batch that asks for more than is left takes what is there
    uint32_t left0 = (uint32_t)__popcll(free0), left1 = (uint32_t)__popcll(free1), took0 = 0, took1 = 0;
    auto batch = [&](uint32_t u, uint32_t num, bool smt_requested) {
        const uint32_t want = smt_node && smt_requested ? (num + 1u) / 2u : num;
        if (u) { const uint32_t t = want < left1 ? want : left1; left1 -= t; took1 += t; }
        else   { const uint32_t t = want < left0 ? want : left0; left0 -= t; took0 += t; }
    };
    uint32_t claimed0 = 0, claimed1 = 0, repriced = 0;
    for (int g = 0; g < G; ++g) {
        const uint32_t u = (m_gpu >> g) & 1u;
        batch(u, (w_proc >> (8 * g)) & 0xFFu, (smt_bits >> g & 1) != 0);
        batch(u, (w_help >> (8 * g)) & 0xFFu, (smt_bits >> (4 + g) & 1) != 0);
        const uint32_t nu = (m_nnuma >> g) & 1u, nk = (m_nidx >> (4 * g)) & 15u;
        if (nic_use >> g & 1) { if (nu) claimed1 |= 1u << nk; else claimed0 |= 1u << nk; }
    }
    batch(mu, w_tail & 0xFFu, ((w_tail >> 16) & 0xFFu) != 0);
    const uint64_t gone0 = took0 ? lowest_bits_wave(free0, took0, lane) : 0ull, gone1 = took1 ? lowest_bits_wave(free1, took1, lane) : 0ull;
    if (lane == 0) {
        s.p0.t0[0] &= ~gone0; s.p0.t0[1] &= ~gone1;
        if (hugepages > 0) s.p2.hp_free -= hugepages;                        // Node.py:794-796
        s.p4.busy_time = busy_time;                                          // SetBusy, nhd/Node.py:843-845
        if (claimed0 | claimed1) {
            // ClaimPodNICResources (pods_add / the capacity class, as commit_node_wave) on the counters and classes READ ONCE: the 32
            // three-bit counters are bytes 82..93 of the record - NUMA 0's sixteen in bits 16..63 of the 8 bytes at 80, NUMA 1's in
            // bits 0..47 of the 8 bytes at 88 -, a NUMA node's sixteen classes two 8-byte words.  (pods_get / pods_add walk them byte
            // by byte: six dependent LDS round trips per claimed NIC on the chain.)
            static_assert(offsetof(nhdfit_detail, nic_pods) == 82 && offsetof(nhdfit_detail, nic_cls) == 16 && NHDFIT_MAX_NICS_PER_NUMA == 16 && sizeof(d.nic_pods) == 12,
                          "packed reads of the detail record");
            uint64_t* dw = reinterpret_cast<uint64_t*>(__builtin_assume_aligned(&d, 8));
            uint64_t pods[2] = {dw[10], dw[11]}, cls[2][2] = {{dw[2], dw[3]}, {dw[4], dw[5]}};
            const uint64_t pods_was[2] = {pods[0], pods[1]};
            for (uint32_t cl = claimed0 | (claimed1 << 16); cl; cl &= cl - 1u) {
                const uint32_t b = (uint32_t)__builtin_ctz(cl), u = b >> 4, k = b & 15u;
                const uint32_t sh = (u ? 0u : 16u) + 3u * k;
                const uint32_t cur = (uint32_t)(pods[u] >> sh) & 7u;
                if (cur == kPodsLost) continue;
                const int nv = (cur < 4u ? (int)cur : (int)cur - 8) + 1;
                const bool lost = nv > 3;                                     // (a claim only ever counts up)
                pods[u] = (pods[u] & ~(7ull << sh)) | ((uint64_t)(lost ? kPodsLost : ((uint32_t)nv & 7u)) << sh);
                if (lost || nv > 0) {                                         // pods_used > 0 (or unknown): the NIC's capacity is gone
                    const uint32_t cs = 8u * (k & 7u);
                    if ((cls[u][k >> 3] >> cs) & 0xFFull) { cls[u][k >> 3] &= ~(0xFFull << cs); repriced |= 1u << u; }
                }
            }
            if (pods[0] != pods_was[0]) dw[10] = pods[0];
            if (pods[1] != pods_was[1]) dw[11] = pods[1];
            if (repriced & 1u) { dw[2] = cls[0][0]; dw[3] = cls[0][1]; }
            if (repriced & 2u) { dw[4] = cls[1][0]; dw[5] = cls[1][1]; }
        }
    }
    repriced = (uint32_t)__builtin_amdgcn_readfirstlane((int)repriced);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int status = kCommitOk;
    for (uint32_t u = 0; u < 2; ++u) {
        // A NUMA node's signature ids are a function of its NICs' capacity classes and the free GPUs behind their switches.  No GPU
        // is taken here, so they only move when a claim takes a NIC's capacity away - its FIRST claim: pods that pile onto a node
        // claim the NIC their predecessor claimed, and the keys (64-bit mixes, hash look-ups: most of this stage) need not be formed
        if (!(repriced >> u & 1u)) continue;
        uint64_t kn, kp;
        uint32_t idn = 0, idp = 0;
        sig_keys_wave(d, u, lane, ncls, kn, kp);
        if (!sig_lookup(sigs, kn, idn) || !sig_lookup(sigs, kp, idp)) status = kCommitNewSig;
        if (lane == 0) { s.p3.sig_numa[u] = (uint16_t)idn; s.p3.sig_pci[u] = (uint16_t)idp; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return status;
}
// stage 2: `free0` / `free1` = the sockets' free sets as stage 1 found them; `out` in LDS; clear0 / clear1 = thread-1 bits to clear
__device__ __forceinline__ int commit_picks_wave(uint64_t free0, uint64_t free1, bool smt_node, const nhdfit_req& r, const nhdfit_mapping& m,
                                                 nhdfit_placement& out, uint32_t lane, uint64_t& clear0, uint64_t& clear1) {
    const int G = (int)r.n_groups;
    int status = kCommitOk;
    {   // the placement record: zeros, 0xFF for the GPU list and the NUMA entries (bytes 144 .. 180 of the 256)
        const uint32_t o = lane * 4u;
        reinterpret_cast<uint32_t*>(&out)[lane] = o >= 144u && o < 180u ? 0xFFFFFFFFu : o == 180u ? 0x000000FFu : 0u;
    }
    uint64_t t0[2] = {free0, free1}, t1[2] = {free0, free1};                  // (take_batch_wave reads t0 & t1 only)
    uint32_t m_gpu = 0, mu = 0;
#pragma unroll
    for (int g = 0; g < kMaxG; ++g) m_gpu |= ((uint32_t)m.gpu[g] & 1u) << g;
#pragma unroll
    for (int g = 0; g <= kMaxG; ++g) if (g == G) mu = (uint32_t)m.cpu[g] & 1u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int g = 0; g < G; ++g) {
        const uint32_t u = (m_gpu >> g) & 1u;
        const WaveBatch pb = take_batch_wave(t0[u], t1[u], smt_node, r.n_proc[g], (r.smt_bits >> g & 1) != 0, lane);
        const WaveBatch hb = take_batch_wave(t0[u], t1[u], smt_node, r.n_help[g], (r.smt_bits >> (4 + g) & 1) != 0, lane);
        if (!pb.ok || !hb.ok) status = kCommitWouldRaise;
        if (lane == 0) {
            out.numa[g] = (int8_t)u;
            out.proc_take[g] = pb.take; out.proc_pair[g] = pb.pair; out.proc_late[g] = pb.late;
            out.help_take[g] = hb.take; out.help_pair[g] = hb.pair; out.help_late[g] = hb.late;
        }
    }
    const WaveBatch mb = take_batch_wave(t0[mu], t1[mu], smt_node, r.n_misc, r.misc_smt_enabled != 0, lane);     // Node.py:799
    if (!mb.ok) status = kCommitWouldRaise;
    if (lane == 0) {
        out.numa[kMaxG] = (int8_t)mu;
        out.misc_take = mb.take; out.misc_pair = mb.pair; out.misc_late = mb.late;
    }
    clear0 = free0 & ~t1[0];                                                  // the pairs' and the run-on walk's sibling bits
    clear1 = free1 & ~t1[1];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return status;
}

// ---- k_decide: speculate, then retire in order ------------------------------------------------------------------------------
// Block 0 decides; blocks 1.. commit the pods with GPUs (workers).  Inside block 0:
//   wavefront 0, the SEQUENCER, walks the pods in the caller's order.  A pod with GPUs costs it a window look-up, a bit and a
//     queue entry (as before).  A pod without GPUs costs it a VALIDATION: some speculator has already found the pod's node and
//     says which version of that node it looked at; the sequencer compares that with the number of decisions made on the node
//     so far and either retires the pod (the decision stands: count + 1, node taken) or sends it back.
//   kSpecWaves SPECULATORS take the pods without GPUs round robin and run ahead of the sequencer: window of the pod's row,
//     candidates in SelectNode's order, each verified against the node's state at the latest version there is (map_on_state_wave).
//     Feasibility only ever shrinks inside a batch, so a candidate that is rejected on ANY past or present state of the node
//     is rejected for good - whatever happens before the pod's turn; only the one candidate that ACCEPTS needs its version
//     checked at the pod's turn.  The speculator posts (node, version), computes the commit while it waits, and on "retire"
//     publishes: the new state goes live in an LDS cache of the block (pods that pile onto the same node read it from there),
//     then into the mirror, results and placement to the caller's arrays, a patch item to the workers.  On "again" it
//     re-examines the node (the sequencer stands still meanwhile, so the second answer is final).
//   Versions: decisions on node v so far = [a pod with GPUs took it] (bit map) + the GPU-less pods retired on it (a small multiset in
//     LDS: one entry per such commit, open addressing) - both written by the sequencer only.  pub[v] (`mat`, global) = commits
//     whose state is in the mirror.  A reader assumes version D, waits for pub[v] == D (D > 0), reads, and confirms afterwards
//     that the count is still D: writers of version D + 1 only start once the count says D + 1, so the read was not torn.
//   the other wavefronts fetch ahead for the pods with GPUs (first window of the row nobody took).
#ifndef NHDFIT_SPEC_WAVES
#define NHDFIT_SPEC_WAVES 6
#endif
#ifndef NHDFIT_DECIDE_WAVES
#define NHDFIT_DECIDE_WAVES 8
#endif
constexpr int kDecideWaves = NHDFIT_DECIDE_WAVES, kDecideRing = 64, kSpecWaves = NHDFIT_SPEC_WAVES, kSpecCache = 4, kWorkerBlocks = 8;
static_assert((kSpecWaves == 6 || kSpecWaves == 9) && kDecideWaves >= 4 * kSpecWaves / 3 && kDecideWaves - 1 - kSpecWaves >= 1, "wavefront roles of block 0: speculators sit on SIMDs 1-3 (wave & 3 != 0), two or three deep; at least one fetcher");
// (eight wavefronts: 256 registers each - with sixteen the speculators' verification and commit spilled 114 VGPRs to scratch memory and
// every step of the chain paid for it: config 4 677 k -> 800 k decisions/s with a single fetcher, profiles/r04/mode_b_variants.log)
constexpr uint32_t kSpinLimit = 1u << 22;                      // x ~100 cycles of s_sleep: a fraction of a second, then give up
constexpr uint32_t kPubPoison = 1u << 31;                      // pub[v]: the node was left in a NIC state without a signature id
constexpr uint32_t kNicSigs = 64;                              // dictionaries up to this many NIC signatures: the pod's NIC-feasible assignments
                                                               // per signature ride along in LDS
__host__ __device__ inline uint32_t decide_hash_slots(uint32_t n_gpu_less) {      // multiset of GPU-less commits: load <= 1/2, >= one wavefront's probe
    uint32_t h = 64;
    while (h < 2u * n_gpu_less) h <<= 1;
    return h;
}

// everything a sequential pass starts from, cleared by one launch (it was seven fill commands in a row in front of the pass)
struct SeqResetArgs {
    uint64_t* taken; uint32_t chunks;            // nodes that received a pod of this batch
    int32_t* touched; uint32_t n;                // first-touch slots (-1 = none)
    uint32_t* counters; uint32_t* flags;         // [4] each
    uint32_t* ctrl; uint32_t* mat;               // the decision engine's: [32] control words, [n] published versions (null: not this pass)
    unsigned long long* queue; uint32_t queue_len;
};
__global__ __launch_bounds__(256) void k_seq_reset(SeqResetArgs a) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x, stride = gridDim.x * 256u;
    for (uint32_t i = t; i < a.chunks; i += stride) a.taken[i] = 0ull;
    for (uint32_t i = t; i < a.n; i += stride) { a.touched[i] = -1; if (a.mat) a.mat[i] = 0u; }
    if (t < 4u) { a.counters[t] = 0u; a.flags[t] = 0u; }
    if (a.ctrl && t < 32u) a.ctrl[t] = 0u;
    if (a.queue) for (uint32_t i = t; i < a.queue_len; i += stride) a.queue[i] = 0ull;
}

// what a fetcher / speculator needs to start on list entry j, gathered once for the whole batch (three dependent look-ups otherwise)
__global__ __launch_bounds__(256) void k_decide_prep(const uint32_t* __restrict__ list, uint32_t n, const uint32_t* __restrict__ order,
                                                     const unsigned long long* __restrict__ score, uint64_t global_base, uint4* __restrict__ out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t e = list[j], pos = order[e];
    const unsigned long long sc = score[pos];
    out[j] = make_uint4(e, pos, sc ? (uint32_t)(NHDFIT_SCORE_INDEX(sc) - global_base) : kNoNode, (uint32_t)(sc >> 63));
}

// G4: the batch holds pods of four processing groups (their mapping is the generic set model's: scratch arrays).  The host launches
// k_decide<false> for every other batch - no private segment to speak of, 10 KB per lane otherwise.
template <bool G4>
__global__ __launch_bounds__(64 * kDecideWaves) void k_decide(DecideArgs q) {
    const SeqArgs& a = q.s;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t tiles = (a.P + kTile - 1) / kTile;
    const uint32_t n_pods = a.P;
    __shared__ Layout s_L[kWClasses];
    __shared__ double s_caps[NHDFIT_MAX_CLASSES];
    __shared__ uint32_t s_ngl;
    constexpr uint32_t kGlLds = 1024;                        // (more tiles with GPU-less pods than this: the rest goes unpatched - hints only)
    __shared__ uint16_t s_gl[kGlLds];
    // per-wavefront scratch: request / result / placement (workers and speculators), node (workers)
    __shared__ PaddedReq s_wreq[kDecideWaves];
    __shared__ NodeState s_wst[kDecideWaves];
    __shared__ nhdfit_detail s_wdet[kDecideWaves];
    __shared__ nhdfit_placement s_wplace[kDecideWaves];
    __shared__ SeqResult s_wres[kDecideWaves];
    // block 0 only.  Pods with GPUs: ring of parked windows
    __shared__ uint64_t s_win[kDecideRing][64];
    __shared__ uint32_t s_base[kDecideRing], s_pos[kDecideRing];
    __shared__ int32_t s_have[kDecideRing];                    // 0 no candidate, 2 window parked
    __shared__ uint32_t s_ready[kDecideRing];                  // sequence number + 1 of the pod parked in the slot
    __shared__ uint32_t s_done, s_abort, s_nitems, s_spec_done;
    // speculators
    __shared__ uint64_t s_swin[kSpecWaves][64];
    __shared__ uint32_t s_snic[kSpecWaves][kNicSigs];          // per signature: low half = assignments that pass the NIC test on NUMA 0, high half on NUMA 1
    __shared__ NodeState s_cst[kSpecWaves * kSpecCache];       // nodes the speculators committed to, most recent kSpecCache each; the entry a
    __shared__ __align__(16) nhdfit_detail s_cdet[kSpecWaves * kSpecCache];  // speculator works in is tagged kNoNode until its commit is retired
    __shared__ uint32_t s_ctag[64], s_cver[64];
    __shared__ NodeState s_pst[kSpecWaves];                    // a never-touched node as it was (first-touch copy, written at retirement)
    __shared__ nhdfit_detail s_pdet[kSpecWaves];
    __shared__ uint32_t s_post[kSpecWaves], s_verd[kSpecWaves];  // speculator -> sequencer: (pod + 1) << 4 | attempt; back: the same << 1 | retire
    __shared__ uint32_t s_rv[kSpecWaves], s_rd[kSpecWaves];    // the posted node (kNoNode: none takes the pod) and the version looked at
    __shared__ uint32_t s_examv[kSpecWaves], s_pende[kSpecWaves];   // the node a speculator is examining / has posted, and its pod: a later pod does not post
                                                               // that node before the earlier one has made up its mind
    __shared__ uint32_t s_cnt[32];                             // tuning aid: [0] failed verifications [1] LDS cache hits [2] published states read [3] untouched [4] waits for an earlier pod's target [5] window rescans
                                                               // [8..14] speculator ticks: set-up, node state, verification, commit stage 1, waiting for the sequencer, publication, commit stage 2
    static_assert(kSpecWaves * kSpecCache <= 64, "the cache tags are searched by one wavefront");
    extern __shared__ __align__(16) uint8_t s_dyn[];

    if (tid == 0) s_ngl = 0;
    if (tid < (uint32_t)kWClasses) s_L[tid] = a.L[tid];
    if (tid < NHDFIT_MAX_CLASSES) s_caps[tid] = a.caps[tid];
    __syncthreads();
    for (uint32_t t = tid; t < tiles; t += 64 * kDecideWaves) {           // tiles that hold pods without GPUs (every block its own list)
        const uint32_t live = a.P - t * kTile < (uint32_t)kTile ? a.P - t * kTile : (uint32_t)kTile;
        const uint64_t lm = live == 64 ? ~0ull : (1ull << live) - 1;
        if (~a.tile_masks[2 * t] & lm) {
            const uint32_t at = atomicAdd(&s_ngl, 1u);
            if (at < kGlLds) s_gl[at] = (uint16_t)t;
        }
    }
    __threadfence_block();
    __syncthreads();
    const uint32_t ngl = s_ngl < kGlLds ? s_ngl : kGlLds;
    const uint16_t* gl_tiles = s_gl;
    auto wg_load = [](const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto wg_store = [](uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto dev_load = [](const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); };

    if (blockIdx.x != 0) {
        // ---- workers: one wavefront per queue entry ----------------------------------------------------------------------
        for (;;) {
            uint32_t ticket = 0;
            if (lane == 0) ticket = atomicAdd(&q.ctrl[0], 1u);
            ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
            if (ticket >= q.queue_len) return;
            unsigned long long item = 0;
            // (relaxed polls, ONE acquire behind the match: an acquire per poll is an L1 invalidate per poll - microseconds each, and a
            // hundred idle pollers doing it take a good part of the chip's bandwidth from the block that decides)
            for (uint32_t spin = 0;; ++spin) {
                item = __hip_atomic_load(&q.queue[ticket], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (item) break;
                const uint32_t fin = __hip_atomic_load(&q.ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (fin && ticket >= fin - 1u) {                          // block 0 is through: did it write this entry before it said so?
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    item = __hip_atomic_load(&q.queue[ticket], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (item) break;
                    return;
                }
                if (spin > kSpinLimit) { if (lane == 0) q.flags[3] = 1u; return; }   // the entry for this ticket may still come: the host must not trust the batch (it undoes it and runs k_seq)
                __builtin_amdgcn_s_sleep(16);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const uint32_t v = (uint32_t)item;
            NodeState& st = s_wst[wave];
            nhdfit_detail& dd = s_wdet[wave];
            if (item & kItemPatch) {                                      // a node a GPU-less pod was retired on: its column (hints; a state
                load_node_lds_coherent(a, v, &st, &dd, lane);             // torn by a later commit lies between two real ones: it clears no
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");    // bit the later one would not clear)
                __builtin_amdgcn_wave_barrier();
                const uint32_t k0 = (uint32_t)(item >> 32) & 0xFFFFu;
                if (ngl) patch_columns(a, v, st, gl_tiles, k0, k0 + patch_span(ngl) < ngl ? k0 + patch_span(ngl) : ngl, s_L, lane);
                continue;
            }
            const uint32_t mine = (uint32_t)(item >> 32) & 0x3FFFFFFFu, pos = a.order[mine];
            if (lane < sizeof(nhdfit_req) / 16) {
                const uint4 r4 = reinterpret_cast<const uint4*>(a.reqs + pos)[lane];
                uint32_t* dst = reinterpret_cast<uint32_t*>(&s_wreq[wave]) + lane * 4;
                dst[0] = r4.x; dst[1] = r4.y; dst[2] = r4.z; dst[3] = r4.w;
            }
            load_node_lds(a, v, &st, &dd, lane);                          // never written before in this batch: the snapshot's copy
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // map + commit of a pod with GPUs on the node the sequencer gave it (the snapshot's verdict stands: nothing touched the node)
            const nhdfit_req& rq = s_wreq[wave].r;
            SeqResult& res = s_wres[wave];
            nhdfit_placement& pl = s_wplace[wave];
            nhdfit_mapping mp = nhdfit_mapping{};
            const uint32_t tile = pos >> 6;
            const uint32_t bits = nic_assignment_bits_wave(a.tabs + (size_t)tile * a.pitch, s_L[a.tile_wcls[tile]], pos & 63, rq.map_type == NHDFIT_MAP_PCI, st.p3, lane);
            const bool ok = map_on_state_wave<G4>(rq, st, dd, s_caps, bits, a.mt, lane, mp);
            __builtin_amdgcn_wave_barrier();
            if (!(kTuning && (q.dbg & 1))) note_first_touch(a, v, st, dd, lane, true);
            __builtin_amdgcn_wave_barrier();
            int32_t status;
            if (lane == 0) { res.node = (int64_t)a.global_base + (int64_t)v; res.map = ok ? mp : nhdfit_mapping{}; }
            if (ok && kTuning && (q.dbg & 2)) status = kCommitOk;
            else if (ok) status = commit_node_wave(st, dd, rq, mp, a.now, a.sigs, q.ncls, pl, lane);
            else {                                                        // the row said feasible, the mapping disagrees: cannot happen
                if (lane < sizeof(nhdfit_placement) / 4) reinterpret_cast<uint32_t*>(&pl)[lane] = 0u;
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) pl.status = kCommitWouldRaise;
                status = kCommitWouldRaise;
            }
            if (lane == 0) { res.status = status; if (status == kCommitNewSig) q.flags[1] = 1u; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (!(kTuning && (q.dbg & 4))) {
                if (lane < sizeof(SeqResult) / 4) reinterpret_cast<uint32_t*>(&a.out[mine])[lane] = reinterpret_cast<const uint32_t*>(&res)[lane];
                if (a.place && lane < sizeof(nhdfit_placement) / 4) reinterpret_cast<uint32_t*>(&a.place[mine])[lane] = reinterpret_cast<const uint32_t*>(&pl)[lane];
                store_node_lds(a, v, &st, &dd, lane);
            }
            __threadfence();                                              // the node's new state is in the mirror: version 1 (a pod with GPUs is
            if (lane == 0)                                                // always the first to touch its node)
                __hip_atomic_store(&q.mat[v], status == kCommitNewSig ? (kPubPoison | 1u) : 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (ngl) patch_columns(a, v, st, gl_tiles, 0, ngl, s_L, lane);
        }
    }

    // ---- block 0: the decision engine ------------------------------------------------------------------------------------
    uint8_t* dynp = s_dyn;
    const uint32_t span = q.span;                                         // the bit maps' reach in chunks (<= a.chunks): no node past it is ever taken
    uint64_t* s_taken = carve<uint64_t>(dynp, span);                      // nodes that received a pod of this batch (busy: gone for pods with GPUs)
    uint64_t* s_tgpu = carve<uint64_t>(dynp, span);                       // ... a pod with GPUs
    uint32_t* s_hash = carve<uint32_t>(dynp, q.hash_slots);               // one entry per GPU-less pod retired: its node
    uint32_t* s_isn = carve<uint32_t>(dynp, (a.P + 31) / 32);             // pods without GPUs
    const uint32_t hmask = q.hash_slots - 1u;
    SigTable sigs = a.sigs;
    MapTables mt = a.mt;
    uint64_t* l_skey = nullptr; uint32_t* l_sid = nullptr; uint64_t* l_info = nullptr; uint32_t* l_next = nullptr; uint32_t* l_asc = nullptr;
    if (q.lds_sigs) { l_skey = carve<uint64_t>(dynp, (size_t)a.sigs.mask + 1); l_sid = carve<uint32_t>(dynp, (size_t)a.sigs.mask + 1); }
    if (q.lds_states) { l_info = carve<uint64_t>(dynp, a.mt.st.n); l_next = carve<uint32_t>(dynp, (size_t)a.mt.st.n * 8); l_asc = carve<uint32_t>(dynp, 256); }
    uint8_t* l_choose = q.lds_choose ? carve<uint8_t>(dynp, kChooseEntries) : nullptr;
    if (tid == 0) { s_done = 0; s_abort = 0; s_nitems = 0; s_spec_done = 0; }
    if (tid < kDecideRing) s_ready[tid] = 0;
    if (tid < 64) { s_ctag[tid] = kNoNode; s_cver[tid] = 0; }
    if (tid < kSpecWaves) { s_post[tid] = 0; s_verd[tid] = 0; s_examv[tid] = kNoNode; s_pende[tid] = 0xFFFFFFFFu; }
    if (tid < 32) s_cnt[tid] = 0;
    for (uint32_t k = tid; k < span; k += 64 * kDecideWaves) { s_taken[k] = 0; s_tgpu[k] = 0; }
    for (uint32_t k = tid; k < q.hash_slots; k += 64 * kDecideWaves) s_hash[k] = kNoNode;
    for (uint32_t k = tid; k < (a.P + 31) / 32; k += 64 * kDecideWaves) s_isn[k] = 0;
    if (q.lds_sigs) {
        for (uint32_t k = tid; k <= a.sigs.mask; k += 64 * kDecideWaves) { l_skey[k] = a.sigs.key[k]; l_sid[k] = a.sigs.id[k]; }
        sigs = SigTable{l_skey, l_sid, a.sigs.mask};
    }
    if (q.lds_states) {
        for (uint32_t k = tid; k < a.mt.st.n; k += 64 * kDecideWaves) l_info[k] = a.mt.st.info[k];
        for (uint32_t k = tid; k < a.mt.st.n * 8; k += 64 * kDecideWaves) l_next[k] = a.mt.st.next[k];
        for (uint32_t k = tid; k < 256; k += 64 * kDecideWaves) l_asc[k] = a.mt.st.asc[k];
        mt.st = SetStates{l_info, l_next, l_asc, a.mt.st.n};
    }
    if (q.lds_choose) {
        const uint4* src = reinterpret_cast<const uint4*>(a.mt.choose_tab);
        uint4* dst = reinterpret_cast<uint4*>(l_choose);
        for (uint32_t k = tid; k < kChooseEntries / 16; k += 64 * kDecideWaves) dst[k] = src[k];
        mt.choose_tab = l_choose;
    }
    __syncthreads();
    for (uint32_t k = tid; k < q.n_n; k += 64 * kDecideWaves) { const uint32_t e = q.list_n[k]; atomicOr(&s_isn[e >> 5], 1u << (e & 31)); }
    __syncthreads();

    // first window of 64 chunks at or after (from_chunk, from_bit) of pod `pos`'s row that holds a candidate -> win.
    //   mode 1: the nodes without GPUs; 2: every node; 3: every node nobody took yet (a pod that requests GPUs);
    //   4: the nodes with GPUs (a GPU-less pod whose pass over the GPU-less nodes found nothing)
    auto scan_window = [&](uint64_t* win, uint32_t pos, uint32_t mode, uint32_t from_chunk, uint32_t from_bit, uint32_t& base_out) -> bool {
        constexpr uint32_t kAhead = 4;                                    // windows requested together: a scan that has to go deep pays one round trip per four
        for (uint32_t base = from_chunk; base < a.chunks; base += 64 * kAhead) {
            uint64_t w[kAhead];
#pragma unroll
            for (uint32_t k = 0; k < kAhead; ++k) {
                const uint32_t c = base + 64 * k + lane;
                w[k] = c < a.chunks ? __hip_atomic_load(&NHDFIT_ROW(a, c, pos), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            }
#pragma unroll
            for (uint32_t k = 0; k < kAhead; ++k) {
                const uint32_t c = base + 64 * k + lane;
                uint64_t x = w[k];
                if (c < a.chunks) {
                    if (mode == 1) x &= a.nogpu[c];
                    else if (mode == 3) x &= c < span ? ~s_taken[c] : ~0ull;
                    else if (mode == 4) x &= ~a.nogpu[c];
                } else x = 0;
                if (c == from_chunk) x &= ~0ull << from_bit;
                if (__ballot(x != 0)) {
                    win[lane] = x;
                    base_out = base + 64 * k;
                    return true;
                }
            }
        }
        return false;
    };
    // decisions made on node v so far (wave-uniform): the sequencer's two structures
    auto hash_of = [&](uint32_t v) { return (v * 2654435761u) >> 7; };
    auto decisions_on = [&](uint32_t v) -> uint32_t {
        uint32_t d = (v >> 6) < span ? (uint32_t)(__hip_atomic_load(&s_tgpu[v >> 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> (v & 63)) & 1u : 0u;
        for (uint32_t h = hash_of(v);; h += 64) {
            const uint32_t key = __hip_atomic_load(&s_hash[(h + lane) & hmask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint64_t eq = __ballot(key == v), em = __ballot(key == kNoNode);
            if (em) return d + (uint32_t)__popcll(eq & ((1ull << __builtin_ctzll(em)) - 1ull));
            d += (uint32_t)__popcll(eq);
        }
    };
    bool stop = false;
    auto give_up = [&]() { stop = true; if (lane == 0) { q.flags[3] = 1u; wg_store(&s_abort, 1u); } };
    auto push = [&](unsigned long long item) {                            // one 8-byte store: the entry itself is the signal
        if (lane == 0) {
            const uint32_t at = atomicAdd(&s_nitems, 1u);
            if (at < q.queue_len) __hip_atomic_store(&q.queue[at], item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto not_placed = [&](uint32_t mine) {
        if (lane == 0) { SeqResult r; r.node = -1; r.map = nhdfit_mapping{}; r.status = 0; a.out[mine] = r; }
        if (a.place && lane < sizeof(nhdfit_placement) / 4) reinterpret_cast<uint32_t*>(&a.place[mine])[lane] = 0u;
    };
    constexpr uint32_t kFetchWaves = kDecideWaves - 1 - kSpecWaves;
    // wavefronts with wave & 3 != 0 below 4 * kSpecWaves / 3 speculate (the sequencer shares its SIMD with fetchers only), the rest fetch
    const int spec_id = (wave & 3u) && wave < 4u * (uint32_t)kSpecWaves / 3u ? (int)((wave >> 2) * 3u + (wave & 3u) - 1u) : -1;

    if (wave != 0 && spec_id < 0) {
        // ---- fetchers: pod e (with GPUs) goes to slot e % kDecideRing once the sequencer is past pod e - kDecideRing.  A fetcher takes
        // kFetchChunk list entries with one load and then eight pods at a time: their first windows are requested together (one round trip
        // for eight pods, the next eight requested before these are parked), masked with what is taken by the time each is parked
        const uint32_t first_all = 4u * (uint32_t)kSpecWaves / 3u;        // wavefronts from here on all fetch; below, those with wave & 3 == 0
        const uint32_t fid = wave < first_all ? (wave >> 2) - 1u : first_all / 4u - 1u + (wave - first_all);     // 0 .. kFetchWaves - 1
        constexpr uint32_t kFetchBatch = 8;
#ifndef NHDFIT_FETCH_CHUNK
#define NHDFIT_FETCH_CHUNK 16
#endif
        // consecutive list entries per fetcher: the ring holds kDecideRing pods, so with 64 entries per fetcher the sequencer is served by
        // one fetcher at a time (the others are a ring ahead and wait); 16 keeps four of them inside the ring
        constexpr uint32_t kFetchChunk = NHDFIT_FETCH_CHUNK;
        static_assert(kFetchChunk <= 64 && kFetchChunk % (2 * kFetchBatch) == 0, "two batches of eight in flight per chunk");
        struct Batch { uint32_t e[kFetchBatch], pos[kFetchBatch], from[kFetchBatch]; uint64_t w[kFetchBatch]; };
        // tuning aid (ctrl[2], [3], [13]; 100 MHz ticks): the fetcher's time by what it waits for - [2] list entries and the first windows of a chunk
        // (global round trips nothing else covers), [3] the ring (the sequencer has to leave a slot), [13] everything else (issue + park)
        unsigned long long f_acc[3] = {0, 0, 0}, f_last = kTuning ? (unsigned long long)wall_clock64() : 0ull;
        auto flap = [&](int k) { if (kTuning) { const unsigned long long t = wall_clock64(); f_acc[k] += t - f_last; f_last = t; } };
        for (uint32_t j0 = fid * kFetchChunk; j0 < q.n_g; j0 += kFetchWaves * kFetchChunk) {
            const uint32_t cnt = q.n_g - j0 < kFetchChunk ? q.n_g - j0 : kFetchChunk;
            uint4 ent = make_uint4(0, 0, kNoNode, 0);
            if (lane < cnt) ent = q.ent_g[j0 + lane];
            if (kTuning) { flap(2); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); flap(0); }
            // entries k0 .. k0 + 7: their first windows requested (one load instruction each, all in flight together)
            auto issue = [&](uint32_t k0, Batch& t) {
#pragma unroll
                for (uint32_t i = 0; i < kFetchBatch; ++i) {
                    const uint32_t k = k0 + i < cnt ? k0 + i : cnt - 1u;  // (past the end: the last entry again, never parked)
                    t.e[i] = (uint32_t)__builtin_amdgcn_readlane((int)ent.x, (int)k);
                    t.pos[i] = (uint32_t)__builtin_amdgcn_readlane((int)ent.y, (int)k);
                    t.from[i] = (uint32_t)__builtin_amdgcn_readlane((int)ent.z, (int)k);
                    const uint32_t c = (t.from[i] >> 6) + lane;
                    t.w[i] = 0;
                    if (t.from[i] != kNoNode && c < a.chunks) t.w[i] = __hip_atomic_load(&NHDFIT_ROW(a, c, t.pos[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            };
            // ... parked, each once the sequencer has left its slot; false = the batch was given up
            auto park = [&](uint32_t k0, const Batch& t) -> bool {
#pragma unroll
                for (uint32_t i = 0; i < kFetchBatch; ++i) {
                    if (k0 + i >= cnt) break;
                    const uint32_t e = t.e[i], slot = e % kDecideRing;
                    flap(2);
                    for (uint32_t spin = 0; e >= wg_load(&s_done) + kDecideRing; ++spin) {
                        if (spin > kSpinLimit || wg_load(&s_abort)) return false;
                        __builtin_amdgcn_s_sleep(2);
                    }
                    flap(1);
                    int32_t have = 0;
                    uint32_t wb = t.from[i] >> 6;
                    if (t.from[i] != kNoNode) {
                        const uint32_t c = wb + lane;
                        uint64_t w = t.w[i] & (c < a.chunks ? (c < span ? ~s_taken[c] : ~0ull) : 0ull);
                        if (lane == 0) w &= ~0ull << (t.from[i] & 63u);
                        if (__ballot(w != 0)) { s_win[slot][lane] = w; have = 2; }
                        else if (wb + 64u < a.chunks && scan_window(s_win[slot], t.pos[i], 3u, wb + 64u, 0, wb)) have = 2;
                    }
                    if (lane == 0) { s_have[slot] = have; s_pos[slot] = t.pos[i]; s_base[slot] = wb; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) wg_store(&s_ready[slot], e + 1);
                }
                return true;
            };
            Batch ba, bb;                                                 // one batch is always in flight while the other is parked
            issue(0, ba);
            for (uint32_t k0 = 0; k0 < cnt; k0 += 2 * kFetchBatch) {
                if (k0 + kFetchBatch < cnt) issue(k0 + kFetchBatch, bb);
                if (!park(k0, ba)) return;
                if (k0 + 2 * kFetchBatch < cnt) issue(k0 + 2 * kFetchBatch, ba);
                if (k0 + kFetchBatch < cnt && !park(k0 + kFetchBatch, bb)) return;
            }
        }
        if (kTuning && lane == 0 && fid == 0) { flap(2); q.ctrl[2] = (uint32_t)f_acc[0]; q.ctrl[3] = (uint32_t)f_acc[1]; q.ctrl[13] = (uint32_t)f_acc[2]; }
        return;
    }

    if (spec_id >= 0) {
        // ---- speculators ----------------------------------------------------------------------------------------------
        const uint32_t sp = (uint32_t)spec_id;
        __builtin_amdgcn_s_setprio(2);
        uint64_t* win = s_swin[sp];
        const nhdfit_req& rq = s_wreq[wave].r;
        SeqResult& res = s_wres[wave];
        nhdfit_placement& pl = s_wplace[wave];
        uint32_t cache_next = 0;
        uint32_t c_fail = 0, c_hit = 0, c_pub = 0, c_plain = 0, c_chain = 0, c_rescan = 0;
        unsigned long long t_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = kTuning ? wall_clock64() : 0;    // tuning aid (100 MHz ticks)
        auto lap = [&](int k) { if (kTuning) { const unsigned long long t = wall_clock64(); t_acc[k] += t - t_last; t_last = t; } };
        // an earlier pod is examining / has posted node v: its word on that node comes first
        auto earlier_pod_on = [&](uint32_t v, uint32_t e) {
            return __ballot(lane < (uint32_t)kSpecWaves && lane != sp && wg_load(&s_examv[lane]) == v && wg_load(&s_pende[lane]) < e) != 0;
        };
        for (uint32_t j = sp; j < q.n_n && !stop; j += kSpecWaves) {
            __builtin_amdgcn_s_setprio(2);
            const uint4 ent = q.ent_n[j];
            const uint32_t e = ent.x, mine = e, pos = ent.y;
            const uint32_t tile = pos >> 6;
            const Layout& L = s_L[a.tile_wcls[tile]];
            const uint8_t* img = a.tabs + (size_t)tile * a.pitch;
            const bool has_winner = ent.z != kNoNode;
            const bool nic_tab = has_winner && L.nsig <= kNicSigs;
            // the request, the pod's NIC-feasible assignments per signature and its first window: three fetches in flight together
            uint4 r4 = make_uint4(0, 0, 0, 0);
            if (lane < sizeof(nhdfit_req) / 16) r4 = reinterpret_cast<const uint4*>(a.reqs + pos)[lane];
            uint32_t nic_word = 0;
            if (nic_tab && lane < L.nsig) {                               // bit p of a half: assignment p passes the NIC test on that NUMA node
                uint32_t m0 = 0, m1 = 0;                                  // for a node with this signature (the cold R rows of the pod's tile)
                for (uint32_t pp = 0; pp < L.W; ++pp) {
                    m0 |= (uint32_t)(ld64(img, L.off_r0 + lane * L.row + pp * 8) >> (pos & 63) & 1) << pp;
                    m1 |= (uint32_t)(ld64(img, L.off_r1 + lane * L.row + pp * 8) >> (pos & 63) & 1) << pp;
                }
                nic_word = m0 | (m1 << 16);
            }
            uint32_t attempt = 0, wbase = 0, pass = 0;
            bool have = false, placed = false;
            if (has_winner) {
                pass = ent.w ? 1u : 2u;                                   // SelectNode: the nodes without GPUs first (the snapshot's winner says whether there is one)
                have = scan_window(win, pos, pass, ent.z >> 6, ent.z & 63u, wbase);
                if (!have && pass == 1u) { pass = 4u; have = scan_window(win, pos, 4u, 0, 0, wbase); }
            }
            if (lane < sizeof(nhdfit_req) / 16) {
                uint32_t* dst = reinterpret_cast<uint32_t*>(&s_wreq[wave]) + lane * 4;
                dst[0] = r4.x; dst[1] = r4.y; dst[2] = r4.z; dst[3] = r4.w;
            }
            if (nic_tab) s_snic[sp][lane] = nic_word;
            if (lane == 0) { s_pende[sp] = e; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            lap(0);
            // posts (v, version) and waits for the sequencer's word: true = retire
            auto post_and_wait = [&](uint32_t v, uint32_t ver, auto&& meanwhile) -> bool {
                const uint32_t id = ((e + 1u) << 4) | (attempt & 15u);
                if (lane == 0) { s_rv[sp] = v; s_rd[sp] = ver; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) wg_store(&s_post[sp], id);
                meanwhile();
                uint32_t word = 0;
                for (uint32_t spin = 0; ((word = wg_load(&s_verd[sp])) >> 1) != id && !stop; ++spin) {
                    if (spin > kSpinLimit) give_up();
                    if (wg_load(&s_abort)) stop = true;
                    __builtin_amdgcn_s_sleep(1);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                ++attempt;
                return !stop && (word & 1u);
            };
            while (have && !placed && !stop) {
                const uint64_t w = win[lane];
                const uint64_t any = __ballot(w != 0);
                if (!any) {                                               // window exhausted: the next one, then the next pass
                    ++c_rescan;
                    const uint32_t nb = wbase + 64;
                    if (nb < a.chunks && scan_window(win, pos, pass, nb, 0, wbase)) continue;
                    if (pass == 1u) { pass = 4u; have = scan_window(win, pos, 4u, 0, 0, wbase); continue; }
                    have = false;
                    continue;
                }
                const int l = __builtin_ctzll(any);
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w, l);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w >> 32), l);
                const uint32_t v = (wbase + (uint32_t)l) * 64u + (uint32_t)__builtin_ctzll(((uint64_t)hi << 32) | lo);
                if (lane == 0) wg_store(&s_examv[sp], v);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                // An earlier pod that is looking at this node (or has posted it) goes first - and this one waits for it BEFORE it
                // reads and verifies the node (round 5): consecutive pods without GPUs pile onto the same node, so the state an
                // earlier pod is about to change is not worth a verification, and a wavefront verifying in vain shares its SIMD's
                // issue slots with the one whose pod is next to retire (config 2: 3.75 verifications per pod, one of them needed).
                if (earlier_pod_on(v, e)) {
                    ++c_chain;
                    __builtin_amdgcn_s_setprio(0);
                    for (uint32_t spin = 0; earlier_pod_on(v, e) && !stop; ++spin) {
                        if (spin > kSpinLimit) give_up();
                        if (wg_load(&s_abort)) stop = true;
                        __builtin_amdgcn_s_sleep(2);
                    }
                    __builtin_amdgcn_s_setprio(2);
                    lap(4);
                    if (stop) break;
                }
                // the pod the sequencer is waiting for runs ahead of its SIMD's other wavefront
                if (wg_load(&s_done) == e) __builtin_amdgcn_s_setprio(3);
                lap(7);
                // the node at the latest version there is -> the cache entry this speculator works in (tagged kNoNode)
                const uint32_t ce = sp * kSpecCache + cache_next % kSpecCache;
                NodeState& st = s_cst[ce];
                nhdfit_detail& dd = s_cdet[ce];
                uint32_t ver = 0;
                bool stale_bit = false;
                for (uint32_t spin = 0; !stop; ++spin) {
                    if (spin > kSpinLimit) { give_up(); break; }
                    if (wg_load(&s_abort)) { stop = true; break; }
                    ver = decisions_on(v);
                    if (ver == 0) { load_node_lds(a, v, &st, &dd, lane); ++c_plain; }
                    else {
                        const uint64_t hit = __ballot(wg_load(&s_ctag[lane]) == v && wg_load(&s_cver[lane]) == ver);
                        bool got = false;
                        if (hit) {
                            const uint32_t src = (uint32_t)__builtin_ctzll(hit);
                            if (lane < sizeof(NodeState) / 4) reinterpret_cast<uint32_t*>(&st)[lane] = reinterpret_cast<const uint32_t*>(&s_cst[src])[lane];
                            if (lane < sizeof(nhdfit_detail) / 4) reinterpret_cast<uint32_t*>(&dd)[lane] = reinterpret_cast<const uint32_t*>(&s_cdet[src])[lane];
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                            got = wg_load(&s_ctag[src]) == v && wg_load(&s_cver[src]) == ver;      // (the owner may have recycled the entry meanwhile)
                            if (got) ++c_hit;
                        }
                        if (!got) {
                            // the mirror, once that version is in it; the row's word for the chunk rides along (bits the workers cleared meanwhile)
                            const uint32_t m = dev_load(&q.mat[v]);
                            const uint64_t fresh = __hip_atomic_load(&NHDFIT_ROW(a, v >> 6, pos), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (m & kPubPoison) { stop = true; break; }   // a NIC state without a signature id (reported by the committer)
                            if (!(fresh >> (v & 63) & 1)) { stale_bit = true; break; }
                            if (m < ver) { __builtin_amdgcn_s_sleep(1); continue; }      // its commit is in flight
                            if (m > ver) continue;                                         // decided and published since: count again
                            load_node_lds_coherent(a, v, &st, &dd, lane);                  // (after pub[v] was seen: the planes are at least that new)
                            ++c_pub;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    if (decisions_on(v) == ver) break;                    // nobody was allowed to write the node while it was read
                }
                if (stop) break;
                lap(1);
                bool ok = !stale_bit && rq.hugepages_gb <= st.p2.hp_free;                  // nhd/Matcher.py:78
                if (ok) {                                                 // cheap necessary condition before the table look-ups: enough free
                    const bool smt = (st.p2.flags & NHDFIT_NF_SMT) != 0;  // physical cores on the node as a whole
                    const uint64_t w_cpu = smt ? *reinterpret_cast<const uint64_t*>(rq.cpu_smt) : *reinterpret_cast<const uint64_t*>(rq.cpu_nosmt);   // (one read, not one per group)
                    uint32_t need = smt ? rq.misc_smt : rq.misc_nosmt;
                    for (uint32_t g = 0; g < rq.n_groups; ++g) need += (uint32_t)(w_cpu >> (16 * g)) & 0xFFFFu;
                    ok = need <= (uint32_t)popc64(st.p0.t0[0] & st.p1.t1[0]) + (uint32_t)popc64(st.p0.t0[1] & st.p1.t1[1]);
                }
                nhdfit_mapping mp = nhdfit_mapping{};
                lap(8);
                if (ok) {
                    const bool pci = rq.map_type == NHDFIT_MAP_PCI;
                    const uint32_t bits = nic_tab ? (s_snic[sp][pci ? st.p3.sig_pci[0] : st.p3.sig_numa[0]] & 0xFFFFu) & (s_snic[sp][pci ? st.p3.sig_pci[1] : st.p3.sig_numa[1]] >> 16)
                                                  : nic_assignment_bits_wave(img, L, pos & 63, pci, st.p3, lane);
                    lap(9);
                    ok = map_on_state_wave<G4>(rq, st, dd, s_caps, bits, mt, lane, mp);
                }
                __builtin_amdgcn_wave_barrier();
                lap(2);
                if (!ok) {                                                // not this node - at no later version either
                    ++c_fail;
                    if (lane == 0) wg_store(&s_examv[sp], kNoNode);
                    if (lane == (uint32_t)l) win[lane] = w & ~(1ull << (v & 63));
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    continue;
                }
                // an earlier pod that is looking at this node (or has posted it) goes first; if it takes the node, this one looks again
                if (earlier_pod_on(v, e)) {                               // (one that came to this node while it was verified)
                    ++c_chain;
                    __builtin_amdgcn_s_setprio(0);
                    for (uint32_t spin = 0; earlier_pod_on(v, e) && !stop; ++spin) {
                        if (spin > kSpinLimit) give_up();
                        if (wg_load(&s_abort)) stop = true;
                        __builtin_amdgcn_s_sleep(2);
                    }
                    __builtin_amdgcn_s_setprio(2);
                    lap(4);
                    if (stop) break;
                    if (decisions_on(v) != ver) continue;                 // (the bit is still set: the same node at its new version)
                }
                if (wg_load(&s_done) == e) __builtin_amdgcn_s_setprio(3);
                lap(10);
                int32_t status = kCommitOk;
                uint64_t free0 = 0, free1 = 0;                            // the sockets' free sets as this pod found them (stage 2 picks from them)
                const bool smt_node = (st.p2.flags & NHDFIT_NF_SMT) != 0;
                const bool retire = post_and_wait(v, ver, [&]() {        // stage 1 of the commit is computed while the sequencer validates
                    if (ver == 0) {                                       // (a never-touched node: its first-touch copy is taken at retirement)
                        if (lane < sizeof(NodeState) / 4) reinterpret_cast<uint32_t*>(&s_pst[sp])[lane] = reinterpret_cast<const uint32_t*>(&st)[lane];
                        if (lane < sizeof(nhdfit_detail) / 4) reinterpret_cast<uint32_t*>(&s_pdet[sp])[lane] = reinterpret_cast<const uint32_t*>(&dd)[lane];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) { res.node = (int64_t)a.global_base + (int64_t)v; res.map = mp; }
                    lap(11);
                    // what the next pod's verification reads of this commit: counts, hugepages, NIC classes, signature ids (commit_summary_wave)
                    status = kTuning && (q.dbg & 2) ? kCommitOk : commit_summary_wave(st, dd, rq, mp, a.now, sigs, q.ncls, lane, free0, free1);
                    lap(3);
                });
                lap(4);
                if (!retire) continue;                                    // the node moved on since it was read: again, from this node
                // retired: the new state goes live in the block's cache - the next pod on this node starts from here - then everything else
                if (lane == 0) { s_cver[ce] = ver + 1u; wg_store(&s_ctag[ce], v); wg_store(&s_examv[sp], kNoNode); }
                ++cache_next;
                if (lane == 0) wg_store(&s_ctag[sp * kSpecCache + cache_next % kSpecCache], kNoNode);     // the entry worked in next
                // stage 2, off the chain: the batches' picks (placement record) and thread 1's bits
                __builtin_amdgcn_s_setprio(1);
                uint64_t clear0 = 0, clear1 = 0;
                if (!(kTuning && (q.dbg & 2))) {
                    const int st2 = commit_picks_wave(free0, free1, smt_node, rq, mp, pl, lane, clear0, clear1);
                    if (st2 == kCommitWouldRaise) status = kCommitWouldRaise;
                }
                if (lane == 0) { res.status = status; pl.status = (uint8_t)status; }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                lap(6);
                if (status == kCommitNewSig && lane == 0) q.flags[1] = 1u;
                // the mirror takes version ver + 1 behind version ver (whose writer may still be in its own stage 2)
                for (uint32_t spin = 0; ver != 0 && (__hip_atomic_load(&q.mat[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ~kPubPoison) < ver && !stop; ++spin) {
                    if (spin > kSpinLimit) give_up();
                    if (wg_load(&s_abort)) stop = true;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (ver != 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (one acquire behind the match, not one per poll)
                if (stop) break;
                if (!(kTuning && (q.dbg & 4))) {
                    if (lane < sizeof(SeqResult) / 4) reinterpret_cast<uint32_t*>(&a.out[mine])[lane] = reinterpret_cast<const uint32_t*>(&res)[lane];
                    if (a.place && lane < sizeof(nhdfit_placement) / 4) reinterpret_cast<uint32_t*>(&a.place[mine])[lane] = reinterpret_cast<const uint32_t*>(&pl)[lane];
                    if (ver == 0 && !(kTuning && (q.dbg & 1))) note_first_touch(a, v, s_pst[sp], s_pdet[sp], lane, true);
                    store_node_lds_chain(a, v, &st, &dd, lane, clear0, clear1);
                }
                __threadfence();
                if (lane == 0) __hip_atomic_store(&q.mat[v], (status == kCommitNewSig ? kPubPoison : 0u) | (ver + 1u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                for (uint32_t k0 = 0; k0 < ngl; k0 += patch_span(ngl)) push(kItemValid | kItemPatch | ((unsigned long long)k0 << 32) | v);
                if (status == kCommitNewSig) give_up();
                placed = true;
                lap(5);
            }
            if (lane == 0) wg_store(&s_examv[sp], kNoNode);
            if (!placed && !stop) {                                       // no node takes the pod - at no later version either
                (void)post_and_wait(kNoNode, 0u, [] {});
                if (!stop) not_placed(mine);
                lap(4);
            }
        }
        if (lane == 0) {
            wg_store(&s_pende[sp], 0xFFFFFFFFu);
            atomicAdd(&s_cnt[0], c_fail); atomicAdd(&s_cnt[1], c_hit); atomicAdd(&s_cnt[2], c_pub); atomicAdd(&s_cnt[3], c_plain);
            atomicAdd(&s_cnt[4], c_chain); atomicAdd(&s_cnt[5], c_rescan);
            if (kTuning) for (int k = 0; k < 16; ++k) atomicAdd(&s_cnt[8 + k], (uint32_t)t_acc[k]);
            __hip_atomic_fetch_add(&s_spec_done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }

    // ---- wavefront 0: the sequencer - every pod, in the caller's order ------------------------------------------------------
    __builtin_amdgcn_s_setprio(3);
    uint32_t n_tn_done = 0, c_redo = 0;
    // tuning aid, 100 MHz ticks (ctrl[9..12]).  In the tuning build ONLY (round 6): a read of the real-time counter is a scalar memory
    // instruction, and the wait for it (lgkmcnt) is a wait for every LDS operation in flight as well - two to four of them per pod sat on
    // the sequencer's chain in the shipped library for nobody to read
    unsigned long long t_ready = 0, t_gpu = 0, t_post = 0, t_retire = 0, t_last = kTuning ? (unsigned long long)wall_clock64() : 0ull;
    auto lap = [&](unsigned long long& acc) { if (kTuning) { const unsigned long long t = wall_clock64(); acc += t - t_last; t_last = t; } };
    unsigned long long t_sub[4] = {0, 0, 0, 0}, t_sub_last = 0;      // tuning aid: the path of a pod with GPUs, finer (ctrl[28..30]): window + pick, take, push
    auto sub = [&](int k) { if (kTuning) { const unsigned long long t = wall_clock64(); if (k == 0) t_sub[0] += t - t_last; else t_sub[k] += t - t_sub_last; t_sub_last = t; } };
    auto take = [&](uint32_t v, bool gpu_pod) {
        if ((v >> 6) >= span) { give_up(); return; }                          // past the bit maps' reach (wave-uniform): the general kernel decides this batch
        if (lane == 0) {
            atomicOr(reinterpret_cast<unsigned long long*>(&s_taken[v >> 6]), 1ull << (v & 63));
            if (gpu_pod) atomicOr(reinterpret_cast<unsigned long long*>(&s_tgpu[v >> 6]), 1ull << (v & 63));
        }
    };
    for (uint32_t e = 0; e < n_pods && !stop; ++e) {
        const uint32_t mine = e;
        if (!(s_isn[e >> 5] >> (e & 31) & 1u)) {
            // a pod with GPUs (or an invalid request): first bit of its parked window nobody took since
            const uint32_t slot = e % kDecideRing;
            for (uint32_t spin = 0; wg_load(&s_ready[slot]) != e + 1 && !stop; ++spin) {
                if (spin > kSpinLimit) give_up();
                if (wg_load(&s_abort)) stop = true;
                __builtin_amdgcn_s_sleep(1);
            }
            if (stop) break;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            lap(t_ready);
            const uint32_t pos = s_pos[slot];
            int32_t have = s_have[slot];
            uint32_t wbase = s_base[slot];
            bool placed = false;
            while (have == 2 && !placed) {
                const uint32_t c = wbase + lane;
                const uint64_t w = s_win[slot][lane] & (c < a.chunks ? (c < span ? ~s_taken[c] : ~0ull) : 0ull);
                const uint64_t any = __ballot(w != 0);
                if (!any) {
                    const uint32_t nb = wbase + 64;
                    have = nb < a.chunks && scan_window(s_win[slot], pos, 3, nb, 0, wbase) ? 2 : 0;
                    continue;
                }
                const int l = __builtin_ctzll(any);
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w, l);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w >> 32), l);
                const uint32_t v = (wbase + (uint32_t)l) * 64u + (uint32_t)__builtin_ctzll(((uint64_t)hi << 32) | lo);
                sub(0);
                take(v, true);
                sub(1);
                push(kItemValid | ((unsigned long long)mine << 32) | v);
                sub(2);
                placed = true;
            }
            if (!placed) not_placed(mine);
            lap(t_gpu);
        } else {
            // a pod without GPUs: validate what its speculator found
            const uint32_t sp = n_tn_done % kSpecWaves;
            for (uint32_t attempt = 0; !stop; ++attempt) {
                const uint32_t id = ((e + 1u) << 4) | (attempt & 15u);
                for (uint32_t spin = 0; wg_load(&s_post[sp]) != id && !stop; ++spin) {
                    if (spin > kSpinLimit) give_up();
                    if (wg_load(&s_abort)) stop = true;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (stop) break;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                lap(t_post);
                const uint32_t v = s_rv[sp], ver = s_rd[sp];
                bool retire = true;
                if (v != kNoNode) {
                    // decisions on v so far, and where the next one goes in the multiset
                    uint32_t d = (v >> 6) < span ? (uint32_t)(s_tgpu[v >> 6] >> (v & 63)) & 1u : 0u, at = 0;
                    for (uint32_t h = hash_of(v);; h += 64) {
                        const uint32_t key = s_hash[(h + lane) & hmask];
                        const uint64_t eq = __ballot(key == v), em = __ballot(key == kNoNode);
                        if (em) { const uint32_t f = (uint32_t)__builtin_ctzll(em); d += (uint32_t)__popcll(eq & ((1ull << f) - 1ull)); at = (h + f) & hmask; break; }
                        d += (uint32_t)__popcll(eq);
                    }
                    retire = d == ver;
                    if (retire) {
                        if (lane == 0) __hip_atomic_store(&s_hash[at], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        take(v, false);                                   // busy for every later pod with GPUs
                    } else ++c_redo;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) wg_store(&s_verd[sp], (id << 1) | (retire ? 1u : 0u));
                lap(t_retire);
                if (retire) break;
            }
            ++n_tn_done;
        }
        if (lane == 0) wg_store(&s_done, e + 1);
    }
    for (uint32_t spin = 0; wg_load(&s_spec_done) < (uint32_t)kSpecWaves && spin < kSpinLimit && !wg_load(&s_abort); ++spin) __builtin_amdgcn_s_sleep(4);   // the speculators have pushed their last items
    if (lane == 0) {
        q.ctrl[4] = s_cnt[0]; q.ctrl[5] = s_cnt[1]; q.ctrl[6] = s_cnt[2]; q.ctrl[7] = s_cnt[3]; q.ctrl[8] = s_cnt[4]; q.ctrl[14] = s_cnt[5]; q.ctrl[15] = c_redo;
        q.ctrl[9] = (uint32_t)t_ready; q.ctrl[10] = (uint32_t)t_gpu; q.ctrl[11] = (uint32_t)t_post; q.ctrl[12] = (uint32_t)t_retire;
        if (kTuning) for (int k = 0; k < 16; ++k) q.ctrl[16 + k] = s_cnt[8 + k];
        if (kTuning) for (int k = 0; k < 3; ++k) q.ctrl[28 + k] = (uint32_t)t_sub[k];      // (t_acc[12..14] of the speculators are not in use)
        wg_store(&s_done, n_pods);                                        // the fetchers run out
        const uint32_t n_items = wg_load(&s_nitems);
        __hip_atomic_store(&q.ctrl[1], (n_items < q.queue_len ? n_items : q.queue_len) + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // the workers leave once the queue is drained
    }
}

// ---- the commit step for one placement (nhdfit_commit) --------------------------------------------------------------------------
// One wavefront: the node's five planes and its detail record are copied to LDS by sixteen lanes at once, the commit runs in its
// wavefront form (commit_node_wave above: a batch of cores is one ballot, not a loop over bits - the form mode B's workers run, held
// to the scalar form on emulated lanes, tests/test_wave_commit_emulation.py), the new state goes back by the same lanes and the
// placement record crosses the link as 64 four-byte stores behind ONE wait (round 6: the scalar form on one lane took 12.2 us per
// launch, profiles/r06 - a third of a pod's share of the scheduler's pod-at-a-time loop, nhd/NHDScheduler.py:289-304).  A commit the
// reference would raise on (nhd/Node.py:700-704 and the IndexError paths behind it) is rare and takes the scalar form, whose partial
// state is the documented one.
__global__ __launch_bounds__(64) void k_commit(CommitArgs a) {
    __shared__ NodeState s_st;
    __shared__ nhdfit_detail s_dd;
    __shared__ nhdfit_placement s_pl;
    const uint32_t lane = threadIdx.x;
    {
        uint32_t* st = reinterpret_cast<uint32_t*>(&s_st);
        if (lane < 5) {
            const uint4 q = lane == 0 ? *reinterpret_cast<const uint4*>(a.p0 + a.node) : lane == 1 ? *reinterpret_cast<const uint4*>(a.p1 + a.node) :
                            lane == 2 ? *reinterpret_cast<const uint4*>(a.p2 + a.node) : lane == 3 ? *reinterpret_cast<const uint4*>(a.p3 + a.node) :
                                        *reinterpret_cast<const uint4*>(a.p4 + a.node);
            st[lane * 4 + 0] = q.x; st[lane * 4 + 1] = q.y; st[lane * 4 + 2] = q.z; st[lane * 4 + 3] = q.w;
        }
        if (lane >= 8 && lane < 8 + sizeof(nhdfit_detail) / 16) {
            const uint4 q = reinterpret_cast<const uint4*>(a.det + a.node)[lane - 8];
            uint32_t* dd = reinterpret_cast<uint32_t*>(&s_dd) + (lane - 8) * 4;
            dd[0] = q.x; dd[1] = q.y; dd[2] = q.z; dd[3] = q.w;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    auto publish = [&]() {                                         // the record in LDS -> the host block, the call's sequence number behind it
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < sizeof(nhdfit_placement) / 4) reinterpret_cast<uint32_t*>(&a.host->place)[lane] = reinterpret_cast<const uint32_t*>(&s_pl)[lane];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) __hip_atomic_store(&a.host->flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    bool nic_missing = false;
    for (uint32_t g = 0; g < a.req.n_groups; ++g)                  // GetNicObjFromIndex returns None: IndexError before anything
        nic_missing |= (uint32_t)a.map.nic_idx[g] >= s_dd.nic_cnt[a.map.nic_numa[g] & 1];   // of that group is touched (nhd/Node.py:700-704);
    if (nic_missing) {                                             // the mirror is left alone
        if (lane < sizeof(nhdfit_placement) / 4) reinterpret_cast<uint32_t*>(&s_pl)[lane] = 0u;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) s_pl.status = kCommitWouldRaise;
        publish();
        return;
    }
    const int status = commit_node_wave(s_st, s_dd, a.req, a.map, a.busy_time, a.sigs, a.ncls, s_pl, lane);
    if (status == kCommitWouldRaise) {                             // (every lane holds the same status)
        if (lane == 0) {
            NodeState s;
            s.p0 = a.p0[a.node]; s.p1 = a.p1[a.node]; s.p2 = a.p2[a.node]; s.p3 = a.p3[a.node]; s.p4 = a.p4[a.node];
            nhdfit_detail d = a.det[a.node];
            nhdfit_placement pl;
            memset(&pl, 0, sizeof pl);
            commit_node(s, d, a.req, a.map, a.busy_time, a.sigs, pl);
            s_st = s; s_dd = d; s_pl = pl;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    {
        const uint32_t* st = reinterpret_cast<const uint32_t*>(&s_st);
        if (lane < 5) {
            const uint4 q = make_uint4(st[lane * 4], st[lane * 4 + 1], st[lane * 4 + 2], st[lane * 4 + 3]);
            if (lane == 0) *reinterpret_cast<uint4*>(a.p0 + a.node) = q;
            else if (lane == 1) *reinterpret_cast<uint4*>(a.p1 + a.node) = q;
            else if (lane == 2) *reinterpret_cast<uint4*>(a.p2 + a.node) = q;
            else if (lane == 3) *reinterpret_cast<uint4*>(a.p3 + a.node) = q;
            else *reinterpret_cast<uint4*>(a.p4 + a.node) = q;
        }
        if (lane >= 8 && lane < 8 + sizeof(nhdfit_detail) / 16) {
            const uint32_t* dd = reinterpret_cast<const uint32_t*>(&s_dd) + (lane - 8) * 4;
            reinterpret_cast<uint4*>(a.det + a.node)[lane - 8] = make_uint4(dd[0], dd[1], dd[2], dd[3]);
        }
    }
    publish();
}
