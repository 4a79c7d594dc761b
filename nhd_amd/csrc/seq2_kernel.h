// seq2_kernel.h - mode B, the two-chain form: decisions under the scheduler's commit semantics without one CU walking all pods.
// Device code of libnhdfit.so; included by nhdfit.hip inside its anonymous namespace after seq_kernel.h.  gfx950 only.
//
// What the sequential loop (nhd/NHDScheduler.py:425-437: FindNode, then SetBusy / SetPhysicalIdsFromMapping /
// ClaimPodNICResources, nhd/NHDScheduler.py:289-304) couples, and what it does not:
//   * a pod that requests GPUs never fits a node without GPUs (GPU stage, nhd/Matcher.py:120-131), and a pod without GPUs
//     takes the first feasible node WITHOUT GPUs when there is one (SelectNode, nhd/Matcher.py:401-413).  So the nodes
//     without GPUs only ever change under the pods without GPUs ("chain A"), and - as long as every such pod finds a
//     GPU-less node or no node at all - the nodes with GPUs only under the pods with GPUs ("chain B");
//   * for a pod WITH GPUs a commit changes exactly one thing: the node is busy from then on (SetBusy, nhd/Node.py:843-850;
//     nhd/Matcher.py:107-111 drops busy nodes for GPU pods), i.e. gone - every other node still is what the snapshot's
//     verdict row says.  Chain B's decisions are therefore a first-fit over bitmaps (pod k takes the first bit of its
//     row no earlier pod took: k_pick_b, one block), and its mappings and commits touch pairwise different nodes: one
//     wavefront per pod, all at once, on the whole chip (k_commit_b);
//   * chain A is a true chain - consecutive GPU-less pods pile onto the same first GPU-less node until it is full, and
//     each must see what the one before left.  It is walked by ONE wavefront (k_chain_a's driver) whose per-pod latency
//     is what counts: the verdict rows are only hints there (a set bit is verified by mapping the pod against the node's
//     state as it is now; feasibility only ever shrinks, so a cleared bit stays right), requests and row windows are
//     fetched ahead by helper wavefronts, committed nodes stay in LDS, their columns are re-evaluated in the background.
// A pod without GPUs that finds no GPU-less node but has candidates among the GPU nodes ("leftover") couples the chains:
// the host then runs the general kernel (k_seq) over chain B's pods plus the leftovers instead of the fast path.
struct Seq2Args {
    SeqArgs s;
    const uint32_t* list;        // the chain's pods, caller's indices, ascending
    uint32_t n_list;
    uint32_t* assign;            // chain B: [n_list] local node index or ~0u
    uint32_t* flags;             // [0] leftovers seen (chain A), [1] a commit met a NIC state without a signature, [2] chain A: pods decided,
                                 // [3] chain A gave up waiting for a helper wavefront (never expected; the host starts over with k_seq)
    uint32_t lds_sigs, lds_states;   // chain A: stage the signature hash table / the set-state tables in LDS (they fit)
};
constexpr uint32_t kNoNode = 0xFFFFFFFFu;

// first-touch copy of a node for apply = 0 (whole wavefront), as in k_seq
__device__ __forceinline__ void note_first_touch(const SeqArgs& a, uint32_t v, const NodeState& st, const nhdfit_detail& dd, uint32_t lane) {
    int32_t seen = 0;
    if (lane == 0) seen = a.touched[v];
    seen = __builtin_amdgcn_readfirstlane(seen);
    if (seen >= 0) return;
    uint32_t slot = 0;
    if (lane == 0) { slot = atomicAdd(&a.counters[0], 1u); a.touched[v] = (int32_t)slot; }
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

// ---- chain B, decisions: first-fit over the snapshot's verdict rows and the taken bits ---------------------------------
// One block, kPickPods pods per round (one per wavefront): every wavefront scans its pod's row minus the taken nodes from
// the pod's snapshot winner on, up to the first window of 64 chunks holding a candidate; wavefront 0 then walks the
// round's pods in order - first bit of the window no earlier pod of the round took - and knocks each pick out of the
// later windows.  A pod whose window ran dry under the round's own picks starts the next round.
template <int kPickPods>
__global__ __launch_bounds__(64 * kPickPods) void k_pick_b(Seq2Args q) {
    const SeqArgs& a = q.s;
    __shared__ uint64_t s_win[kPickPods][64];
    __shared__ uint32_t s_base[kPickPods];
    __shared__ int32_t s_have[kPickPods];            // -2 past the end, 0 no candidate, 1 window parked
    __shared__ uint32_t s_pick[kPickPods];
    __shared__ uint32_t s_keep;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t i = 0;
    while (i < q.n_list) {
        const uint32_t mine = i + wave;
        int32_t have = -2;
        if (mine < q.n_list) {
            have = 0;
            const uint32_t pos = a.order[q.list[mine]];
            const unsigned long long score_a = a.score[pos];
            if (score_a) {
                const int64_t from = (int64_t)(NHDFIT_SCORE_INDEX(score_a) - a.global_base);
                for (uint32_t base = (uint32_t)(from >> 6); base < a.chunks && !have; base += 64) {
                    const uint32_t c = base + lane;
                    uint64_t w = 0;
                    if (c < a.chunks)
                        w = a.rows[(size_t)c * a.P + pos] & ~__hip_atomic_load(&a.taken[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (c == (uint32_t)(from >> 6)) w &= ~0ull << (from & 63);
                    if (__ballot(w != 0)) {
                        s_win[wave][lane] = w;
                        if (lane == 0) s_base[wave] = base;
                        have = 1;
                    }
                }
            }
        }
        if (lane == 0) s_have[wave] = have;
        __syncthreads();
        if (wave == 0) {
            const int32_t my_hv = lane < (uint32_t)kPickPods ? s_have[lane] : -2;
            const uint32_t my_base = lane < (uint32_t)kPickPods ? s_base[lane] : 0u;
            uint32_t keep = 0;
            for (; keep < (uint32_t)kPickPods; ++keep) {
                const int32_t hv = __builtin_amdgcn_readlane(my_hv, (int)keep);
                if (hv == -2) break;
                uint32_t nd = kNoNode;
                if (hv == 1) {
                    const uint64_t w = s_win[keep][lane];
                    const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)my_base, (int)keep);
                    const uint64_t any = __ballot(w != 0);
                    if (!any) break;                                      // ran dry under this round's picks: next round rescans
                    const int l = __builtin_ctzll(any);
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w, l);
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w >> 32), l);
                    nd = (base + (uint32_t)l) * 64u + (uint32_t)__builtin_ctzll(((uint64_t)hi << 32) | lo);
                    if (lane > keep && my_hv == 1) {                      // busy for the later pods of the round
                        const uint32_t idx = (nd >> 6) - my_base;
                        if (idx < 64u) s_win[lane][idx] &= ~(1ull << (nd & 63));
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                if (lane == 0) s_pick[keep] = nd;
            }
            if (lane == 0) s_keep = keep;
        }
        __syncthreads();
        const uint32_t keep = s_keep;
        if (tid < keep) {
            const uint32_t nd = s_pick[tid];
            q.assign[i + tid] = nd;
            if (nd != kNoNode) atomicOr(reinterpret_cast<unsigned long long*>(&a.taken[nd >> 6]), 1ull << (nd & 63));
        }
        __threadfence();                                                  // the taken bits are in L2 before the next scan
        __syncthreads();
        i += keep;
    }
}

// ---- chain B, mapping + commit: one wavefront per pod, every pod on a node of its own ----------------------------------
constexpr int kCommitWaves = 4;
__global__ __launch_bounds__(64 * kCommitWaves) void k_commit_b(Seq2Args q) {
    const SeqArgs& a = q.s;
    __shared__ PaddedReq s_req[kCommitWaves];
    __shared__ nhdfit_detail s_det[kCommitWaves];
    __shared__ NodeState s_st[kCommitWaves];
    __shared__ nhdfit_placement s_place[kCommitWaves];
    __shared__ SeqResult s_res[kCommitWaves];
    __shared__ Layout s_L[kWClasses];
    __shared__ double s_caps[NHDFIT_MAX_CLASSES];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < (uint32_t)kWClasses) s_L[tid] = a.L[tid];
    if (tid < NHDFIT_MAX_CLASSES) s_caps[tid] = a.caps[tid];
    __syncthreads();
    const uint32_t e = blockIdx.x * kCommitWaves + wave;
    if (e >= q.n_list) return;
    const uint32_t mine = q.list[e], v = q.assign[e];
    if (v == kNoNode) {
        if (lane == 0) { SeqResult r; r.node = -1; r.map = nhdfit_mapping{}; r.status = 0; a.out[mine] = r; }
        if (a.place && lane < sizeof(nhdfit_placement) / 4) reinterpret_cast<uint32_t*>(&a.place[mine])[lane] = 0u;
        return;
    }
    const uint32_t pos = a.order[mine];
    if (lane < sizeof(nhdfit_req) / 16) {
        const uint4 rq4 = reinterpret_cast<const uint4*>(a.reqs + pos)[lane];
        uint32_t* dst = reinterpret_cast<uint32_t*>(&s_req[wave]) + lane * 4;
        dst[0] = rq4.x; dst[1] = rq4.y; dst[2] = rq4.z; dst[3] = rq4.w;
    }
    load_node_lds(a, v, &s_st[wave], &s_det[wave], lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const nhdfit_req& rq = s_req[wave].r;
    NodeState& st = s_st[wave];
    nhdfit_detail& dd = s_det[wave];
    const uint32_t tile = pos >> 6;
    const uint32_t bits = nic_assignment_bits_wave(a.tabs + (size_t)tile * a.pitch, s_L[a.tile_wcls[tile]], pos & 63, rq.map_type == NHDFIT_MAP_PCI, st.p3, lane);
    nhdfit_mapping mp;
    const bool mapped = map_on_state_wave(rq, st, dd, s_caps, bits, a.mt, lane, mp);
    __builtin_amdgcn_wave_barrier();
    note_first_touch(a, v, st, dd, lane);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        SeqResult& res = s_res[wave];
        nhdfit_placement& pl = s_place[wave];
        res.node = (int64_t)a.global_base + (int64_t)v;
        res.map = mp;
        if (mapped) res.status = commit_node(st, dd, rq, res.map, a.now, a.sigs, pl);
        else {                                                            // the row said feasible, the mapping disagrees: cannot happen
            memset(&pl, 0, sizeof pl);
            res.map = nhdfit_mapping{};
            res.status = kCommitWouldRaise;
            pl.status = kCommitWouldRaise;
        }
        if (res.status == kCommitNewSig) q.flags[1] = 1u;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < sizeof(SeqResult) / 4) reinterpret_cast<uint32_t*>(&a.out[mine])[lane] = reinterpret_cast<const uint32_t*>(&s_res[wave])[lane];
    if (a.place && lane < sizeof(nhdfit_placement) / 4) reinterpret_cast<uint32_t*>(&a.place[mine])[lane] = reinterpret_cast<const uint32_t*>(&s_place[wave])[lane];
    store_node_lds(a, v, &st, &dd, lane);
}

// ---- chain A: the pods without GPUs over the nodes without GPUs, one after the other -----------------------------------
// Wavefront 0 (the driver) decides pod after pod; wavefronts 1 .. kFetchers fetch ahead (request record, first window of
// the pod's row over the GPU-less nodes); the rest re-evaluate committed nodes against the tiles that hold GPU-less pods and
// clear the bits of the pods that lost them ("columns", as in k_seq) - hints for the pods to come, never read for a decision:
// the driver verifies every candidate against the node's current state.
constexpr int kChainWaves = 16, kChainFetchers = 8, kChainRing = 16, kChainCache = 16, kPatchRing = 16;
struct PatchItem { uint32_t node, busy; NodeIdx ni; nhdfit_plane3 p3; };

__global__ __launch_bounds__(64 * kChainWaves) void k_chain_a(Seq2Args q) {
    const SeqArgs& a = q.s;
    __shared__ PaddedReq s_req[kChainRing];
    __shared__ uint64_t s_win[kChainRing][64];
    __shared__ uint32_t s_base[kChainRing], s_pos[kChainRing];
    __shared__ int32_t s_have[kChainRing];
    __shared__ uint32_t s_ready[kChainRing];                   // sequence number + 1 of the pod parked in the slot
    __shared__ uint32_t s_done;                                // pods the driver is through with
    __shared__ uint32_t s_abort;                               // a wait ran out (never expected): every wavefront leaves, the host falls back
    __shared__ NodeState s_cst[kChainCache];                   // nodes this batch committed to, most recent kChainCache
    __shared__ nhdfit_detail s_cdet[kChainCache];
    __shared__ uint32_t s_ctag[kChainCache];
    __shared__ NodeState s_st;                                 // scratch: a node fetched from global memory
    __shared__ nhdfit_detail s_det;
    __shared__ nhdfit_placement s_place;
    __shared__ SeqResult s_res;
    __shared__ PatchItem s_patch[kPatchRing];
    __shared__ uint32_t s_patch_head, s_patch_tail[kChainWaves];   // produced / consumed per patch wavefront
    __shared__ uint32_t s_ngl;
    constexpr uint32_t kGlLds = 256;
    __shared__ uint16_t s_gl[kGlLds];
    __shared__ Layout s_L[kWClasses];
    __shared__ double s_caps[NHDFIT_MAX_CLASSES];
    extern __shared__ __align__(16) uint8_t s_dyn[];           // modified-node bitmap [chunks] words, then the staged look-up tables
    uint8_t* dynp = s_dyn;
    uint64_t* s_mod = carve<uint64_t>(dynp, a.chunks);
    // what the driver's chain would otherwise fetch from L2 pod after pod (signature ids of the committed node, the ~40
    // dependent steps of the three-group set model): staged once
    SigTable sigs = a.sigs;
    MapTables mt = a.mt;
    uint64_t* l_skey = nullptr; uint32_t* l_sid = nullptr; uint64_t* l_info = nullptr; uint32_t* l_next = nullptr; uint32_t* l_asc = nullptr;
    if (q.lds_sigs) { l_skey = carve<uint64_t>(dynp, (size_t)a.sigs.mask + 1); l_sid = carve<uint32_t>(dynp, (size_t)a.sigs.mask + 1); }
    if (q.lds_states) { l_info = carve<uint64_t>(dynp, a.mt.st.n); l_next = carve<uint32_t>(dynp, (size_t)a.mt.st.n * 8); l_asc = carve<uint32_t>(dynp, 256); }
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t tiles = (a.P + kTile - 1) / kTile;
    constexpr uint32_t kPatchers = kChainWaves - 1 - kChainFetchers;
    if (tid == 0) { s_done = 0; s_patch_head = 0; s_ngl = 0; s_abort = 0; }
    constexpr uint32_t kSpinLimit = 1u << 22;                  // x ~100 cycles of s_sleep: a fraction of a second, then give up
    if (tid < kChainRing) s_ready[tid] = 0;
    if (tid < kChainCache) s_ctag[tid] = kNoNode;
    if (tid < kChainWaves) s_patch_tail[tid] = 0;
    if (tid < (uint32_t)kWClasses) s_L[tid] = a.L[tid];
    if (tid < NHDFIT_MAX_CLASSES) s_caps[tid] = a.caps[tid];
    for (uint32_t k = tid; k < a.chunks; k += 64 * kChainWaves) s_mod[k] = 0;
    if (q.lds_sigs) {
        for (uint32_t k = tid; k <= a.sigs.mask; k += 64 * kChainWaves) { l_skey[k] = a.sigs.key[k]; l_sid[k] = a.sigs.id[k]; }
        sigs = SigTable{l_skey, l_sid, a.sigs.mask};
    }
    if (q.lds_states) {
        for (uint32_t k = tid; k < a.mt.st.n; k += 64 * kChainWaves) l_info[k] = a.mt.st.info[k];
        for (uint32_t k = tid; k < a.mt.st.n * 8; k += 64 * kChainWaves) l_next[k] = a.mt.st.next[k];
        for (uint32_t k = tid; k < 256; k += 64 * kChainWaves) l_asc[k] = a.mt.st.asc[k];
        mt.st = SetStates{l_info, l_next, l_asc, a.mt.st.n};
    }
    __syncthreads();
    for (uint32_t t = tid; t < tiles; t += 64 * kChainWaves) {            // tiles that hold pods without GPUs
        const uint32_t live = a.P - t * kTile < (uint32_t)kTile ? a.P - t * kTile : (uint32_t)kTile;
        const uint64_t lm = live == 64 ? ~0ull : (1ull << live) - 1;
        if (~a.tile_masks[2 * t] & lm) {
            const uint32_t at = atomicAdd(&s_ngl, 1u);
            a.gl_tiles[at] = (uint16_t)t;
            if (at < kGlLds) s_gl[at] = (uint16_t)t;
        }
    }
    __threadfence_block();
    __syncthreads();
    const uint32_t ngl = s_ngl;
    const uint16_t* gl_tiles = ngl <= kGlLds ? s_gl : a.gl_tiles;
    auto lds_load = [](const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto lds_store = [](uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); };

    // first window of pod `pos`'s row over the GPU-less nodes at or after chunk `from_chunk` -> slot; returns have
    auto scan_window = [&](uint32_t slot, uint32_t pos, uint32_t from_chunk, uint32_t from_bit, uint32_t& base_out) -> int32_t {
        for (uint32_t base = from_chunk; base < a.chunks; base += 64) {
            const uint32_t c = base + lane;
            uint64_t w = 0;
            if (c < a.chunks) w = __hip_atomic_load(&a.rows[(size_t)c * a.P + pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & a.nogpu[c];
            if (c == from_chunk) w &= ~0ull << from_bit;
            if (__ballot(w != 0)) {
                s_win[slot][lane] = w;
                base_out = base;
                return 1;
            }
        }
        return 0;
    };

    if (wave >= 1 && wave <= (uint32_t)kChainFetchers) {
        // ---- fetchers: pod e goes to slot e % kChainRing once the driver is past pod e - kChainRing
        for (uint32_t e = wave - 1; e < q.n_list; e += kChainFetchers) {
            const uint32_t slot = e % kChainRing;
            for (uint32_t spin = 0; e >= lds_load(&s_done) + kChainRing; ++spin) {
                if (spin > kSpinLimit || lds_load(&s_abort)) return;
                __builtin_amdgcn_s_sleep(2);
            }
            const uint32_t pos = a.order[q.list[e]];
            if (lane < sizeof(nhdfit_req) / 16) {
                const uint4 v = reinterpret_cast<const uint4*>(a.reqs + pos)[lane];
                uint32_t* dst = reinterpret_cast<uint32_t*>(&s_req[slot]) + lane * 4;
                dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
            }
            int32_t have = 0;
            const unsigned long long score_a = a.score[pos];
            if (score_a >> 63) {                                          // the snapshot had a GPU-less candidate: start there
                const int64_t from = (int64_t)(NHDFIT_SCORE_INDEX(score_a) - a.global_base);
                uint32_t wb = 0;
                have = scan_window(slot, pos, (uint32_t)(from >> 6), (uint32_t)(from & 63), wb);
                if (lane == 0) s_base[slot] = wb;
            }
            if (lane == 0) { s_have[slot] = have; s_pos[slot] = pos; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) lds_store(&s_ready[slot], e + 1);
        }
        return;
    }
    if (wave > (uint32_t)kChainFetchers) {
        // ---- patchers: committed node x tiles with GPU-less pods, sixteen lanes per (node, tile)
        const uint32_t me = wave - 1 - kChainFetchers;                    // 0 .. kPatchers-1: items me, me + kPatchers, ...
        const uint32_t p = lane & 15u, grp = lane >> 4;
        for (uint32_t item = me;; item += kPatchers) {
            for (uint32_t spin = 0; lds_load(&s_patch_head) <= item; ++spin) {
                // the driver publishes its last item before it reports the last pod: done first, then the head once more
                if (lds_load(&s_done) >= q.n_list && lds_load(&s_patch_head) <= item) return;
                if (spin > 64u * kSpinLimit || lds_load(&s_abort)) return;
                __builtin_amdgcn_s_sleep(4);
            }
            const PatchItem it = s_patch[item % kPatchRing];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            if (lane == 0) lds_store(&s_patch_tail[wave], item + 1);      // copied: the driver may reuse the ring slot
            for (uint32_t k0 = 0; k0 < ngl; k0 += 4) {
                const uint32_t k = k0 + grp;
                uint64_t lost = 0;
                uint32_t t = 0;
                if (k < ngl) {
                    t = gl_tiles[k];
                    const uint64_t need = a.tile_masks[2 * t];
                    const uint8_t* img = a.tabs + (size_t)t * a.pitch;
                    const Layout& L = s_L[a.tile_wcls[t]];
                    uint64_t term = p < L.W ? node_term_cold(img, L, it.ni, it.p3, a.tile_masks[2 * t + 1], p) : 0ull;
                    for (int m = 1; m < 16; m <<= 1) term |= __shfl_xor(term, m, 16);
                    lost = ~(term & node_pred_cold(img, L, it.ni, it.busy != 0, need)) & ~need;
                }
                for (uint32_t qq = 0; qq < 4; ++qq) {
                    const uint32_t j = qq * 16 + p;
                    if ((lost >> j & 1) && (size_t)t * 64 + j < a.P)
                        atomicAnd(reinterpret_cast<unsigned long long*>(&a.rows[(size_t)(it.node >> 6) * a.P + (size_t)t * 64 + j]), ~(1ull << (it.node & 63)));
                }
            }
        }
    }

    // ---- the driver ----------------------------------------------------------------------------------------------------
    __builtin_amdgcn_s_setprio(3);
    uint32_t n_commit = 0, cache_next = 0;
    bool stop = false;
    for (uint32_t e = 0; e < q.n_list && !stop; ++e) {
        const uint32_t slot = e % kChainRing, mine = q.list[e];
        for (uint32_t spin = 0; lds_load(&s_ready[slot]) != e + 1 && !stop; ++spin) {
            if (spin > kSpinLimit) { stop = true; if (lane == 0) { q.flags[3] = 1u; lds_store(&s_abort, 1u); } }
            __builtin_amdgcn_s_sleep(1);
        }
        if (stop) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const nhdfit_req& rq = s_req[slot].r;
        const uint32_t pos = s_pos[slot], tile = pos >> 6;
        int32_t have = s_have[slot];
        uint32_t wbase = have == 1 ? s_base[slot] : 0u;
        bool placed = false;
        while (have == 1 && !placed) {
            // first candidate of the window
            const uint64_t w = s_win[slot][lane];
            const uint64_t any = __ballot(w != 0);
            if (!any) {                                                   // window exhausted: the next one
                const uint32_t nb = wbase + 64;
                have = nb < a.chunks ? scan_window(slot, pos, nb, 0, wbase) : 0;
                continue;
            }
            const int l = __builtin_ctzll(any);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w, l);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w >> 32), l);
            const uint32_t v = (wbase + (uint32_t)l) * 64u + (uint32_t)__builtin_ctzll(((uint64_t)hi << 32) | lo);
            // the node as it is now: LDS if this batch committed to it recently, else global memory
            NodeState* st = &s_st;
            nhdfit_detail* dd = &s_det;
            int cidx = -1;
            const bool modified = (s_mod[v >> 6] >> (v & 63) & 1) != 0;
            if (modified) {
                const uint64_t hit = __ballot(lane < (uint32_t)kChainCache && s_ctag[lane] == v);
                if (hit) cidx = __builtin_ctzll(hit);
            }
            if (cidx >= 0) { st = &s_cst[cidx]; dd = &s_cdet[cidx]; }
            else {
                if (modified) { __threadfence(); load_node_lds_coherent(a, v, &s_st, &s_det, lane); }   // evicted from LDS: its store was this
                else load_node_lds(a, v, &s_st, &s_det, lane);                                        // wavefront's own, now past L1
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            // verify: hugepages, then the mapping against the current state (the row bit is a hint)
            bool ok = rq.hugepages_gb <= st->p2.hp_free;                  // nhd/Matcher.py:78
            nhdfit_mapping mp = nhdfit_mapping{};
            if (ok) {
                const uint32_t bits = nic_assignment_bits_wave(a.tabs + (size_t)tile * a.pitch, s_L[a.tile_wcls[tile]], pos & 63,
                                                               rq.map_type == NHDFIT_MAP_PCI, st->p3, lane);
                ok = map_on_state_wave(rq, *st, *dd, s_caps, bits, mt, lane, mp);
            }
            __builtin_amdgcn_wave_barrier();
            if (!ok) {                                                    // stale hint: not this node (any more)
                if (lane == (uint32_t)l) s_win[slot][lane] = w & ~(1ull << (v & 63));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                continue;
            }
            // commit: into an LDS cache entry of its own
            if (cidx < 0) {
                cidx = (int)(cache_next++ % kChainCache);
                const uint32_t* src = reinterpret_cast<const uint32_t*>(&s_st);
                if (lane < sizeof(NodeState) / 4) reinterpret_cast<uint32_t*>(&s_cst[cidx])[lane] = src[lane];
                if (lane < sizeof(nhdfit_detail) / 4) reinterpret_cast<uint32_t*>(&s_cdet[cidx])[lane] = reinterpret_cast<const uint32_t*>(&s_det)[lane];
                if (lane == 0) s_ctag[cidx] = v;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                st = &s_cst[cidx]; dd = &s_cdet[cidx];
            }
            if (!modified) {
                note_first_touch(a, v, *st, *dd, lane);
                if (lane == 0) s_mod[v >> 6] |= 1ull << (v & 63);
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                s_res.node = (int64_t)a.global_base + (int64_t)v;
                s_res.map = mp;
                s_res.status = commit_node(*st, *dd, rq, s_res.map, a.now, sigs, s_place);
                if (s_res.status == kCommitNewSig) q.flags[1] = 1u;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < sizeof(SeqResult) / 4) reinterpret_cast<uint32_t*>(&a.out[mine])[lane] = reinterpret_cast<const uint32_t*>(&s_res)[lane];
            if (a.place && lane < sizeof(nhdfit_placement) / 4) reinterpret_cast<uint32_t*>(&a.place[mine])[lane] = reinterpret_cast<const uint32_t*>(&s_place)[lane];
            store_node_lds(a, v, st, dd, lane);
            if (s_res.status == kCommitNewSig) stop = true;
            // its columns, for the pods to come
            if (ngl) {
                const uint32_t item = n_commit;
                for (uint32_t spin = 0; item >= kPatchRing; ++spin) {     // ring slot free once its previous item was copied out:
                    const uint32_t old = item - kPatchRing;               // item x is consumed by patcher x % kPatchers, in order
                    if (lds_load(&s_patch_tail[1 + kChainFetchers + old % kPatchers]) > old) break;
                    if (spin > kSpinLimit) { stop = true; if (lane == 0) { q.flags[3] = 1u; lds_store(&s_abort, 1u); } break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (lane == 0) {
                    PatchItem& pi = s_patch[item % kPatchRing];
                    pi.node = v;
                    pi.busy = (a.now - st->p4.busy_time) < kMinBusySecs ? 1u : 0u;
                    pi.ni = node_index(st->p0, st->p1, st->p2, st->p4, a.fc_dim, a.fg_dim, a.ngs);
                    pi.p3 = st->p3;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) lds_store(&s_patch_head, item + 1);
            }
            ++n_commit;
            placed = true;
        }
        if (!placed) {
            // no GPU-less node for this pod.  Candidates among the nodes with GPUs couple the chains: report, the host reruns
            // the general kernel for chain B and these pods
            bool left = false;
            const unsigned long long score_a = a.score[pos];
            if (score_a) {
                for (uint32_t base = 0; base < a.chunks && !left; base += 64) {
                    const uint32_t c = base + lane;
                    const uint64_t w = c < a.chunks ? (a.rows[(size_t)c * a.P + pos] & ~a.nogpu[c]) : 0ull;
                    if (__ballot(w != 0)) left = true;
                }
            }
            if (left) { if (lane == 0) q.flags[0] = 1u; }
            if (lane == 0) { SeqResult r; r.node = left ? -2 : -1; r.map = nhdfit_mapping{}; r.status = 0; a.out[mine] = r; }
            if (a.place && lane < sizeof(nhdfit_placement) / 4) reinterpret_cast<uint32_t*>(&a.place[mine])[lane] = 0u;
        }
        if (lane == 0) lds_store(&s_done, e + 1);
    }
    if (lane == 0) {
        q.flags[2] = stop ? s_done : q.n_list;                            // pods of the list decided (fewer after a stop)
        lds_store(&s_done, q.n_list);                                     // the fetchers run out, the patchers leave once nothing is queued
    }
}
