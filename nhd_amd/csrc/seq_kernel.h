// seq_kernel.h - mode B on the device: sequential-commit batch kernel (k_seq), undo, single commit (k_commit), deltas (k_delta).
// Device code of libnhdfit.so; included by nhdfit.hip inside its anonymous namespace, in this order: step_digest.h,
// step_fit.h, step_map.h, step_kernel.h, seq_kernel.h (one translation unit: the roles are fused into one kernel).
// gfx950 only.
// ---- mode B: sequential commit on the device (seq_core.h) ---------------------------------------------
__global__ __launch_bounds__(64) void k_nogpu(const nhdfit_plane2* __restrict__ p2, uint32_t n, uint64_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    const uint64_t w = __ballot(i < n && !(p2[i].flags & NHDFIT_NF_HAS_GPU));
    if (threadIdx.x == 0) out[blockIdx.x] = w;
}

// per tile: pods that request GPUs / are in PCI mode (node_word_cold's masks)
__global__ __launch_bounds__(64) void k_tile_masks(const PodHeader* __restrict__ hdr, uint32_t tiles, uint64_t* __restrict__ out) {
    const PodHeader h = hdr[blockIdx.x * 64 + threadIdx.x];
    const uint64_t need = __ballot((h.flags & kPodNeedGpu) != 0), pci = __ballot((h.flags & kPodPci) != 0);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = need; out[2 * blockIdx.x + 1] = pci; }
}

// word `c` of the verdict row of the pod staged at position `pos`
#define NHDFIT_ROW(a, c, pos) ((a).rows[(size_t)(pos) * (a).chunks + (c)])

struct UndoRec { uint32_t node, pad[3]; NodeState st; nhdfit_detail d; };

struct SeqArgs {
    nhdfit_plane0* p0; nhdfit_plane1* p1; nhdfit_plane2* p2; nhdfit_plane3* p3; nhdfit_plane4* p4; nhdfit_detail* det;   // the mirror (modified)
    uint32_t n, chunks; uint64_t global_base; double now;
    const nhdfit_req* reqs; const unsigned long long* score; uint32_t P;
    const uint32_t* order;           // caller's pod i -> staged (class-sorted) position
    const uint32_t* list; uint32_t n_list;   // optional: the pods to decide (caller's indices, ascending) instead of all P
    const uint8_t* tabs; uint32_t pitch; const uint8_t* tile_wcls; Layout L[kWClasses];
    uint64_t* rows;                  // [P][chunks] verdict rows of the snapshot (k_rows_t: a pod's window of 64 chunks is 512 contiguous bytes -
                                     // with [chunks][P] it was 64 cache lines on 64 pages); kept current for the pods without GPUs
    uint64_t* taken;                 // [chunks] nodes that received a pod of this batch: busy, i.e. gone for every pod with GPUs
    const uint64_t* nogpu;           // [chunks] nodes without a GPU installed
    const uint64_t* tile_masks;      // [tiles][2]: pods that request GPUs / are in PCI mode
    const double* caps; SigTable sigs; uint32_t fc_dim, fg_dim, ngs;
    MapTables mt;
    UndoRec* undo; int32_t* touched; uint32_t* counters;     // first-touch copies (apply = 0), [n] -1 / slot, [0] = undo records
    SeqResult* out; nhdfit_placement* place;                  // [P], caller's order
    uint32_t* n_done; uint16_t* gl_tiles;          // scratch: [tiles] the tiles that hold pods without GPUs
    uint32_t lds_tables;
    unsigned long long* prof;        // tuning aid (NHDFIT_SEQ_PROF): ticks (100 MHz) per phase, rounds, pods
    uint32_t keep_undo;
};

// ---- wave-cooperative forms of the mapping arithmetic (winner_map.h), for the sequential kernel ---------------
// One lane working through candidate_masks / first_nic_choice / nic_assignment_bits costs ~15 us per pod - the whole
// wavefront is there, so every tuple code / NIC choice / table row gets a lane.  Same arithmetic, same order of the
// f64 subtractions; the host twin and the mode-A roles keep the scalar forms (tests compare both).
__device__ __forceinline__ uint32_t nic_assignment_bits_wave(const uint8_t* img, const Layout& L, uint32_t col, bool pci, const nhdfit_plane3& q3, uint32_t lane) {
    const uint32_t o0 = L.off_r0 + (pci ? q3.sig_pci[0] : q3.sig_numa[0]) * L.row;
    const uint32_t o1 = L.off_r1 + (pci ? q3.sig_pci[1] : q3.sig_numa[1]) * L.row;
    const bool ok = lane < L.W && ((ld64(img, o0 + lane * 8) & ld64(img, o1 + lane * 8)) >> col & 1);
    return (uint32_t)__ballot(ok);
}
__device__ __forceinline__ void candidate_masks_wave(const nhdfit_req& r, const WinnerState& w, uint32_t lane, uint32_t& sg_mask, uint32_t& sc_mask) {
    const int G = (int)r.n_groups, U = w.U;
    const uint32_t nG = ipow(U, G), nC = ipow(U, G + 1);
    // the request lives in LDS and the wavefront walks this chain alone: its operands are read ONCE, as four independent loads
    // (16-bit fields, four groups per 64-bit word), not field by field inside the loops - each of those a round trip with a wait
    static_assert(kMaxG == 4 && offsetof(nhdfit_req, gpus) % 8 == 0 && offsetof(nhdfit_req, cpu_smt) % 8 == 0 && offsetof(nhdfit_req, cpu_nosmt) % 8 == 0 &&
                  offsetof(nhdfit_req, misc_nosmt) == offsetof(nhdfit_req, misc_smt) + 2 && offsetof(nhdfit_req, misc_smt) % 4 == 0, "packed reads of the request");
    const uint64_t w_gpus = *reinterpret_cast<const uint64_t*>(r.gpus);
    const uint64_t w_cpu = w.smt ? *reinterpret_cast<const uint64_t*>(r.cpu_smt) : *reinterpret_cast<const uint64_t*>(r.cpu_nosmt);
    const uint32_t w_misc = *reinterpret_cast<const uint32_t*>(&r.misc_smt);
    const uint32_t misc = w.smt ? (w_misc & 0xFFFFu) : (w_misc >> 16);
    bool okg = false, okc = false;
    if (lane < nG) {
        uint32_t t0 = 0, t1 = 0;
        for (int g = 0; g < G; ++g) { const uint32_t d = (uint32_t)(w_gpus >> (16 * g)) & 0xFFFFu; if (tup_digit(lane, G, U, g)) t1 += d; else t0 += d; }
        okg = t0 <= (uint32_t)w.free_g[0] && t1 <= (uint32_t)w.free_g[1];
    }
    if (lane >= 32 && lane - 32 < nC) {
        const uint32_t code = lane - 32;
        uint32_t t0 = 0, t1 = 0;
        for (int g = 0; g <= G; ++g) {
            const uint32_t d = g < G ? (uint32_t)(w_cpu >> (16 * g)) & 0xFFFFu : misc;
            if (tup_digit(code, G + 1, U, g)) t1 += d; else t0 += d;
        }
        okc = t0 <= (uint32_t)w.free_c[0] && t1 <= (uint32_t)w.free_c[1];
    }
    sg_mask = (uint32_t)__ballot(okg);
    sc_mask = (uint32_t)(__ballot(okc) >> 32);
}
// first_nic_choice: lane = position in the reference's enumeration order (an odometer whose most significant digits are
// the NUMA-0 groups in ascending order, then the NUMA-1 groups; last digit fastest), 64 positions per pass
// (the choice comes back nibble-packed, group g in nibble g: a mapping indexed by a run-time group number would live in scratch
// memory - a memory round trip per access on the chain of the sequential kernels)
__device__ __forceinline__ bool first_nic_choice_wave(const nhdfit_req& r, const WinnerState& w, uint32_t gcode, bool pci, uint32_t lane, uint32_t& nic_nibbles) {
    const int G = (int)r.n_groups;
    // wave-uniform operands read ONCE (request and detail live in LDS: a read inside the loops below is a round trip with a wait
    // on a chain this wavefront walks alone): the two NIC counts, the four groups' rx / tx - picked by run-time group with selects
    static_assert(kMaxG == 4, "sel4 picks among four groups");
    const uint32_t cnt0 = w.d->nic_cnt[0], cnt1 = w.d->nic_cnt[1];
    const double rx0 = r.rx[0], rx1 = r.rx[1], rx2 = r.rx[2], rx3 = r.rx[3];
    const double tx0 = r.tx[0], tx1 = r.tx[1], tx2 = r.tx[2], tx3 = r.tx[3];
    auto sel4 = [](int h, double a0, double a1, double a2, double a3) { return h == 0 ? a0 : h == 1 ? a1 : h == 2 ? a2 : a3; };
    uint32_t order = 0, numa = 0;
    int n = 0;
    for (int u = 0; u < w.U; ++u)
        for (int g = 0; g < G; ++g)
            if (tup_digit(gcode, G, w.U, g) == u) { order = nib_set(order, n, (uint32_t)g); numa |= (uint32_t)u << g; ++n; }
    uint32_t total = 1;
    for (int g = 0; g < G; ++g) {
        const uint32_t k = (numa >> g) & 1 ? cnt1 : cnt0;
        if (k == 0) return false;
        total *= k;
    }
    for (uint32_t base = 0; base < total; base += 64) {
        uint32_t rem = base + lane, pick = 0;
        const bool live = rem < total;
        for (int pos = G - 1; pos >= 0; --pos) {
            const int g = (int)nib_get(order, pos);
            const uint32_t k = (numa >> g) & 1 ? cnt1 : cnt0;
            pick = nib_set(pick, g, rem % k);
            rem /= k;
        }
        bool ok = live;
        for (int g = 0; g < G && ok; ++g) {
            const uint32_t u = (numa >> g) & 1, k = nib_get(pick, g);
            bool first_on_nic = true;
            for (int h = 0; h < g; ++h)
                if (((numa >> h) & 1) == u && nib_get(pick, h) == k) first_on_nic = false;
            if (!first_on_nic) continue;
            double rx = w.caps[w.d->nic_cls[u][k]], tx = rx;                     // Matcher.py:261-263, group order
            for (int h = g; h < G; ++h)
                if (((numa >> h) & 1) == u && nib_get(pick, h) == k) { rx = rx - sel4(h, rx0, rx1, rx2, rx3); tx = tx - sel4(h, tx0, tx1, tx2, tx3); }
            if (rx < 0 || tx < 0) ok = false;                                    // Matcher.py:267
        }
        if (ok && pci) {                                                         // Matcher.py:312-322
            for (int g = 0; g < G && ok; ++g) {
                const uint32_t sw = w.d->nic_sw[(numa >> g) & 1][nib_get(pick, g)];
                uint32_t cnt = 0;
                for (int h = 0; h < G; ++h)
                    if (w.d->nic_sw[(numa >> h) & 1][nib_get(pick, h)] == sw) ++cnt;
                if (cnt > w.d->sw_free[sw]) ok = false;
            }
        }
        const uint64_t any = __ballot(ok);
        if (any) {
            nic_nibbles = (uint32_t)__builtin_amdgcn_readlane((int)pick, __builtin_ctzll(any));
            return true;
        }
    }
    return false;
}
// rare paths of the mapping, kept out of line: inlined, their scratch arrays (generic set model) and scalar-register
// spills (insertion-by-insertion model) would be paid by every pod of the sequential kernel
__device__ __noinline__ bool map_generic_cold(const nhdfit_req* r, const WinnerState* w, uint32_t codes, nhdfit_mapping* m) {
    return map_winner_t<GenericOps>(*r, *w, codes, *m);
}
__device__ __noinline__ uint32_t choose_model_cold(int G, int U, uint32_t sg, uint32_t sc, uint32_t cd, const AscEntry* asc) {
    uint32_t gcode = 0;
    int ccode = -1;
    const bool ok = choose_tuples<SmallOps>(G, U, sg, sc, cd, gcode, ccode, asc);
    return choose_result_word(ok, gcode, ccode);
}
// map_on_state (seq_core.h) with the parallel pieces; every lane returns the same mapping
// G4 = false: the caller knows that no pod of its batch has four processing groups - the generic set model (10 KB of private memory per
// lane, reserved for every wavefront of a kernel that merely CONTAINS the call) is not compiled in (k_decide<false>, round 6)
template <bool G4 = true>
__device__ __forceinline__ bool map_on_state_wave(const nhdfit_req& r, const NodeState& s, const nhdfit_detail& d, const double* caps, uint32_t nic_bits,
                                                  const MapTables& t, uint32_t lane, nhdfit_mapping& m) {
    const WinnerState w = state_view(s, d, caps);
    const int G = (int)r.n_groups, U = w.U;
    m = nhdfit_mapping{};
    const uint32_t codes = nic_codes_from_table_bits(nic_bits, G, U);
    if constexpr (!G4) { if (G > 3) return false; }
    if constexpr (G4) if (G > 3) {                                        // (copies: nothing the hot path keeps in registers has its address taken)
        WinnerState wc = w;
        nhdfit_mapping tmp = nhdfit_mapping{};
        const bool ok = map_generic_cold(&r, &wc, codes, &tmp);
        m = tmp;
        return ok;
    }
    uint32_t sg, sc;
    candidate_masks_wave(r, w, lane, sg, sc);
    const uint32_t cd = codes & ((1u << ipow(U, G)) - 1u);
    if (!sg || !sc || !cd) return false;
    uint32_t res;
    if (t.choose_tab && choose_tabulated(G, U)) res = choose_from_table(t.choose_tab, G, sg, sc, cd);
    else if (t.st.info && G == 3 && U == 2) res = choose_g3(t.st, t.asc, sg, sc, cd);
    else res = choose_model_cold(G, U, sg, sc, cd, t.asc);
    if (!(res >> 8 & 1)) return false;
    const uint32_t gcode = (res >> 4) & 7u;
    const int ccode = (int)(res & 15u);
    uint32_t nic_nibbles = 0;
    const bool nic_ok = first_nic_choice_wave(r, w, gcode, r.map_type == NHDFIT_MAP_PCI, lane, nic_nibbles);
    // every element written under a compile-time index (the struct stays in registers)
#pragma unroll
    for (int g = 0; g < kMaxG; ++g) {
        const bool in = g < G;
        m.gpu[g] = in ? (int8_t)tup_digit(gcode, G, U, g) : (int8_t)-1;
        m.nic_numa[g] = m.gpu[g];
        m.nic_idx[g] = in ? (int8_t)nib_get(nic_nibbles, g) : (int8_t)-1;
    }
#pragma unroll
    for (int g = 0; g <= kMaxG; ++g) m.cpu[g] = g <= G ? (int8_t)tup_digit((uint32_t)ccode, G + 1, U, g) : (int8_t)-1;
    m.valid = nic_ok ? 1 : 0;
    if (!nic_ok) m = nhdfit_mapping{};
    return nic_ok;
}

// One block walks the batch in the caller's order, kSeqPods pods per round (one per wavefront).
// What a commit changes for the pods that follow (nhd/NHDScheduler.py:289-304):
//   * SetBusy: the node is busy until now + 30 s, and a busy node is dropped for every pod that requests GPUs
//     (nhd/Matcher.py:107-111, nhd/Node.py:843-850) - for those pods the kernel keeps ONE bit per node ("taken") next
//     to the snapshot's verdict rows;
//   * for the pods without GPUs the node stays a candidate as far as its resources go: the committed nodes are
//     re-evaluated against the tiles that hold such pods (cold rows) and those pods' rows patched.
// Per round:
//   (1) every wavefront scans its pod's row (minus the taken nodes if the pod wants GPUs) up to the first window of 64
//       chunks with a candidate and parks the window's 64 words in LDS;
//   (2) wavefront 0 walks the round's pods in order.  A pod with GPUs gets the first bit of its window that no earlier
//       pod of the round took (those nodes are busy by the time it is the pod's turn, nothing else changed for it).  A
//       pod without GPUs gets the first bit of its window; if an earlier pod of the round took that very node, what
//       is left of the node decides - the round ends before this pod.  So does a pod whose window ran dry;
//   (3) one wavefront per kept pod: node record -> LDS, mapping against the node's state at this turn, commit
//       (commit_core.h), record and placement written back;
//   (4) all threads: the committed nodes against the tiles with GPU-less pods, sixteen lanes per (node, tile), which
//       then clear the node's bit in the rows of the pods that lost it.
template <int kSeqPods>
__global__ __launch_bounds__(64 * kSeqPods) void k_seq(SeqArgs a) {
    constexpr int kSeqThreads = 64 * kSeqPods;
    __shared__ PaddedReq s_req[kSeqPods];
    __shared__ nhdfit_detail s_det[kSeqPods];
    __shared__ NodeState s_st[kSeqPods];
    __shared__ uint64_t s_win[kSeqPods][64];
    __shared__ uint32_t s_base[kSeqPods];
    __shared__ int32_t s_have[kSeqPods];             // -2: past the end of the batch, 0: no candidate, 1: window parked, +2: pod wants GPUs
    __shared__ int64_t s_node[kSeqPods];
    __shared__ uint32_t s_pos[kSeqPods];
    __shared__ int32_t s_status[kSeqPods];
    __shared__ nhdfit_placement s_place[kSeqPods];
    __shared__ SeqResult s_res[kSeqPods];
    __shared__ uint32_t s_keep;
    __shared__ int32_t s_stop;
    __shared__ uint32_t s_ngl;                       // tiles that hold pods without GPUs
    constexpr uint32_t kGlLds = 256;
    __shared__ uint16_t s_gl[kGlLds];                // their list (a.gl_tiles when it is longer)
    __shared__ Layout s_L[kWClasses];                // kernel-argument arrays indexed at run time would live in scratch memory
    __shared__ double s_caps[NHDFIT_MAX_CLASSES];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t tiles = (a.P + kTile - 1) / kTile;
    if (tid == 0) { s_stop = 0; s_ngl = 0; }
    if (tid < (uint32_t)kWClasses) s_L[tid] = a.L[tid];
    if (tid < NHDFIT_MAX_CLASSES) s_caps[tid] = a.caps[tid];
    // small per-batch look-up data the chain would otherwise fetch from L2 pod after pod: staged in LDS once
    // (a.lds_tables = 0: the batch is too large, they stay in global memory)
    extern __shared__ __align__(16) uint8_t s_dyn[];
    const uint32_t* order = a.order;
    const uint64_t* tile_masks = a.tile_masks;
    const uint8_t* tile_wcls = a.tile_wcls;
    SigTable sigs = a.sigs;
    __syncthreads();
    for (uint32_t t = tid; t < tiles; t += kSeqThreads) {         // the order of the list does not matter
        const uint32_t live = a.P - t * kTile < (uint32_t)kTile ? a.P - t * kTile : (uint32_t)kTile;
        const uint64_t lm = live == 64 ? ~0ull : (1ull << live) - 1;
        if (~a.tile_masks[2 * t] & lm) {
            const uint32_t at = atomicAdd(&s_ngl, 1u);
            a.gl_tiles[at] = (uint16_t)t;
            if (at < kGlLds) s_gl[at] = (uint16_t)t;
        }
    }
    if (a.lds_tables) {
        uint8_t* q = s_dyn;
        uint64_t* l_masks = carve<uint64_t>(q, (size_t)tiles * 2);
        uint64_t* l_skey = carve<uint64_t>(q, (size_t)a.sigs.mask + 1);
        uint32_t* l_sid = carve<uint32_t>(q, (size_t)a.sigs.mask + 1);
        uint32_t* l_order = carve<uint32_t>(q, a.P);
        uint8_t* l_wcls = carve<uint8_t>(q, tiles);
        for (uint32_t k = tid; k < tiles * 2; k += kSeqThreads) l_masks[k] = a.tile_masks[k];
        for (uint32_t k = tid; k <= a.sigs.mask; k += kSeqThreads) { l_skey[k] = a.sigs.key[k]; l_sid[k] = a.sigs.id[k]; }
        for (uint32_t k = tid; k < a.P; k += kSeqThreads) l_order[k] = a.order[k];
        for (uint32_t k = tid; k < tiles; k += kSeqThreads) l_wcls[k] = a.tile_wcls[k];
        order = l_order; tile_masks = l_masks; tile_wcls = l_wcls;
        sigs = SigTable{l_skey, l_sid, a.sigs.mask};
    }
    __threadfence_block();
    __syncthreads();
    const uint32_t ngl = s_ngl;
    const uint16_t* gl_tiles = ngl <= kGlLds ? s_gl : a.gl_tiles;

    auto load_node = [&](uint32_t slot, uint32_t v) {             // planes + detail of node v -> LDS slot (one wavefront)
        uint32_t* st = reinterpret_cast<uint32_t*>(&s_st[slot]);
        if (lane < 5) {
            const uint4 q = lane == 0 ? *reinterpret_cast<const uint4*>(a.p0 + v) : lane == 1 ? *reinterpret_cast<const uint4*>(a.p1 + v) :
                            lane == 2 ? *reinterpret_cast<const uint4*>(a.p2 + v) : lane == 3 ? *reinterpret_cast<const uint4*>(a.p3 + v) :
                                        *reinterpret_cast<const uint4*>(a.p4 + v);
            st[lane * 4 + 0] = q.x; st[lane * 4 + 1] = q.y; st[lane * 4 + 2] = q.z; st[lane * 4 + 3] = q.w;
        }
        if (lane >= 8 && lane < 16) {
            const uint4 q = reinterpret_cast<const uint4*>(a.det + v)[lane - 8];
            uint32_t* dd = reinterpret_cast<uint32_t*>(&s_det[slot]) + (lane - 8) * 4;
            dd[0] = q.x; dd[1] = q.y; dd[2] = q.z; dd[3] = q.w;
        }
    };
    auto store_node = [&](uint32_t slot, uint32_t v) {
        const uint32_t* st = reinterpret_cast<const uint32_t*>(&s_st[slot]);
        if (lane < 5) {
            const uint4 q = make_uint4(st[lane * 4], st[lane * 4 + 1], st[lane * 4 + 2], st[lane * 4 + 3]);
            if (lane == 0) *reinterpret_cast<uint4*>(a.p0 + v) = q;
            else if (lane == 1) *reinterpret_cast<uint4*>(a.p1 + v) = q;
            else if (lane == 2) *reinterpret_cast<uint4*>(a.p2 + v) = q;
            else if (lane == 3) *reinterpret_cast<uint4*>(a.p3 + v) = q;
            else *reinterpret_cast<uint4*>(a.p4 + v) = q;
        }
        if (lane >= 8 && lane < 16) {
            const uint32_t* dd = reinterpret_cast<const uint32_t*>(&s_det[slot]) + (lane - 8) * 4;
            reinterpret_cast<uint4*>(a.det + v)[lane - 8] = make_uint4(dd[0], dd[1], dd[2], dd[3]);
        }
    };
    uint32_t i = 0;
    const uint32_t n_pods = a.list ? a.n_list : a.P;
    unsigned long long t_find = 0, t_pick = 0, t_map = 0, n_rounds = 0, tick = a.prof ? wall_clock64() : 0;
    unsigned long long t_sub[5] = {0, 0, 0, 0, 0}, sub = 0, t_c[5] = {0, 0, 0, 0, 0};
    auto sublap = [&](int k) { if (a.prof && wave == 0) { const unsigned long long t = wall_clock64(); t_sub[k] += t - sub; sub = t; } };
    auto lap = [&](unsigned long long& acc) { if (a.prof) { const unsigned long long t = wall_clock64(); acc += t - tick; tick = t; } };
    while (i < n_pods) {
        if (s_stop) break;
        // (1) wavefront w: pod i + w
        const uint32_t entry = i + wave;
        const uint32_t mine = entry < n_pods ? (a.list ? a.list[entry] : entry) : 0u;
        int32_t have = -2;
        if (entry < n_pods) {
            have = 0;
            const uint32_t pos = order[mine];
            if (lane < sizeof(nhdfit_req) / 16) {
                const uint4 v = reinterpret_cast<const uint4*>(a.reqs + pos)[lane];
                uint32_t* dst = reinterpret_cast<uint32_t*>(&s_req[wave]) + lane * 4;
                dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
            }
            const bool wants_gpu = (tile_masks[2 * (pos >> 6)] >> (pos & 63) & 1) != 0;
            const unsigned long long score_a = a.score[pos];
            if (score_a) {      // first window with a candidate, GPU-less nodes first for a GPU-less pod
                const int64_t winner_a = (int64_t)(NHDFIT_SCORE_INDEX(score_a) - a.global_base);
                for (int pass = (score_a >> 63) ? 0 : 1; pass < 2 && !have; ++pass) {
                    const bool pref = pass == 0;
                    const int64_t from = pref ? winner_a : ((score_a >> 63) ? 0 : winner_a);
                    for (uint32_t base = (uint32_t)(from >> 6); base < a.chunks && !have; base += 64) {
                        const uint32_t c = base + lane;
                        uint64_t w = 0;
                        if (c < a.chunks) {
                            // rows / taken: patched with atomics by the other wavefronts, read past the CU's vector cache
                            w = __hip_atomic_load(&NHDFIT_ROW(a, c, pos), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (wants_gpu) w &= ~__hip_atomic_load(&a.taken[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (pref) w &= a.nogpu[c];
                        }
                        if (c == (uint32_t)(from >> 6)) w &= ~0ull << (from & 63);
                        if (__ballot(w != 0)) {
                            s_win[wave][lane] = w;
                            if (lane == 0) s_base[wave] = base;
                            have = 1;
                        }
                    }
                }
            }
            if (wants_gpu) have += 2;
            if (lane == 0) s_pos[wave] = pos;
        }
        if (lane == 0) { s_have[wave] = have; s_status[wave] = 0; }
        __syncthreads();
        lap(t_find);
        // (2) the round's pods in order
        if (wave == 0) {
            const int32_t my_hv = lane < (uint32_t)kSeqPods ? s_have[lane] : -2;       // lane e: the round's pod e
            const uint32_t my_base = lane < (uint32_t)kSeqPods ? s_base[lane] : 0u;
            uint32_t chosen = 0xFFFFFFFFu;
            uint32_t keep = 0;
            for (; keep < (uint32_t)kSeqPods; ++keep) {
                const int32_t hv = __builtin_amdgcn_readlane(my_hv, (int)keep);
                if (hv == -2) break;
                int64_t nd = -1;
                if (hv & 1) {
                    const uint64_t w = s_win[keep][lane];             // earlier pods' nodes are already knocked out (below)
                    const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)my_base, (int)keep);
                    const uint64_t any = __ballot(w != 0);
                    if (!any) break;                              // window ran dry (never pod 0: nothing is excluded for it)
                    const int l = __builtin_ctzll(any);
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w, l);
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w >> 32), l);
                    nd = (int64_t)(base + l) * 64 + __builtin_ctzll(((uint64_t)hi << 32) | lo);
                    if (!(hv & 2) && __ballot(lane < keep && chosen == (uint32_t)nd)) break;   // the node's state after that commit decides
                    if (lane == keep) chosen = (uint32_t)nd;
                    if (lane > keep && (my_hv & 3) == 3) {        // busy for the later pods with GPUs
                        const uint32_t idx = ((uint32_t)nd >> 6) - my_base;
                        if (idx < 64u) s_win[lane][idx] &= ~(1ull << (nd & 63));
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                if (lane == 0) s_node[keep] = nd;
            }
            if (lane == 0) s_keep = keep;
        }
        __syncthreads();
        lap(t_pick);
        const uint32_t keep = s_keep;
        // (3) map + commit: one wavefront per kept pod
        if (wave < keep) {
            const int64_t nd = s_node[wave];
            if (nd < 0) {
                if (lane == 0) { SeqResult r; r.node = -1; r.map = nhdfit_mapping{}; r.status = 0; a.out[mine] = r; }
                if (a.place && lane < sizeof(nhdfit_placement) / 4) reinterpret_cast<uint32_t*>(&a.place[mine])[lane] = 0u;
            } else {
                const uint32_t v = (uint32_t)nd;
                if (a.prof && wave == 0) sub = wall_clock64();
                unsigned long long was = 0;
                if (lane == 0) was = atomicOr(reinterpret_cast<unsigned long long*>(&a.taken[v >> 6]), 1ull << (v & 63));
                const int32_t seen = lane == 0 ? a.touched[v] : 0;       // requested together with the node record
                load_node(wave, v);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                {
                    const nhdfit_req& rq = s_req[wave].r;
                    NodeState& st = s_st[wave];
                    nhdfit_detail& dd = s_det[wave];
                    sublap(0);
                    const uint32_t pos = s_pos[wave], tile = pos >> 6;
                    const uint32_t bits = nic_assignment_bits_wave(a.tabs + (size_t)tile * a.pitch, s_L[tile_wcls[tile]], pos & 63,
                                                                   rq.map_type == NHDFIT_MAP_PCI, st.p3, lane);
                    sublap(1);
                    nhdfit_mapping mp;
                    const bool mapped = map_on_state_wave(rq, st, dd, s_caps, bits, a.mt, lane, mp);      // all lanes, same result
                    __builtin_amdgcn_wave_barrier();
                    sublap(2);
                    const int32_t first_touch = __builtin_amdgcn_readfirstlane(seen) < 0;
                    if (first_touch) {                                   // first touch of this batch: keep the original (whole wavefront copies)
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
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) {
                        SeqResult& res = s_res[wave];
                        nhdfit_placement& pl = s_place[wave];
                        res.node = (int64_t)a.global_base + nd;
                        res.map = mp;
                        if (mapped) {
                            res.status = commit_node(st, dd, rq, res.map, a.now, sigs, pl);
                        } else {
                            memset(&pl, 0, sizeof pl);
                            res.map = nhdfit_mapping{};
                            res.status = kCommitWouldRaise;              // the row said feasible, the mapping disagrees: cannot happen
                            pl.status = kCommitWouldRaise;
                        }
                        s_status[wave] = res.status;
                        if (res.status == kCommitNewSig) s_stop = 1;
                    }
                    __builtin_amdgcn_wave_barrier();
                    sublap(3);
                    if (lane < sizeof(SeqResult) / 4) reinterpret_cast<uint32_t*>(&a.out[mine])[lane] = reinterpret_cast<const uint32_t*>(&s_res[wave])[lane];
                    if (a.place && lane < sizeof(nhdfit_placement) / 4)
                        reinterpret_cast<uint32_t*>(&a.place[mine])[lane] = reinterpret_cast<const uint32_t*>(&s_place[wave])[lane];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                store_node(wave, v);
                asm volatile("" :: "v"(was));                            // the taken bit has reached L2 before the round ends
                sublap(4);
            }
        }
        __syncthreads();
        lap(t_map);
        // (4) what is left of the committed nodes, for the pods without GPUs (not after a stop: the host patches those
        // nodes and starts over with the pods that are left)
        if (ngl && !s_stop) {
            // sixteen lanes per (node, tile): lane p evaluates assignment word p (W <= 16; the W words of a table row are
            // contiguous), the group ORs them together and clears the node's bit in the rows of the pods that lost it
            const uint32_t items = keep * ngl, p = lane & 15u, grp = tid >> 4;
            constexpr uint32_t kGroups = kSeqThreads / 16, kDepth = 4;       // kDepth items per group in flight: one memory round trip
            for (uint32_t k0 = 0; k0 < items; k0 += kGroups * kDepth) {
                uint64_t lost[kDepth];
                uint32_t vv[kDepth], tt[kDepth];
#pragma unroll
                for (uint32_t u = 0; u < kDepth; ++u) {
                    const uint32_t k = k0 + u * kGroups + grp;
                    lost[u] = 0; vv[u] = 0; tt[u] = 0;
                    if (k >= items) continue;                         // group-uniform
                    const uint32_t slot = k / ngl, t = gl_tiles[k % ngl];
                    if (s_node[slot] < 0) continue;
                    const NodeState& st = s_st[slot];
                    const NodeIdx ni = node_index(st.p0, st.p1, st.p2, st.p4, a.fc_dim, a.fg_dim, a.ngs);
                    const bool busy = (a.now - st.p4.busy_time) < kMinBusySecs;
                    const uint64_t need = tile_masks[2 * t];
                    const uint8_t* img = a.tabs + (size_t)t * a.pitch;
                    const Layout& L = s_L[tile_wcls[t]];
                    uint64_t term = p < L.W ? node_term_cold(img, L, ni, st.p3, tile_masks[2 * t + 1], p) : 0ull;
                    for (int m = 1; m < 16; m <<= 1) term |= __shfl_xor(term, m, 16);
                    lost[u] = ~(term & node_pred_cold(img, L, ni, busy, need)) & ~need;     // pods with GPUs go by the taken bits
                    vv[u] = (uint32_t)s_node[slot]; tt[u] = t;
                }
                if (a.prof && tid == 0) { const unsigned long long tq = wall_clock64(); t_c[0] += tq - tick; tick = tq; }
#pragma unroll
                for (uint32_t u = 0; u < kDepth; ++u)
                    for (uint32_t q = 0; q < 4; ++q) {
                        const uint32_t j = q * 16 + p;
                        // already clear for most: no harm; two nodes of one 64-node chunk may hit the same word: atomic
                        if ((lost[u] >> j & 1) && (size_t)tt[u] * 64 + j < a.P)
                            atomicAnd(reinterpret_cast<unsigned long long*>(&NHDFIT_ROW(a, vv[u] >> 6, (size_t)tt[u] * 64 + j)), ~(1ull << (vv[u] & 63)));
                    }
            }
            if (a.prof && tid == 0) { const unsigned long long tq = wall_clock64(); t_c[1] += tq - tick; tick = tq; }
            __threadfence();                                      // the patches are in L2 before the next scan
            if (a.prof && tid == 0) { const unsigned long long tq = wall_clock64(); t_c[2] += tq - tick; tick = tq; }
            __syncthreads();
            if (a.prof && tid == 0) { const unsigned long long tq = wall_clock64(); t_c[3] += tq - tick; tick = tq; t_c[4] += items; }
        }
        if (a.prof) { const unsigned long long t = wall_clock64(); tick = t; }
        ++n_rounds;
        i += keep;
    }
    if (tid == 0) *a.n_done = i;
    if (tid == 0 && a.prof) { a.prof[0] = t_find; a.prof[1] = t_map; a.prof[2] = t_pick; a.prof[3] = n_rounds; a.prof[4] = i;
                              for (int k = 0; k < 5; ++k) a.prof[5 + k] = t_sub[k]; a.prof[10] = t_c[0] + t_c[1] + t_c[2] + t_c[3]; for (int k = 0; k < 5; ++k) a.prof[11 + k] = t_c[k]; }
}

// apply = 0: put the touched nodes back
__global__ __launch_bounds__(64) void k_undo(SeqArgs a) {
    const uint32_t k = blockIdx.x;
    if (k >= a.counters[0]) return;
    const UndoRec& u = a.undo[k];
    const uint32_t lane = threadIdx.x, v = u.node;
    const uint32_t* st = reinterpret_cast<const uint32_t*>(&u.st);
    if (lane < 5) {
        const uint4 q = make_uint4(st[lane * 4], st[lane * 4 + 1], st[lane * 4 + 2], st[lane * 4 + 3]);
        if (lane == 0) *reinterpret_cast<uint4*>(a.p0 + v) = q;
        else if (lane == 1) *reinterpret_cast<uint4*>(a.p1 + v) = q;
        else if (lane == 2) *reinterpret_cast<uint4*>(a.p2 + v) = q;
        else if (lane == 3) *reinterpret_cast<uint4*>(a.p3 + v) = q;
        else *reinterpret_cast<uint4*>(a.p4 + v) = q;
    }
    if (lane >= 8 && lane < 16) reinterpret_cast<uint4*>(a.det + v)[lane - 8] = reinterpret_cast<const uint4*>(&u.d)[lane - 8];
}

// K3 (nhdfit_apply_deltas): one lane per run of deltas that name the same node (the host sorts the array by node,
// keeping the order inside a node): load the node, apply the run in order, store it.  Runs are independent.
struct DeltaArgs {
    nhdfit_plane0* p0; nhdfit_plane1* p1; nhdfit_plane2* p2; nhdfit_plane3* p3; nhdfit_plane4* p4; nhdfit_detail* det;
    nhdfit_origin* origin;
    const nhdfit_delta* deltas; const uint32_t* run; uint32_t n_runs;     // run[r] .. run[r+1]: deltas of one node
    SigTable sigs; uint8_t* status;
};
__global__ __launch_bounds__(64) void k_delta(DeltaArgs a) {
    const uint32_t r = blockIdx.x * 64 + threadIdx.x;
    if (r >= a.n_runs) return;
    const uint32_t lo = a.run[r], hi = a.run[r + 1], v = a.deltas[lo].node;
    NodeState s;
    s.p0 = a.p0[v]; s.p1 = a.p1[v]; s.p2 = a.p2[v]; s.p3 = a.p3[v]; s.p4 = a.p4[v];
    nhdfit_detail d = a.det[v];
    nhdfit_origin o = a.origin[v];
    for (uint32_t k = lo; k < hi; ++k) a.status[k] = (uint8_t)apply_delta(s, d, o, a.deltas[k], a.sigs);
    a.p0[v] = s.p0; a.p1[v] = s.p1; a.p2[v] = s.p2; a.p3[v] = s.p3; a.p4[v] = s.p4;
    a.det[v] = d;
    a.origin[v] = o;
}

// the commit step for one placement (nhdfit_commit)
// The placement goes straight into a fine-grained host block, the call's sequence number behind it (system scope): the host polls that
// word - no copy command behind the launch, no stream wait (round 6: the commit of the scheduler's pod-at-a-time loop, nhd/NHDScheduler.py:
// 289-304, was a launch + a D2H copy into pageable memory + a wait).
struct CommitHost { uint32_t flag; uint32_t pad[3]; nhdfit_placement place; };
struct CommitArgs {
    nhdfit_plane0* p0; nhdfit_plane1* p1; nhdfit_plane2* p2; nhdfit_plane3* p3; nhdfit_plane4* p4; nhdfit_detail* det;
    uint32_t node; nhdfit_req req; nhdfit_mapping map; double busy_time; SigTable sigs; CommitHost* host; uint32_t seq;
    uint32_t ncls;                 // capacity classes of the dictionary (the wavefront form's signature keys)
};
__device__ __forceinline__ void commit_publish(CommitHost* h, const nhdfit_placement& pl, uint32_t seq) {
    h->place = pl;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&h->flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// (k_commit itself: seq2_kernel.h, behind the wavefront form of the commit step)
