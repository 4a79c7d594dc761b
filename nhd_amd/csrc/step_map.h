// step_map.h - mapping roles of the step kernel (shapes / choose / finish) and the generic G = 4 mapping kernel.
// Device code of libnhdfit.so; included by nhdfit.hip inside its anonymous namespace, in this order: step_digest.h,
// step_fit.h, step_map.h, step_kernel.h, seq_kernel.h (one translation unit: the roles are fused into one kernel).
// gfx950 only.
// ---- winner mapping for G <= 3 pods, de-duplicated by candidate-set shape --------------------------
// The sequential CPython-set model (choose_tuples) is a pure function of 35 bits (shape_key).  So: (1) every pod
// derives its shape in parallel, the distinct shapes of each 64-pod tile are collected (wave ballots, no atomics),
// (2) one wavefront per distinct shape runs the set model, (3) every pod finishes its mapping (first valid NIC
// choice) in parallel.  Nothing survives the step.
__device__ __forceinline__ bool load_winner(const MapArgs& a, uint32_t p, WinnerState& w, uint32_t& i) {
    const unsigned long long s = a.score[p];
    if (!s) return false;
    const uint64_t gi = NHDFIT_SCORE_INDEX(s);
    if (gi < a.global_base || gi >= a.global_base + a.n) return false;
    i = (uint32_t)(gi - a.global_base);
    const nhdfit_plane0 q0 = a.p0[i];
    const nhdfit_plane1 q1 = a.p1[i];
    const nhdfit_plane2 q2 = a.p2[i];
    w.d = a.det + i;
    w.U = w.d->numa_nodes;
    w.smt = (q2.flags & NHDFIT_NF_SMT) != 0;
    w.free_c[0] = popc64(q0.t0[0] & q1.t1[0]);
    w.free_c[1] = popc64(q0.t0[1] & q1.t1[1]);
    w.free_g[0] = popc32(q2.gpu_free & ~q2.gpu_numa1);
    w.free_g[1] = popc32(q2.gpu_free & q2.gpu_numa1);
    w.caps = a.caps;
    return true;
}

struct ShapeArgs {
    unsigned long long* keys;    // [tiles*64] distinct shapes of tile t at [64 t, 64 t + count[t])
    uint32_t* result;            // [tiles*64] ok << 8 | gcode << 4 | ccode of the shape in the same slot
    int32_t* slot_of_pod;        // [P] slot of the pod's shape, < 0: nothing to map
    uint32_t* count;             // [tiles]
    const AscEntry* asc;         // layouts of ascending-filled sets (winner_map.h), built once per context
    const uint8_t* choose_tab;   // tabulated choose_tuples for U = 2, G <= 2 (winner_map.h), or null
    SetStates st;                // set-layout state machine for U = 2, G = 3 (set_states.h); info == null: not used
};
// slot_of_pod encodings: >= 0 slot of the pod's shape; -1 nothing to map; <= -2: the result word itself, -2 - word
// (shapes answered from choose_tab never reach the choose role)

__global__ __launch_bounds__(256) void k_build_choose(const AscEntry* asc, uint8_t* table) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e < kChooseEntries) table[e] = choose_entry_build(asc, e);
}

// one thread per (tuple length, subset): the set model itself fills the table
__global__ __launch_bounds__(256) void k_build_asc(AscEntry* table) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= kAscEntries) return;
    const int len = e >= kAscOffset[4] ? 4 : e >= kAscOffset[3] ? 3 : e >= kAscOffset[2] ? 2 : 1;
    table[e] = asc_entry_build(len, e - kAscOffset[len]);
}


__device__ __forceinline__ unsigned long long shfl64(unsigned long long v, int lane) {
    return ((unsigned long long)(uint32_t)__shfl((int)(v >> 32), lane, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)v, lane, 64);
}

// Staging for the lane = pod mapping roles.  The mapping arithmetic (candidate masks, first NIC choice) indexes the
// request record and the winner's detail record dynamically inside nested loops; against global memory every such
// access is a dependent L2 round trip and a role becomes a 25-30 us latency chain.  So a block first copies the
// records of its pods (THREADS / 4 of them: the copies are cooperative, the arithmetic runs on a quarter of the
// threads) into LDS with wide coalesced loads - three round trips in all (score, node records, table rows).
struct PaddedDet { nhdfit_detail d; uint32_t pad; };           // 33-word stride: lane j -> bank j
struct StagedNode {                                            // 9 words
    int32_t node;                                              // local index of the pod's winner, -1: none on this shard
    uint32_t free_c[2], free_g[2];
    uint16_t sig_numa[2], sig_pci[2];
    uint32_t smt, U;
};
struct MapStage {
    PaddedReq* req; PaddedDet* det; StagedNode* w; nhdfit_mapping* map; double* caps;
};
template <int THREADS>
constexpr size_t map_lds_bytes() {
    constexpr size_t pods = THREADS / 4;
    return lds_slice(pods * sizeof(PaddedReq)) + lds_slice(pods * sizeof(PaddedDet)) + lds_slice(pods * sizeof(StagedNode)) +
           lds_slice(pods * sizeof(nhdfit_mapping)) + lds_slice(NHDFIT_MAX_CLASSES * sizeof(double));
}
// map_one_tile also keeps the set-state tables behind the staging area (up to kStLdsStates states; the enumeration yields 338)
constexpr uint32_t kStLdsStates = 384;
constexpr size_t kStLdsBytes = lds_slice(kStLdsStates * sizeof(uint64_t)) + lds_slice((size_t)kStLdsStates * 8 * sizeof(uint32_t)) + lds_slice(256 * sizeof(uint32_t));
template <int THREADS>
constexpr size_t map_tile_lds_bytes() { return map_lds_bytes<THREADS>() + kStLdsBytes; }
template <int THREADS>
__device__ __forceinline__ MapStage stage_winners(const MapArgs& a, uint32_t pod0, uint8_t* lds) {
    constexpr uint32_t PODS = THREADS / 4;
    MapStage s;
    s.req = carve<PaddedReq>(lds, PODS);
    s.det = carve<PaddedDet>(lds, PODS);
    s.w = carve<StagedNode>(lds, PODS);
    s.map = carve<nhdfit_mapping>(lds, PODS);
    s.caps = carve<double>(lds, NHDFIT_MAX_CLASSES);
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t kParts = sizeof(nhdfit_req) / 16;
    const uint32_t live = pod0 < a.P ? (a.P - pod0 < PODS ? a.P - pod0 : PODS) : 0u;
    {   // request records, coalesced
        const uint4* src = reinterpret_cast<const uint4*>(a.reqs + pod0);
        for (uint32_t c = tid; c < PODS * kParts; c += THREADS) {
            const uint32_t j = c / kParts;
            const uint4 v = j < live ? src[c] : make_uint4(0u, 0u, 0u, 0u);
            uint32_t* dst = reinterpret_cast<uint32_t*>(&s.req[j]) + (c % kParts) * 4;
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
    }
    if (tid < NHDFIT_MAX_CLASSES) s.caps[tid] = a.caps[tid];                // the dictionary buffer holds >= 16 entries
    if (tid < PODS) {                                                       // winners and their plane-derived counts
        StagedNode n;
        n.node = -1;
        n.free_c[0] = n.free_c[1] = n.free_g[0] = n.free_g[1] = 0; n.smt = 0; n.U = 1;
        n.sig_numa[0] = n.sig_numa[1] = n.sig_pci[0] = n.sig_pci[1] = 0;
        const unsigned long long sc = tid < live ? a.score[pod0 + tid] : 0ull;
        if (sc) {
            const uint64_t gi = NHDFIT_SCORE_INDEX(sc);
            if (gi >= a.global_base && gi < a.global_base + a.n) {
                const uint32_t i = (uint32_t)(gi - a.global_base);
                const nhdfit_plane0 q0 = a.p0[i];
                const nhdfit_plane1 q1 = a.p1[i];
                const nhdfit_plane2 q2 = a.p2[i];
                const nhdfit_plane3 q3 = a.p3[i];
                n.node = (int32_t)i;
                n.smt = (q2.flags & NHDFIT_NF_SMT) != 0;
                n.free_c[0] = popc64(q0.t0[0] & q1.t1[0]); n.free_c[1] = popc64(q0.t0[1] & q1.t1[1]);
                n.free_g[0] = popc32(q2.gpu_free & ~q2.gpu_numa1); n.free_g[1] = popc32(q2.gpu_free & q2.gpu_numa1);
                n.sig_numa[0] = q3.sig_numa[0]; n.sig_numa[1] = q3.sig_numa[1];
                n.sig_pci[0] = q3.sig_pci[0]; n.sig_pci[1] = q3.sig_pci[1];
            }
        }
        s.w[tid] = n;
    }
    __syncthreads();
    constexpr uint32_t kDetParts = sizeof(nhdfit_detail) / 16;              // detail records of the winners: 8 lanes x 16 B per pod
    for (uint32_t c = tid; c < PODS * kDetParts; c += THREADS) {
        const uint32_t j = c / kDetParts, part = c % kDetParts;
        const int32_t i = s.w[j].node;
        if (i >= 0) {
            const uint4 v = reinterpret_cast<const uint4*>(a.det + i)[part];
            uint32_t* dst = reinterpret_cast<uint32_t*>(&s.det[j]) + part * 4;
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
    }
    __syncthreads();
    return s;
}
__device__ __forceinline__ WinnerState staged_state(const MapStage& s, uint32_t j) {
    WinnerState w;
    w.d = &s.det[j].d;
    w.U = s.det[j].d.numa_nodes;
    w.smt = s.w[j].smt != 0;
    w.free_c[0] = (int)s.w[j].free_c[0]; w.free_c[1] = (int)s.w[j].free_c[1];
    w.free_g[0] = (int)s.w[j].free_g[0]; w.free_g[1] = (int)s.w[j].free_g[1];
    w.caps = s.caps;
    return w;
}

// (1) lane = pod, wavefront = tile: derive the shape, de-duplicate within the tile
template <int THREADS>
__device__ __forceinline__ void role_shapes(const MapArgs& a, const ShapeArgs& h, uint32_t blk, uint8_t* lds) {
    constexpr uint32_t PODS = THREADS / 4;
    const uint32_t pod0 = blk * PODS;
    const MapStage st = stage_winners<THREADS>(a, pod0, lds);
    if (threadIdx.x >= PODS) return;
    const uint32_t j = threadIdx.x, p = pod0 + j, tile = p >> 6, lane = threadIdx.x & 63;
    if (tile * 64 >= a.P) return;                      // whole wavefront past the end
    int32_t slot = -1;
    unsigned long long key = 0;
    const nhdfit_req& rq = st.req[j].r;
    if (p < a.P && rq.n_groups <= 3 && st.w[j].node >= 0) {
        const WinnerState w = staged_state(st, j);
        nhdfit_plane3 q3;
        q3.groups = 0;
        q3.sig_numa[0] = st.w[j].sig_numa[0]; q3.sig_numa[1] = st.w[j].sig_numa[1];
        q3.sig_pci[0] = st.w[j].sig_pci[0]; q3.sig_pci[1] = st.w[j].sig_pci[1];
        const uint32_t bits = nic_assignment_bits(a.tabs + (size_t)tile * a.pitch, a.L[a.tile_wcls[tile]], lane,
                                                  rq.map_type == NHDFIT_MAP_PCI, q3);
        const uint32_t codes = nic_codes_from_table_bits(bits, (int)rq.n_groups, w.U);
        uint32_t sg, sc;
        candidate_masks(rq, w, sg, sc);
        if (sg && sc && codes) {
            if (h.choose_tab && choose_tabulated((int)rq.n_groups, w.U))
                slot = -2 - (int32_t)choose_from_table(h.choose_tab, (int)rq.n_groups, sg, sc, codes);
            else
                key = shape_key((int)rq.n_groups, w.U, sg, sc, codes);
        }
    }
    // distinct shapes of the tile (pods of a tile mostly share a handful): slot 64 tile + j for the j-th one.
    // No cross-tile interning: it needs a hash table in global memory, and its atomics cost the concurrently
    // running fit role more than the extra runs of the set model cost the choose role.
    unsigned long long todo = __ballot(key != 0ull);
    uint32_t nd = 0;
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const unsigned long long k = shfl64(key, leader);
        if ((int)lane == leader) h.keys[tile * 64 + nd] = k;
        if (key == k) slot = (int32_t)(tile * 64 + nd);
        todo &= ~__ballot(key == k);
        ++nd;
    }
    if (lane == 0) h.count[tile] = nd;
    if (p < a.P) h.slot_of_pod[p] = slot;
}

// (2) one wavefront (its lane 0: the model is strictly sequential) per distinct shape.  The model lives in
// scalar registers (it is wave-uniform); not inlined into k_step so that its SGPR spill slots do not become
// VGPRs of every role - the fit role's occupancy is set by the kernel's VGPR count.
__device__ __forceinline__ void role_choose(const ShapeArgs& h, uint32_t w, uint32_t waves, uint32_t tiles) {
    // wavefront w of the role takes the shapes j = sub, sub + S, ... of tile (w mod tiles): a few shapes per wave
    // keeps the role on few CUs (its scalar code competes with the fit role for the scalar unit and the I-cache)
    const uint32_t S = waves / tiles ? waves / tiles : 1u;
    if (w >= S * tiles) return;
    const uint32_t tile = w % tiles, sub = w / tiles;
    const uint32_t count = (uint32_t)__builtin_amdgcn_readfirstlane((int)h.count[tile]);
    for (uint32_t j = sub; j < count; j += S) {
        const uint32_t k = tile * 64 + j;
        const unsigned long long kv = h.keys[k];
        const unsigned long long key = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(kv >> 32)) << 32) |
                                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)kv);
        const int G = (int)(key & 3), U = (int)((key >> 2) & 1) + 1;
        uint32_t gcode = 0;
        int ccode = -1;
        if (h.st.info && G == 3 && U == 2) {         // ~40 table look-ups instead of the insertion-by-insertion model
            h.result[k] = choose_g3(h.st, h.asc, (uint32_t)(key >> 3) & 0xFF, (uint32_t)(key >> 19) & 0xFFFF, (uint32_t)(key >> 11) & 0xFF);
            continue;
        }
        const bool ok = choose_tuples<SmallOps>(G, U, (uint32_t)(key >> 3) & 0xFF, (uint32_t)(key >> 19) & 0xFFFF,
                                                (uint32_t)(key >> 11) & 0xFF, gcode, ccode, h.asc);
        h.result[k] = ((uint32_t)ok << 8) | ((gcode & 7u) << 4) | ((uint32_t)ccode & 15u);
    }
}

// (2') the same role lane-parallel: wavefront = tile, lane = one of its distinct shapes.  The G = 3 state machine is ~40
// dependent table look-ups per shape whichever lane runs it, so a tile's shapes cost one chain instead of a wavefront each -
// 8 blocks for 64 tiles instead of 128, which leaves their block slots to the roles that wait for one.  Shapes outside the
// state machine (no tables, U = 1) run the sequential model one after the other on lane 0, as before.
__device__ __forceinline__ void role_choose_lanes(const ShapeArgs& h, uint32_t tile, uint32_t tiles) {
    if (tile >= tiles) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t count = (uint32_t)__builtin_amdgcn_readfirstlane((int)h.count[tile]);
    const uint32_t k = tile * 64 + lane;
    unsigned long long key = 0;
    bool generic = false;
    if (lane < count) {
        key = h.keys[k];
        const int G = (int)(key & 3), U = (int)((key >> 2) & 1) + 1;
        if (h.st.info && G == 3 && U == 2)
            h.result[k] = choose_g3(h.st, h.asc, (uint32_t)(key >> 3) & 0xFF, (uint32_t)(key >> 19) & 0xFFFF, (uint32_t)(key >> 11) & 0xFF);
        else
            generic = true;
    }
    unsigned long long todo = __ballot(generic);
    while (todo) {
        const int j = __builtin_ctzll(todo);
        todo &= todo - 1;
        const unsigned long long kv = shfl64(key, j);
        if (lane == 0) {
            const unsigned long long kk = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(kv >> 32)) << 32) |
                                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)kv);
            const int G = (int)(kk & 3), U = (int)((kk >> 2) & 1) + 1;
            uint32_t gcode = 0;
            int ccode = -1;
            const bool ok = choose_tuples<SmallOps>(G, U, (uint32_t)(kk >> 3) & 0xFF, (uint32_t)(kk >> 19) & 0xFFFF,
                                                    (uint32_t)(kk >> 11) & 0xFF, gcode, ccode, h.asc);
            h.result[tile * 64 + j] = ((uint32_t)ok << 8) | ((gcode & 7u) << 4) | ((uint32_t)ccode & 15u);
        }
    }
}

// (3) lane = pod: first valid NIC choice under the chosen tuples, from the staged copies; the mappings leave the
// block as one coalesced store.
template <int THREADS>
__device__ __forceinline__ void role_finish(const MapArgs& a, const ShapeArgs& h, uint32_t blk, uint8_t* lds) {
    constexpr uint32_t PODS = THREADS / 4;
    const uint32_t pod0 = blk * PODS;
    const MapStage st = stage_winners<THREADS>(a, pod0, lds);
    const uint32_t j = threadIdx.x, p = pod0 + j;
    if (j < PODS) {
        nhdfit_mapping& m = st.map[j];
        memset(&m, 0, sizeof(m));
        const nhdfit_req& rq = st.req[j].r;
        if (p < a.P && rq.n_groups <= 3) {
            const int32_t slot = h.slot_of_pod[p];
            if (slot != -1) {
                const uint32_t res = slot >= 0 ? h.result[slot] : (uint32_t)(-2 - slot);
                if ((res >> 8 & 1) && st.w[j].node >= 0) finish_mapping(rq, staged_state(st, j), (res >> 4) & 7u, (int)(res & 15u), m);
            }
        }
    }
    __syncthreads();
    // 20-byte records, PODS of them: copied out as words; pods with more than 3 groups belong to k_map<true>
    const uint32_t live = pod0 < a.P ? (a.P - pod0 < PODS ? a.P - pod0 : PODS) : 0u;
    constexpr uint32_t kWords = sizeof(nhdfit_mapping) / 4;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(st.map);
    uint32_t* dst = reinterpret_cast<uint32_t*>(a.out + pod0);
    for (uint32_t c = threadIdx.x; c < live * kWords; c += THREADS)
        if (st.req[c / kWords].r.n_groups <= 3) dst[c] = src[c];
}

// The three mapping roles for ONE pod tile in one block with ONE staging - the tail of the single-launch find (k_find).  A tile's
// pods are one wavefront, its distinct shapes at most 64: lane nd keeps shape nd's key in a register, runs the set model for
// it (the state machine; anything else one after the other on lane 0, as role_choose_lanes does) and hands the result to the
// pods of that shape with a lane shuffle - nothing goes through memory between the phases, the winners are staged once.
// `wcls`: the tile's row width class (the step roles read it from tile_wcls).
// LONE: the tile is one pod and there is no table image - the NIC-feasible assignments come from the pod's own masks (`lone`).
// `tile`: which tile of the staged batch (k_find: its only one; k_map_tiles: any).
template <int THREADS, bool LONE = false>
__device__ __forceinline__ void map_one_tile(const MapArgs& a, const ShapeArgs& h, uint32_t wcls, uint8_t* lds, const LoneMasks* lone = nullptr,
                                             uint32_t tile = 0, unsigned long long* clk = nullptr) {
    // tuning aid (k_map_tiles, NHDFIT_DRAIN_PROF): the latest time after the block's start at which any block passed phase k -> clk[k]
    const unsigned long long t_blk = kTuning && clk ? (unsigned long long)wall_clock64() : 0ull;
    auto lap = [&](int k) { if (kTuning && clk && threadIdx.x == 0) atomicMax(&clk[k], (unsigned long long)wall_clock64() - t_blk); };
    static_assert(THREADS / 4 == kTile, "the pods of the tile are one wavefront");
    const uint32_t pod0 = tile * kTile;
    // The set-layout state machine of the three-group shapes is ~40 DEPENDENT look-ups per shape: against global memory that is
    // the longest chain of the block (the tiles with three-group pods set the length of a drain launch, ~30 us).  Its tables are
    // 14.5 KB - every thread copies its share into LDS behind the staging area (the copy rides along with the winners' staging)
    // and the look-ups stay in the CU.
    SetStates lst = h.st;
    if (h.st.info && h.st.n <= kStLdsStates) {
        uint8_t* q = lds + map_lds_bytes<THREADS>();
        uint64_t* l_info = carve<uint64_t>(q, kStLdsStates);
        uint32_t* l_next = carve<uint32_t>(q, (size_t)kStLdsStates * 8);
        uint32_t* l_asc = carve<uint32_t>(q, 256);
        for (uint32_t i = threadIdx.x; i < h.st.n; i += THREADS) l_info[i] = h.st.info[i];
        for (uint32_t i = threadIdx.x; i < h.st.n * 8; i += THREADS) l_next[i] = h.st.next[i];
        for (uint32_t i = threadIdx.x; i < 256; i += THREADS) l_asc[i] = h.st.asc[i];
        lst = SetStates{l_info, l_next, l_asc, h.st.n};                   // (stage_winners' barriers below order the copies)
    }
    const MapStage st = stage_winners<THREADS>(a, pod0, lds);
    lap(0);
    const uint32_t j = threadIdx.x;
    if (j < (uint32_t)kTile) {
        const uint32_t lane = j;
        nhdfit_mapping& m = st.map[j];
        memset(&m, 0, sizeof(m));
        const nhdfit_req& rq = st.req[j].r;
        const bool live = pod0 + j < a.P && rq.n_groups <= 3 && st.w[j].node >= 0;
        int32_t slot = -1;
        unsigned long long key = 0;
        if (live) {                                                       // (1) the pod's shape
            const WinnerState w = staged_state(st, j);
            nhdfit_plane3 q3;
            q3.groups = 0;
            q3.sig_numa[0] = st.w[j].sig_numa[0]; q3.sig_numa[1] = st.w[j].sig_numa[1];
            q3.sig_pci[0] = st.w[j].sig_pci[0]; q3.sig_pci[1] = st.w[j].sig_pci[1];
            uint32_t bits;
            if constexpr (LONE) bits = lone_nic_bits(*lone, rq.map_type == NHDFIT_MAP_PCI, q3);
            else bits = nic_assignment_bits(a.tabs + (size_t)tile * a.pitch, a.L[wcls], lane, rq.map_type == NHDFIT_MAP_PCI, q3);
            const uint32_t codes = nic_codes_from_table_bits(bits, (int)rq.n_groups, w.U);
            uint32_t sg, sc;
            candidate_masks(rq, w, sg, sc);
            if (sg && sc && codes) {
                if (h.choose_tab && choose_tabulated((int)rq.n_groups, w.U))
                    slot = -2 - (int32_t)choose_from_table(h.choose_tab, (int)rq.n_groups, sg, sc, codes);
                else
                    key = shape_key((int)rq.n_groups, w.U, sg, sc, codes);
            }
        }
        lap(1);
        unsigned long long todo = __ballot(key != 0ull), mine = 0;
        uint32_t nd = 0;
        while (todo) {                                                    // distinct shapes: lane nd keeps the nd-th
            const int leader = __builtin_ctzll(todo);
            const unsigned long long k = shfl64(key, leader);
            if (lane == nd) mine = k;
            if (key == k) slot = (int32_t)nd;
            todo &= ~__ballot(key == k);
            ++nd;
        }
        lap(2);
        uint32_t res = 0;                                                 // (2) the set model per distinct shape
        bool generic = false;
        if (lane < nd) {
            const int G = (int)(mine & 3), U = (int)((mine >> 2) & 1) + 1;
            if (lst.info && G == 3 && U == 2)
                res = choose_g3(lst, h.asc, (uint32_t)(mine >> 3) & 0xFF, (uint32_t)(mine >> 19) & 0xFFFF, (uint32_t)(mine >> 11) & 0xFF);
            else
                generic = true;
        }
        lap(3);
        unsigned long long gtodo = __ballot(generic);
        while (gtodo) {
            const int jj = __builtin_ctzll(gtodo);
            gtodo &= gtodo - 1;
            const unsigned long long kv = shfl64(mine, jj);
            uint32_t r = 0;
            if (lane == 0) {
                const unsigned long long kk = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(kv >> 32)) << 32) |
                                              (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)kv);
                const int G = (int)(kk & 3), U = (int)((kk >> 2) & 1) + 1;
                uint32_t gcode = 0;
                int ccode = -1;
                const bool ok = choose_tuples<SmallOps>(G, U, (uint32_t)(kk >> 3) & 0xFF, (uint32_t)(kk >> 19) & 0xFFFF,
                                                        (uint32_t)(kk >> 11) & 0xFF, gcode, ccode, h.asc);
                r = ((uint32_t)ok << 8) | ((gcode & 7u) << 4) | ((uint32_t)ccode & 15u);
            }
            r = (uint32_t)__shfl((int)r, 0, 64);
            if ((int)lane == jj) res = r;
        }
        lap(4);
        const uint32_t got = (uint32_t)__shfl((int)res, slot >= 0 ? slot : 0, 64);      // (3) first valid NIC choice under the chosen tuples
        if (live && slot != -1) {
            const uint32_t rr = slot >= 0 ? got : (uint32_t)(-2 - slot);
            if (rr >> 8 & 1) finish_mapping(rq, staged_state(st, j), (rr >> 4) & 7u, (int)(rr & 15u), m);
        }
        lap(5);
    }
    __syncthreads();
    const uint32_t livep = pod0 < a.P ? (a.P - pod0 < (uint32_t)kTile ? a.P - pod0 : (uint32_t)kTile) : 0u;
    constexpr uint32_t kWords = sizeof(nhdfit_mapping) / 4;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(st.map);
    uint32_t* dst = reinterpret_cast<uint32_t*>(a.out + pod0);
    for (uint32_t c = threadIdx.x; c < livep * kWords; c += THREADS)
        if (st.req[c / kWords].r.n_groups <= 3) dst[c] = src[c];
}

// Draining the pipeline (sync / fetch behind the last step): the steps whose mapping phases have not all run are mapped from
// their scores in ONE launch, one block per (step, tile) with the three phases back to back in the block (map_one_tile) -
// instead of up to three more role launches in a row, each a latency chain of its own behind a launch gap (~70 us -> ~30
// for a drain; it is what a short run of steps pays at its end).  The phases a step had already been through are simply
// redone: they are pure functions of the step's scores, the mirror and the staged requests.
constexpr int kDrainSteps = 3;
struct DrainArgs {
    uint32_t nsteps, tiles;
    MapArgs m[kDrainSteps];
    ShapeArgs h;                      // its static tables only (asc, choose_tab, st); the per-step shape buffers are not used
    unsigned long long* clk;          // tuning aid (NHDFIT_DRAIN_PROF): [0..5] latest time after a block's start at which a phase was passed, [6] the launch's span
};
__global__ __launch_bounds__(256) void k_map_tiles(DrainArgs a) {
    extern __shared__ __align__(16) uint8_t lds_drain[];
    const uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x / a.tiles)), tile = blockIdx.x - s * a.tiles;
    const MapArgs& m = a.m[s < (uint32_t)kDrainSteps ? s : 0];
    const unsigned long long t0 = kTuning && a.clk ? (unsigned long long)wall_clock64() : 0ull;
    map_one_tile<256>(m, a.h, m.tile_wcls[tile], lds_drain, nullptr, tile, kTuning ? a.clk : nullptr);
    if (kTuning && a.clk && threadIdx.x == 0) { atomicMin(&a.clk[7], t0); atomicMax(&a.clk[6], (unsigned long long)wall_clock64()); }
}
