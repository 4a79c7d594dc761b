// big_kernel.h - the general path for REQUESTS: pods with 5..8 processing groups (nhdfit_big_req, include/nhdfit.h) against
// every node of the mirror.  Device code of libnhdfit.so; included by nhdfit.hip inside its anonymous namespace.  gfx950 only.
//
// The reference enumerates itertools.product(range(numa_nodes), repeat=len(req)) for any group count (nhd/Matcher.py:118, 203,
// 242); the table-driven pass holds masks over 2^G <= 16 assignments.  A big request is therefore answered the way a wide
// node is (wide_core.h): explicit enumeration with the reference's own arithmetic, here lane = node, grid.y = pod.  An ordinary
// node is read through wide_view (its five planes and detail record in the wide record's terms), a wide node from its record;
// the wavefront max-reduces its 64 score words (DPP-free: __shfl_xor over the 64 lanes) and posts one atomicMax per pod.
// k_big_map gives each winner its mapping from the general CPython set model, k_big_commit applies the commit step on
// whichever form the winner is mirrored in (commit_node_t on the planes, wide_commit on a wide record).
// Nothing here is fast - a cluster's rare many-group pod costs milliseconds, not the table pass's microseconds - everything
// here is exact.
// ---- the winner's mapping with the wavefront's lanes ---------------------------------------------------------------------------------
// wide_map (wide_core.h) for a big request on a node of at most two NUMA nodes - 2^G <= 256 G-tuples, 2^(G+1) (G+1)-tuples.  One thread
// spent its time (profiles/r04: 1.9 ms per mapping) on the stages - per tuple a GPU / CPU sum and two NIC searches on a cut-down copy of
// the request - and on tuple hashes, recomputed digit by digit at every probe of the set model.  Both are independent per tuple: lane =
// tuple answers the three stages into bit rows and computes every tuple's hash once, into LDS; the set model itself (CPython sets filled
// in product order, intersected, iterated in slot order: a chain) then runs on the first lane over those tables (wide_map_model with
// WideStagesLds).  Same searches, same steps in total (nic_stage_ok_plain), same model: the answer is wide_map's, bit for bit
// (tests/test_wave_commit_emulation.py runs this text under the 64-thread emulation against wide_map).
constexpr uint32_t kBigWaveG = 256, kBigWaveC = 512;             // tuples the LDS tables hold: two NUMA nodes, eight groups
struct BigWaveLds {
    nhdfit_wide_node view;                                        // the winner, as the general path reads it
    nhdfit_big_req req;
    uint64_t hash_g[kBigWaveG], hash_c[kBigWaveC];                // tuplehash of every G- / (G+1)-tuple
    uint64_t bit_g[kBigWaveG / 64], bit_n[kBigWaveG / 64], bit_c[kBigWaveC / 64];   // the stages' answers, bit = tuple
    uint32_t steps[64];                                           // NIC search steps per lane
    int32_t rc;
};
struct WideStagesLds {
    const BigWaveLds& t;
    NHD_HD bool gpu(uint32_t code) const { return t.bit_g[code >> 6] >> (code & 63) & 1ull; }
    NHD_HD bool nic(uint32_t code) const { return t.bit_n[code >> 6] >> (code & 63) & 1ull; }
    NHD_HD bool cpu(uint32_t code) const { return t.bit_c[code >> 6] >> (code & 63) & 1ull; }
    NHD_HD bool exhausted() const { return false; }               // (decided before the model starts)
    NHD_HD const uint64_t* hash_g() const { return t.hash_g; }
    NHD_HD const uint64_t* hash_c() const { return t.hash_c; }
};
// `t.view` / `t.req` are in place (and visible to every lane); `tables`: big_scratch_words(2, G) int32 of LDS.  Every lane returns wide_map's
// code; lane 0 holds the mapping in `out`.
__device__ __noinline__ int wide_map_wave(BigWaveLds& t, const WideCaps& caps, int32_t* tables, nhdfit_big_mapping& out, int32_t slots_g, int32_t slots_c, uint32_t lane) {
    const nhdfit_wide_node& n = t.view;
    const nhdfit_big_req& r = t.req;
    wide_map_clear(out, NHDFIT_BIG_MAX_GROUPS);
    if (!req_valid(r) || !wide_shape_ok(n)) return 0;
    const WideFree f = wide_free(n);
    const uint32_t G = r.n_groups, U = f.U, nG = wide_ipow(U, G), nC = nG * U;
    const bool separable = nic_separable(n, r);
    NicSearch ns{8u * NHDFIT_BIG_NIC_BUDGET, false};              // (a lane that alone outruns the call's budget has outrun it)
    for (uint32_t base = 0; base < nG; base += 64) {
        const uint32_t code = base + lane;
        bool g = false, k = false;
        if (code < nG) {
            g = wide_gpu_ok(r, f, code);
            k = nic_stage_ok_plain(separable, n, r, caps, code, &ns);
            t.hash_g[code] = wide_tuple_hash(code, G, U);
        }
        const uint64_t mg = __ballot(g), mk = __ballot(k);
        if (lane == 0) { t.bit_g[base >> 6] = mg; t.bit_n[base >> 6] = mk; }
    }
    for (uint32_t base = 0; base < nC; base += 64) {
        const uint32_t code = base + lane;
        bool ok = false;
        if (code < nC) {
            ok = wide_cpu_ok(r, f, code);
            t.hash_c[code] = wide_tuple_hash(code, G + 1, U);
        }
        const uint64_t mc = __ballot(ok);
        if (lane == 0) t.bit_c[base >> 6] = mc;
    }
    t.steps[lane] = 8u * NHDFIT_BIG_NIC_BUDGET - ns.left;
    const bool any_out = __ballot(ns.exhausted) != 0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane == 0) {
        uint64_t total = 0;
        for (int l = 0; l < 64; ++l) total += t.steps[l];
        int rc = -2;                                              // the budget of the call's searches, summed over the tuples as one thread spends it
        if (!any_out && total <= (uint64_t)(8u * NHDFIT_BIG_NIC_BUDGET)) {
            WideStagesLds st{t};
            rc = wide_map_model(n, r, caps, f, tables, out, slots_g, slots_c, st);
        }
        t.rc = rc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return t.rc;
}

// ---- the kernels of the general path for requests ----------------------------------------------------------------------------------------
struct BigEvalArgs {
    const nhdfit_plane0* p0; const nhdfit_plane1* p1; const nhdfit_plane2* p2; const nhdfit_plane3* p3; const nhdfit_plane4* p4;
    const nhdfit_detail* det; uint32_t n;
    const nhdfit_wide_node* wide; uint32_t n_wide;
    const nhdfit_big_req* reqs; uint32_t P;
    const double* caps; double busy_from;
    const uint64_t* cand;                          // optional [chunks] candidate nodes
    unsigned long long* score; uint64_t global_base;
    uint32_t* flags;                               // [0] a set of the model outgrew its table, [1] a (pod, node) pair ran out of NIC search budget,
                                                   // [2] the most search steps any (pod, node) pair of the call took
    const nhdfit_wide_share* share;                // optional [n_wide]: ENABLE_SHARING arithmetic
};

// grid.x = 64-node chunks (the planes' nodes, then the wide records), grid.y = pod; lane = node: the scalar tests and the totals turn most nodes
// of a cluster away for a pod of this size, the others walk their assignments to the first one that passes (wide_fits).  (Round 5 also
// measured the walk with the wavefront's lanes - the 64 lanes take the nodes that pass one after the other, lane = assignment - against
// this form: 2.8 ms against 0.44 per call on config 4's 65 536 nodes.  The nodes of a chunk are mostly of one kind, so their walks run in
// step and lane = node wastes little; one node at a time pays the walk's set-up 64 times over.  profiles/r05/README.md.)
__global__ __launch_bounds__(64) void k_big_eval(BigEvalArgs a) {
    const uint32_t v = blockIdx.x * 64u + threadIdx.x, i = blockIdx.y;
    unsigned long long s = 0;
    if (v < a.n + a.n_wide) {
        nhdfit_wide_node view;
        if (v < a.n) wide_view(a.p0[v], a.p1[v], a.p2[v], a.p3[v], a.p4[v], a.det[v], v, view);
        else view = a.wide[v - a.n];
        const bool listed = !a.cand || (a.cand[view.index >> 6] >> (view.index & 63) & 1ull);
        if (listed && view.numa_nodes) {           // (a placeholder of the planes has no NUMA nodes: its record answers, further down the grid)
            const nhdfit_big_req& r = a.reqs[i];
            NicSearch ns{NHDFIT_BIG_NIC_BUDGET, false};
            const bool ok = wide_fits(view, r, view.busy_time >= a.busy_from, WideCaps(a.caps, a.share && v >= a.n ? a.share + (v - a.n) : nullptr), &ns);
            if (ns.exhausted) atomicOr(&a.flags[1], 1u);
            if (ns.left != NHDFIT_BIG_NIC_BUDGET) atomicMax(&a.flags[2], NHDFIT_BIG_NIC_BUDGET - ns.left);   // the deepest NIC search of the call (nhdfit_stats)
            if (ok) {
                uint32_t want = 0;
                for (uint32_t g = 0; g < r.n_groups; ++g) want += r.gpus[g];
                s = (unsigned long long)score_of(want == 0 && view.n_gpus == 0, a.global_base + view.index);   // SelectNode, Matcher.py:401-421
            }
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(s, d, 64);
        s = o > s ? o : s;
    }
    if (threadIdx.x == 0 && s) atomicMax(&a.score[i], s);
}

struct BigMapArgs {
    const nhdfit_plane0* p0; const nhdfit_plane1* p1; const nhdfit_plane2* p2; const nhdfit_plane3* p3; const nhdfit_plane4* p4;
    const nhdfit_detail* det; uint32_t n;
    const nhdfit_wide_node* wide; uint32_t n_wide;
    const nhdfit_big_req* reqs; uint32_t P;
    const double* caps;
    const unsigned long long* score; uint64_t global_base;
    nhdfit_big_mapping* out;
    int32_t* scratch;                              // [workers][stride]: the set tables of one mapping each (wide_core.h big_scratch_words)
    size_t stride; int32_t slots_g, slots_c; uint32_t workers;
    uint32_t lds_tables;                           // the set tables of a mapping fit the block's LDS (launched with stride * 4 bytes of it)
    uint32_t* flags;
    const nhdfit_wide_share* share;                // optional [n_wide]
};
__global__ __launch_bounds__(64) void k_big_map(BigMapArgs a) {
    extern __shared__ __align__(16) int32_t s_tables[];
    __shared__ BigWaveLds s_wave;
    const uint32_t tid = blockIdx.x, lane = threadIdx.x;          // one mapping per wavefront at a time
    if (tid >= a.workers) return;
    // The set model's walk is a chain of dependent probes into the sets' tables: in LDS when they fit (two sockets, eight groups: 28 KB - a probe
    // costs an LDS round trip instead of one to L2), in the call's scratch memory otherwise (nodes with more sockets)
    int32_t* scratch = a.lds_tables ? s_tables : a.scratch + (size_t)tid * a.stride;
    for (uint32_t i = tid; i < a.P; i += a.workers) {
        nhdfit_big_mapping m;
        wide_map_clear(m, NHDFIT_BIG_MAX_GROUPS);
        const unsigned long long s = a.score[i];
        const uint64_t gi = s ? NHDFIT_SCORE_INDEX(s) : 0;
        if (s && gi >= a.global_base && gi < a.global_base + a.n) {          // this shard's node (else: another shard maps it)
            const uint32_t v = (uint32_t)(gi - a.global_base);
            const int slot = a.n_wide ? wide_slot_of(a.wide, a.n_wide, v) : -1;
            if (lane == 0) {
                if (slot >= 0) s_wave.view = a.wide[slot];
                else wide_view(a.p0[v], a.p1[v], a.p2[v], a.p3[v], a.p4[v], a.det[v], v, s_wave.view);
            }
            reinterpret_cast<uint32_t*>(&s_wave.req)[lane] = reinterpret_cast<const uint32_t*>(&a.reqs[i])[lane];     // (256 bytes: a dword per lane)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const WideCaps caps(a.caps, a.share && slot >= 0 ? a.share + slot : nullptr);
            // lane = tuple where the tuples fit the LDS tables (every node of the fast layout: two NUMA nodes); else one thread, as before
            const bool wave = a.lds_tables && s_wave.view.numa_nodes <= 2u;
            int rc = 0;
            if (wave) rc = wide_map_wave(s_wave, caps, scratch, m, a.slots_g, a.slots_c, lane);
            else if (lane == 0) rc = wide_map(s_wave.view, s_wave.req, caps, scratch, m, a.slots_g, a.slots_c);
            if (lane == 0 && rc < 0) { m.valid = 0; atomicOr(&a.flags[rc == -2 ? 1 : 0], 1u); }
        }
        if (lane == 0) a.out[i] = m;
        __builtin_amdgcn_wave_barrier();                                      // (the LDS record is the next mapping's)
    }
}

struct BigCommitArgs {
    nhdfit_plane0* p0; nhdfit_plane1* p1; nhdfit_plane2* p2; nhdfit_plane3* p3; nhdfit_plane4* p4; nhdfit_detail* det;
    nhdfit_wide_node* wide; int slot;              // slot >= 0: the node is that wide record
    uint32_t node; nhdfit_big_req req; nhdfit_big_mapping map; double busy_time; SigTable sigs;
    nhdfit_big_placement* out;
    nhdfit_wide_share* share;                      // optional [n_wide]
};
__global__ __launch_bounds__(64) void k_big_commit(BigCommitArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const nhdfit_big_req& r = a.req;
    const nhdfit_big_mapping& m = a.map;
    nhdfit_big_placement pl;
    memset(&pl, 0, sizeof pl);
    if (a.slot >= 0) {
        nhdfit_wide_node n = a.wide[a.slot];
        wide_commit(n, r, m, a.busy_time, pl, a.share ? a.share + a.slot : nullptr);
        pl.pod = 0; pl.node = n.index;
        a.wide[a.slot] = n;
        *a.out = pl;
        return;
    }
    NodeState s;
    s.p0 = a.p0[a.node]; s.p1 = a.p1[a.node]; s.p2 = a.p2[a.node]; s.p3 = a.p3[a.node]; s.p4 = a.p4[a.node];
    nhdfit_detail d = a.det[a.node];
    for (uint32_t g = 0; g < r.n_groups; ++g)                          // GetNicObjFromIndex returns None: IndexError before anything
        if ((uint32_t)m.nic_idx[g] >= d.nic_cnt[m.nic_numa[g] & 1]) {  // of that group is touched (nhd/Node.py:700-704); the mirror is left alone
            pl.status = kCommitWouldRaise;
            pl.node = a.node;
            *a.out = pl;
            return;
        }
    commit_node_t<nhdfit_big_req, nhdfit_big_placement>(s, d, r, m, a.busy_time, a.sigs, pl);
    pl.pod = 0; pl.node = a.node;
    a.p0[a.node] = s.p0; a.p1[a.node] = s.p1; a.p2[a.node] = s.p2; a.p3[a.node] = s.p3; a.p4[a.node] = s.p4;
    a.det[a.node] = d;
    *a.out = pl;
}
