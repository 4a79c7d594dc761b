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
    const uint32_t tid = blockIdx.x;                // one worker per wavefront, its first lane: the set model is one long serial walk, and
    if (threadIdx.x != 0 || tid >= a.workers) return;   // lanes walking different pods' sets would only take turns inside a wavefront
    // The walk is a chain of dependent probes into the sets' tables: in LDS when they fit (two sockets, eight groups: 28 KB - a probe
    // costs an LDS round trip instead of one to L2; round 5), in the call's scratch memory otherwise (nodes with more sockets)
    int32_t* scratch = a.lds_tables ? s_tables : a.scratch + (size_t)tid * a.stride;
    for (uint32_t i = tid; i < a.P; i += a.workers) {
        nhdfit_big_mapping m;
        for (int g = 0; g < NHDFIT_BIG_MAX_GROUPS; ++g) { m.gpu[g] = m.nic_numa[g] = m.nic_idx[g] = -1; }
        for (int g = 0; g <= NHDFIT_BIG_MAX_GROUPS; ++g) m.cpu[g] = -1;
        m.valid = 0; m.pad[0] = m.pad[1] = 0;
        const unsigned long long s = a.score[i];
        const uint64_t gi = s ? NHDFIT_SCORE_INDEX(s) : 0;
        if (s && gi >= a.global_base && gi < a.global_base + a.n) {          // this shard's node (else: another shard maps it)
            const uint32_t v = (uint32_t)(gi - a.global_base);
            nhdfit_wide_node view;
            const int slot = a.n_wide ? wide_slot_of(a.wide, a.n_wide, v) : -1;
            if (slot >= 0) view = a.wide[slot];
            else wide_view(a.p0[v], a.p1[v], a.p2[v], a.p3[v], a.p4[v], a.det[v], v, view);
            const int rc = wide_map(view, a.reqs[i], WideCaps(a.caps, a.share && slot >= 0 ? a.share + slot : nullptr), scratch, m, a.slots_g, a.slots_c);
            if (rc < 0) { m.valid = 0; atomicOr(&a.flags[rc == -2 ? 1 : 0], 1u); }
        }
        a.out[i] = m;
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
