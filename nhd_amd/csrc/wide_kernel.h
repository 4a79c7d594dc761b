// wide_kernel.h - the general path on the device: nodes beyond the fast layout (wide_core.h) against the staged pods.
// Device code of libnhdfit.so; included by nhdfit.hip inside its anonymous namespace.  gfx950 only.
//
// k_wide_eval runs right behind the fit role of a step, on the same stream: lane = (wide node, pod).  Its verdict bit
// goes into the step's node-major verdict matrix (the fit role wrote zeros for the node's placeholder), its score into the
// step's score words with atomicMax - the word carries the global node index, so the first feasible node of the whole
// candidate order wins wherever it is mirrored, and a sharded run's all-reduce sees the merged scores.
// k_wide_map runs when results are fetched: a pod whose winner is a wide node gets its mapping from the general set model.
// k_wide_commit: the commit step for one placement on a wide node.
struct WideArgs {
    const nhdfit_wide_node* wide; uint32_t n_wide;
    const nhdfit_req* reqs; uint32_t P;            // staged (class-sorted) order
    const double* caps; double busy_from;
    const uint64_t* cand;                          // optional [chunks] candidate nodes
    unsigned long long* nm; uint32_t chunks;       // optional node-major verdict words [tiles][chunks * 64]
    unsigned long long* score; uint64_t global_base;
    const nhdfit_wide_share* share;                // optional [n_wide]: ENABLE_SHARING arithmetic (include/nhdfit.h)
};

__global__ __launch_bounds__(256) void k_wide_eval(WideArgs a) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (uint64_t)a.n_wide * a.P) return;
    const uint32_t w = (uint32_t)(idx / a.P), i = (uint32_t)(idx % a.P);   // consecutive lanes: consecutive pods of one node
    const nhdfit_wide_node& n = a.wide[w];
    if (a.cand && !(a.cand[n.index >> 6] >> (n.index & 63) & 1ull)) return;
    const nhdfit_req& r = a.reqs[i];
    const bool busy = n.busy_time >= a.busy_from;                          // IsBusy, as the fit role asks it (fit_core.h busy_threshold)
    if (!wide_fits(n, r, busy, WideCaps(a.caps, a.share ? a.share + w : nullptr))) return;
    if (a.nm) atomicOr(&a.nm[(size_t)(i >> 6) * a.chunks * 64 + n.index], 1ull << (i & 63));
    uint32_t want = 0;
    for (uint32_t g = 0; g < r.n_groups; ++g) want += r.gpus[g];
    atomicMax(&a.score[i], (unsigned long long)score_of(want == 0 && n.n_gpus == 0, a.global_base + n.index));   // SelectNode, Matcher.py:401-421
}

struct WideMapArgs {
    const nhdfit_wide_node* wide; uint32_t n_wide;
    const nhdfit_req* reqs; uint32_t P;
    const double* caps;
    const unsigned long long* score; uint64_t global_base; uint32_t n;
    nhdfit_mapping* out;
    int16_t* scratch;                              // [threads][kWideScratchWords]
    uint32_t* flags;                               // [0] a set of the model outgrew its table (never expected)
    const nhdfit_wide_share* share;                // optional [n_wide]
};
__device__ inline int wide_slot_of(const nhdfit_wide_node* wide, uint32_t n_wide, uint32_t index) {   // records are sorted by index
    int lo = 0, hi = (int)n_wide - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const uint32_t v = wide[mid].index;
        if (v == index) return mid;
        if (v < index) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}
__global__ __launch_bounds__(64) void k_wide_map(WideMapArgs a) {
    // one worker per wavefront (its first lane): the set model is a long serial walk - lanes of one wavefront walking different
    // pods' sets would only take turns
    if (threadIdx.x != 0) return;
    const uint32_t tid = blockIdx.x, nthreads = gridDim.x;
    int16_t* scratch = a.scratch + (size_t)tid * kWideScratchWords;
    for (uint32_t i = tid; i < a.P; i += nthreads) {
        const unsigned long long s = a.score[i];
        if (!s) continue;
        const uint64_t gi = NHDFIT_SCORE_INDEX(s);
        if (gi < a.global_base || gi >= a.global_base + a.n) continue;      // another shard's node
        const int slot = wide_slot_of(a.wide, a.n_wide, (uint32_t)(gi - a.global_base));
        if (slot < 0) continue;                                             // an ordinary node: the mapping roles answered
        nhdfit_mapping m;
        const int rc = wide_map(a.wide[slot], a.reqs[i], WideCaps(a.caps, a.share ? a.share + slot : nullptr), scratch, m);
        if (rc < 0) { m.valid = 0; atomicOr(&a.flags[0], 1u); }
        a.out[i] = m;
    }
}

struct WideCommitArgs {
    nhdfit_wide_node* wide; uint32_t slot;
    nhdfit_req req; nhdfit_mapping map; double busy_time;
    nhdfit_wide_placement* out;
    nhdfit_wide_share* share;                      // optional [n_wide]
};
__global__ void k_wide_commit(WideCommitArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    nhdfit_wide_node n = a.wide[a.slot];
    nhdfit_wide_placement pl;
    wide_commit(n, a.req, a.map, a.busy_time, pl, a.share ? a.share + a.slot : nullptr);
    pl.pod = 0; pl.node = n.index;
    a.wide[a.slot] = n;
    *a.out = pl;
}
