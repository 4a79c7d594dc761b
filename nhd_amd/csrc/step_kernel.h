// step_kernel.h - k_step itself (five roles, one launch), its per-role stand-alone form and the node-record kernels.
// Device code of libnhdfit.so; included by nhdfit.hip inside its anonymous namespace, in this order: step_digest.h,
// step_fit.h, step_map.h, step_kernel.h, seq_kernel.h (one translation unit: the roles are fused into one kernel).
// gfx950 only.
// ---- the step kernel ---------------------------------------------------------------------------------
// ONE launch per step.  The grid is the union of five block ranges ("roles") that work on five different steps of
// the software pipeline:  [choose(i-2) | shapes(i-1) | finish(i-3) | digest(i+1) | fit(i)].  The four side roles
// are short on work and long on latency (sequential set model, dependent look-ups); scheduled first, they run in
// the shadow of the chip-filling fit role instead of serialising the stream with ~20-40 us kernels of their own.
// Dependencies only cross launches (stream order).  A role with zero blocks is simply absent: the same kernel
// serves a single find (five launches, one role each) and the pipeline flush.
struct StepArgs {
    uint32_t nb_fit, nb_choose, nb_shapes, nb_finish, nb_digest;     // blocks per role, in grid order
    uint32_t shapes_P;                                       // pods (= upper bound of the choose role's shape slots)
    uint32_t side_prio;                                      // raise the side roles' issue priority
    uint32_t choose_lanes;                                   // choose role: wavefront = tile, lane = shape (else a wavefront per shape)
    ShapeArgs choose;
    MapArgs shapes_m; ShapeArgs shapes_h;
    MapArgs finish_m; ShapeArgs finish_h;
    DigestArgs digest;
    FitArgs fit;
    unsigned long long* role_clock;      // profiling aid (NHDFIT_ROLE_TIMES): [5][2] first start / last end per role, 100 MHz ticks
};

__device__ __forceinline__ void stamp(unsigned long long* role_clock, int role, unsigned long long t0) {
    if (role_clock && threadIdx.x == 0) {
        atomicMin(&role_clock[2 * role], t0);
        atomicMax(&role_clock[2 * role + 1], (unsigned long long)wall_clock64());
    }
}

#ifdef NHDFIT_TUNING      // profiling aid of the tuning build (NHDFIT_SPLIT): the fit role as a launch of its own
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_fit_only(FitArgs a) {
    extern __shared__ __align__(16) uint8_t lds[];
    role_fit<BLOCK>(a, a.busy_from, blockIdx.x, lds);
}
#endif

// ---- node records ------------------------------------------------------------------------------------
// Everything the fit role needs from a node's five planes depends only on the mirror and the dictionary, not on the
// pod tile or the step.  After nodes change: (1) k_xkeys interns the (NUMA, free GPUs, NUMA-mode signature,
// PCI-mode signature) class of both NUMA nodes of every touched node in a device hash table, (2) k_xassign gives new
// classes the next X row, (3) k_xrecords writes the 16-byte records (one array per row width).  Classes are never
// removed, rows are provisioned in powers of two: records stay valid while classes are appended.
constexpr uint32_t kXSlots = 1u << 15;          // open-addressing table; the host grows nothing: > kXSlots / 2 classes is NHDFIT_E_LIMIT
struct XTable {
    unsigned long long* key;                    // [kXSlots], 0 = empty
    uint32_t* id;                               // [kXSlots], ~0u = not assigned yet
    uint64_t* cls;                              // [kXSlots / 2]: key of X row k
    uint32_t* nx;                               // [0] = classes, [1] = overflow flag
};
struct RecArgs {
    const nhdfit_plane0* p0; const nhdfit_plane1* p1; const nhdfit_plane2* p2; const nhdfit_plane3* p3; const nhdfit_plane4* p4;
    uint32_t n, npad;                           // nodes / nodes rounded up to whole chunks
    uint32_t first, count;                      // nodes to (re)do
    uint32_t fc_dim, fg_dim, ngs;
    XTable x;
    Layout L[kWClasses];
    NodeRec* rec[kWClasses];
    double* bt[kWClasses];                      // busy times beside the records, same order
    uint32_t crow_D[2];                         // pair-table dimension the records' C rows are written for (NodeRec::flags), per narrow row width
};
__device__ __forceinline__ uint32_t xhash(uint64_t k) {
    k ^= k >> 33; k *= 0xFF51AFD7ED558CCDull; k ^= k >> 29;
    return (uint32_t)k & (kXSlots - 1);
}
__device__ __forceinline__ uint32_t xslot_find(const XTable& x, uint64_t key) {        // the key is present
    uint32_t s = xhash(key);
    while (x.key[s] != key) s = (s + 1) & (kXSlots - 1);
    return s;
}
__global__ __launch_bounds__(256) void k_xkeys(RecArgs a) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= a.count) return;
    const uint32_t i = a.first + t;
    if (i >= a.n) return;
    const NodeIdx n = node_index(a.p0[i], a.p1[i], a.p2[i], a.p4[i], a.fc_dim, a.fg_dim, a.ngs);
    const nhdfit_plane3 q3 = a.p3[i];
    for (uint32_t u = 0; u < 2; ++u) {
        const unsigned long long key = xkey(u, u ? n.f1 : n.f0, q3.sig_numa[u], q3.sig_pci[u]);
        // a cluster has a few hundred distinct classes at most: all but the first insert of a class find it with a plain
        // (L2-coherent) read and never issue the compare-and-swap - 131 072 CAS on ~100 words serialised in L2 before
        uint32_t s = xhash(key);
        for (uint32_t probes = 0; probes < kXSlots; ++probes, s = (s + 1) & (kXSlots - 1)) {
            unsigned long long prev = __hip_atomic_load(&a.x.key[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == key) break;
            if (prev == 0ull) prev = atomicCAS(&a.x.key[s], 0ull, key);
            if (prev == 0ull || prev == key) break;
        }
    }
}
__global__ __launch_bounds__(1024) void k_xassign(XTable x) {      // one block: new classes get rows in slot order
    for (uint32_t s = threadIdx.x; s < kXSlots; s += 1024)
        if (x.key[s] != 0ull && x.id[s] == ~0u) {
            const uint32_t k = atomicAdd(&x.nx[0], 1u);
            if (k < kXSlots / 2) { x.id[s] = k; x.cls[k] = x.key[s]; }
            else { x.id[s] = 0; x.nx[1] = 1; }
        }
}
__global__ __launch_bounds__(256) void k_xrecords(RecArgs a) {      // (first / count: whole chunks - the records are written in node order)
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= a.count) return;
    const uint32_t i = a.first + t;
    if (i >= a.npad) return;
    if (i >= a.n) {                                                // padding of the last chunk
        for (int w = 0; w < kWClasses; ++w) { a.rec[w][i] = dead_record(a.L[w], i & 63u); a.bt[w][i] = 0.0; }
        return;
    }
    const nhdfit_plane4 q4 = a.p4[i];
    const NodeIdx n = node_index(a.p0[i], a.p1[i], a.p2[i], q4, a.fc_dim, a.fg_dim, a.ngs);
    const nhdfit_plane3 q3 = a.p3[i];
    const uint32_t x0 = a.x.id[xslot_find(a.x, xkey(0, n.f0, q3.sig_numa[0], q3.sig_pci[0]))];
    const uint32_t x1 = a.x.id[xslot_find(a.x, xkey(1, n.f1, q3.sig_numa[1], q3.sig_pci[1]))];
    for (int w = 0; w < kWClasses; ++w) { a.rec[w][i] = make_record(n, x0, x1, a.L[w], i & 63u, w < 2 ? a.crow_D[w] : 0u); a.bt[w][i] = q4.busy_time; }
}
// the records' C rows for another pair-table dimension (a staged batch changed it): the flags word of every record of the two narrow widths
struct CrowArgs { NodeRec* rec[2]; uint32_t npad; uint32_t D[2]; };
__global__ __launch_bounds__(256) void k_xcrow(CrowArgs a) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.npad) return;
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        NodeRec* r = a.rec[w] + i;
        r->flags = (uint16_t)((r->flags & kRecNoGpu) | (a.D[w] ? pair_c_row(r->cc, a.D[w]) << 1 : 0u));
    }
}

// ---- lane order of a chunk (fit_core.h NodeRec "LANE ORDER") ----------------------------------------------------------------
// The fit role's sweep is a gather: every lane fetches the table rows its node's record names, and a ds_read_b128 is serviced 16
// lanes at a time (four fixed lane groups) from 16 bank groups of 16 bytes - lanes of one group whose rows differ but share a
// bank group take turns.  With the nodes in index order the rows are as good as random: 16 picks among 16 bank groups pile up
// three deep, 39-44 % of the LDS cycles of a step were such turns (profiles/r04/rocprof_pmc_summary.txt).  Which lane works on
// which node of a chunk is free, so the chunk's 64 records are dealt to the four lane groups greedily, one after the other,
// each to the group where it adds the fewest turns: per fetch family (the C row, the two class rows; for the wide tiles the two
// sockets' CPU rows and the two class rows) a group keeps, per bank group, how many DIFFERENT rows it already fetches there -
// equal rows are one broadcast.  One wavefront per (chunk, row width): lane = record, the counters live in the lanes of four
// registers (lane = lane group x 16 + bank group).  Any order gives the same verdicts; this one gives fewer LDS cycles
// (tools/lds_bank_model.py on config 4's cluster: 16.2 -> 13.6 cycles per chunk at two assignments, 39 -> 30 at four, 160 -> 112
// at eight).  Runs behind k_xrecords over the chunks it rewrote, and over every chunk when a staged batch changes the pair
// table's dimension (the C row of a node depends on it).
struct OrderArgs {
    NodeRec* rec[kWClasses];
    double* bt[kWClasses];
    uint32_t first_chunk, n_chunks;
    uint32_t pair_D[2];                             // as FitArgs: 0 = that width sweeps six rows
};
__device__ __forceinline__ uint32_t b128_lane(uint32_t group, uint32_t idx) {      // idx-th lane of a ds_read_b128 lane group (MI355X_MICROARCH.md, LDS)
    const uint32_t in_half = (group & 1u) == 0u ? (idx < 4u ? idx : idx < 8u ? idx + 8u : idx + 12u)     // {0-3, 12-15, 20-27}
                                                : (idx < 8u ? idx + 4u : idx < 12u ? idx + 8u : idx + 16u);   // {4-11, 16-19, 28-31}
    return in_half + (group >> 1) * 32u;
}
__global__ __launch_bounds__(256) void k_xorder(OrderArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t task = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));
    if (task >= a.n_chunks * (uint32_t)kWClasses) return;
    const uint32_t w = task % (uint32_t)kWClasses, c = a.first_chunk + task / (uint32_t)kWClasses;
    NodeRec* __restrict__ recs = a.rec[w] + (size_t)c * 64;
    double* __restrict__ bts = a.bt[w] + (size_t)c * 64;
    const NodeRec r = recs[lane];
    const double bt = bts[lane];
    // fetch families: (row id, weight); the bank group of a row is its id mod 16 up to a constant per family
    const uint32_t D = w < 2u ? a.pair_D[w] : 0u;
    constexpr int K = 4;
    uint32_t key[K], wt[K];
    if (D) { key[0] = pair_c_row(r.cc, D); key[1] = r.x0 >> 1; key[2] = r.x1 >> 1; key[3] = 0; wt[0] = wt[1] = wt[2] = 1; wt[3] = 0; }
    else   { key[0] = r.w0 >> 1; key[1] = r.w1 >> 1; key[2] = r.x0 >> 1; key[3] = r.x1 >> 1; wt[0] = wt[1] = 2; wt[2] = wt[3] = 1; }
    uint32_t load[K] = {0, 0, 0, 0};               // lane g * 16 + s: different rows of family k group g fetches from bank group s
    uint32_t deepest[4][K];                         // (uniform) the deepest bank group of (group, family): what a fetch of that family costs the group
    uint32_t members[4] = {0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < K; ++k) deepest[g][k] = 1;
    uint32_t my_group = 0xFFu, my_idx = 0;
    for (uint32_t i = 0; i < 64u; ++i) {
        uint32_t ki[K];
#pragma unroll
        for (int k = 0; k < K; ++k) ki[k] = (uint32_t)__builtin_amdgcn_readlane((int)key[k], (int)i);
        uint32_t best_cost = ~0u, best_g = 0, best_new[K] = {0, 0, 0, 0};
        uint32_t dup_bits = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint32_t inc = 0, fresh[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const bool dup = __ballot(my_group == (uint32_t)g && key[k] == ki[k]) != 0ull;      // the group fetches this very row already
                const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)load[k], (int)((uint32_t)g * 16u + (ki[k] & 15u)));
                fresh[k] = cur + (dup ? 0u : 1u);
                if (wt[k] && fresh[k] > deepest[g][k]) inc += wt[k] * (fresh[k] - deepest[g][k]);
                if (dup) dup_bits |= 1u << (g * K + k);
            }
            const uint32_t cost = members[g] >= 16u ? ~0u : inc * 32u + members[g];
            if (cost < best_cost) {
                best_cost = cost; best_g = (uint32_t)g;
#pragma unroll
                for (int k = 0; k < K; ++k) best_new[k] = fresh[k];
            }
        }
        uint32_t idx = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if ((uint32_t)g == best_g) {
                idx = members[g]++;
#pragma unroll
                for (int k = 0; k < K; ++k) if (best_new[k] > deepest[g][k]) deepest[g][k] = best_new[k];
            }
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (!(dup_bits >> (best_g * K + k) & 1u) && lane == best_g * 16u + (ki[k] & 15u)) load[k] += 1u;
        if (lane == i) { my_group = best_g; my_idx = idx; }
    }
    const uint32_t to = b128_lane(my_group, my_idx);
    // a permutation, or the chunk stays as it is
    uint64_t seen = 1ull << to;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) seen |= __shfl_xor(seen, m, 64);
    if (seen != ~0ull) return;
    recs[to] = r;                                   // (every lane read its record before any lane writes)
    bts[to] = bt;
}

// node-major verdict words [tiles][chunks*64] -> pod-major rows [chunks][P] (the layout of nhdfit_find's
// bitmap_out and of the sequential resolver): one wavefront per (tile, chunk), 64 x 64 bit transpose in registers
__global__ __launch_bounds__(256) void k_rows(const uint64_t* __restrict__ nm, uint64_t* __restrict__ rows, uint32_t chunks, uint32_t P) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t tiles = (P + kTile - 1) / kTile;
    if (w >= tiles * chunks) return;
    const uint32_t tile = w / chunks, c = w % chunks;
    const uint64_t v = nm[((size_t)tile * chunks + c) * 64 + lane];
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    transpose64(lo, hi);
    const uint32_t pod = tile * kTile + lane;
    if (pod < P) rows[(size_t)c * P + pod] = ((uint64_t)hi << 32) | lo;
}

// the same matrix for the sequential kernels, [P][chunks]: a pod's row contiguous.  One wavefront per (tile, 16 chunks): sixteen
// 64 x 64 bit transposes in registers, staged through LDS (17-word rows: lane = pod hits 64 banks), written as 128-byte runs
constexpr uint32_t kRowsTChunks = 16;
__global__ __launch_bounds__(256) void k_rows_t(const uint64_t* __restrict__ nm, uint64_t* __restrict__ rows, uint32_t chunks, uint32_t P) {
    __shared__ uint64_t s_t[4][64][kRowsTChunks + 1];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t groups = (chunks + kRowsTChunks - 1) / kRowsTChunks;
    const uint32_t tiles = (P + kTile - 1) / kTile;
    const uint32_t w = blockIdx.x * 4 + wv;
    if (w >= tiles * groups) return;
    const uint32_t tile = w / groups, c0 = (w % groups) * kRowsTChunks;
#pragma unroll 4
    for (uint32_t k = 0; k < kRowsTChunks; ++k) {
        const uint32_t c = c0 + k;
        const uint64_t v = c < chunks ? nm[((size_t)tile * chunks + c) * 64 + lane] : 0ull;
        uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
        transpose64(lo, hi);
        s_t[wv][lane][k] = ((uint64_t)hi << 32) | lo;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t ck = lane & (kRowsTChunks - 1);
    for (uint32_t r = 0; r < 64; r += 64 / kRowsTChunks) {
        const uint32_t j = r + lane / kRowsTChunks, pod = tile * kTile + j;
        if (pod < P && c0 + ck < chunks) rows[(size_t)pod * chunks + c0 + ck] = s_t[wv][j][ck];
    }
}

template <int BLOCK, bool SPILL>      // SPILL: some tiles stage only a prefix of their hot section (refresh_layouts)
__device__ __forceinline__ void step_body(const StepArgs& a, const double busy_from, uint8_t* lds) {
    uint32_t blk = blockIdx.x;
    const unsigned long long t0 = a.role_clock ? (unsigned long long)wall_clock64() : 0ull;
    // Grid order: the fit role first.  Its blocks are sized to fill two of the three block slots of every CU, so all of
    // them start at once; the side roles (short latency chains on few wavefronts) take the third slot, with issue
    // priority so that they finish - and hand the slot on - sooner.
    if (blk < a.nb_fit) {
        role_fit<BLOCK, SPILL>(a.fit, busy_from, blk, lds);
        stamp(a.role_clock, 4, t0);
        return;
    }
    blk -= a.nb_fit;
    if (a.side_prio) __builtin_amdgcn_s_setprio(3);
    if (blk < a.nb_choose) {
        if (a.choose_lanes)
            role_choose_lanes(a.choose, (uint32_t)__builtin_amdgcn_readfirstlane((int)(blk * (BLOCK / 64) + (threadIdx.x >> 6))), (a.shapes_P + kTile - 1) / kTile);
        else if ((threadIdx.x & 63) == 0)
            role_choose(a.choose, (uint32_t)__builtin_amdgcn_readfirstlane((int)(blk * (BLOCK / 64) + (threadIdx.x >> 6))),
                        a.nb_choose * (BLOCK / 64), (a.shapes_P + kTile - 1) / kTile);
        stamp(a.role_clock, 0, t0);
        return;
    }
    blk -= a.nb_choose;
    if (blk < a.nb_shapes) { role_shapes<BLOCK>(a.shapes_m, a.shapes_h, blk, lds); stamp(a.role_clock, 1, t0); return; }
    blk -= a.nb_shapes;
    if (blk < a.nb_finish) { role_finish<BLOCK>(a.finish_m, a.finish_h, blk, lds); stamp(a.role_clock, 2, t0); return; }
    blk -= a.nb_finish;
    role_digest<BLOCK>(a.digest, blk, lds);
    stamp(a.role_clock, 3, t0);
}

// The step launch.  (The ~3 KB argument block travels by value: a device-resident copy behind a pointer was measured in
// round 3 - no gain, the scalar loads through the pointer cost what the kernarg fetch saved: 236 spilled SGPRs against 19.)
#ifndef NHDFIT_STEP_WAVES
#define NHDFIT_STEP_WAVES 6          // wavefronts per SIMD the 512-thread form is compiled for (A/B builds: 8 = four blocks per CU, <= 64 VGPRs)
#endif
template <int BLOCK, bool SPILL = false>
__global__ __launch_bounds__(BLOCK, SPILL ? (BLOCK == 512 ? 4 : 5) : (BLOCK == 512 ? NHDFIT_STEP_WAVES : 7)) void k_step(StepArgs a) {   // SPILL: two blocks per CU (LDS)   // 512 threads = 2 waves per SIMD: 3 blocks per CU either way
    extern __shared__ __align__(16) uint8_t lds[];
    step_body<BLOCK, SPILL>(a, a.fit.busy_from, lds);
}

#ifdef NHDFIT_TUNING
// Profiling aid of the tuning build (NHDFIT_ROLE_KERNELS=1): one role per launch, so that rocprofv3 --stats names each role's
// stand-alone time.  Not in libnhdfit.so.
template <int BLOCK, int ROLE>
__global__ __launch_bounds__(BLOCK, 6) void k_role(StepArgs a) {
    extern __shared__ __align__(16) uint8_t lds[];
    const uint32_t blk = blockIdx.x;
    if constexpr (ROLE == 0) {
        if (a.choose_lanes)
            role_choose_lanes(a.choose, (uint32_t)__builtin_amdgcn_readfirstlane((int)(blk * (BLOCK / 64) + (threadIdx.x >> 6))), (a.shapes_P + kTile - 1) / kTile);
        else if ((threadIdx.x & 63) == 0)
            role_choose(a.choose, (uint32_t)__builtin_amdgcn_readfirstlane((int)(blk * (BLOCK / 64) + (threadIdx.x >> 6))),
                        a.nb_choose * (BLOCK / 64), (a.shapes_P + kTile - 1) / kTile);
    } else if constexpr (ROLE == 1) role_shapes<BLOCK>(a.shapes_m, a.shapes_h, blk, lds);
    else if constexpr (ROLE == 2) role_finish<BLOCK>(a.finish_m, a.finish_h, blk, lds);
    else if constexpr (ROLE == 3) role_digest<BLOCK>(a.digest, blk, lds);
    else role_fit<BLOCK>(a.fit, a.fit.busy_from, blk, lds);
}
#endif

// ---- hand-offs between the blocks of one launch (k_find, k_find1, k_findn) -----------------------------------------------------------------
// gfx950: a CU's L1 is never refreshed by other CUs' stores, the XCDs' L2s are not coherent with each other.  What a block publishes -
// plain stores - becomes visible through ONE agent-scope release by one lane behind the block's barrier (every storing wave drains its
// stores first), then a relaxed agent-scope counter; the consumer polls that counter RELAXED (an acquire per poll costs ~1.7 us each and
// a few hundred pollers take a third of the chip's bandwidth), ONE lane acquires once behind the match, a barrier, then plain loads.
// Agent-scope atomics (the fit role's atomicMax on the score words) need no release: they are performed at the coherence point, the
// wave waits for them (vmcnt) and the counter follows.  The fences are per lane, not per thread: 256 threads fencing cost 2-4 x one.
#define NHDFIT_DRAIN_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// every thread of the block calls it behind its last store; returns the counter's value before this block's arrival (thread 0 only)
__device__ __forceinline__ uint32_t publish_and_count(uint32_t* counter, bool plain_stores) {
    NHDFIT_DRAIN_VMEM();
    __syncthreads();
    uint32_t before = 0;
    if (threadIdx.x == 0) {
        if (plain_stores) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); NHDFIT_DRAIN_VMEM(); }
        before = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        NHDFIT_DRAIN_VMEM();
    }
    return before;
}
// thread 0 only: wait until *counter >= want (false: gave up), then the one acquire
__device__ __forceinline__ bool poll_then_acquire(const uint32_t* counter, uint32_t want, uint32_t limit) {
    bool ok = true;
    for (uint32_t spin = 0; __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want; ++spin) {
        if (spin > limit) { ok = false; break; }
        __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok;
}

// ---- one launch for a small find ---------------------------------------------------------------------
// nhdfit_find for at most one pod tile (the scheduler's pod-at-a-time FindNode): the five roles of a step in ONE launch
// instead of five launches in a row, requests read straight from page-locked host memory, results stored straight into
// it - no copy engine on either side, no launch gaps: the call's latency is one launch plus the chain of the roles.
//   * blocks [0, nb_digest): the digest of the tile; each counts itself in sync[0] when its rows are out.
//   * blocks [nb_digest, +nb_fit): fit blocks, chunk range by block index.  They wait for sync[0] == nb_digest (the digest
//     blocks lead the grid and wait for nothing, so they are always running or done when a fit block spins), sweep, and
//     take a ticket in sync[1].
//   * the fit block with the last ticket sees every score final: it maps the tile's winners (map_one_tile, step_map.h), stores
//     scores and mappings into the host block and then the call's sequence number behind them (system scope); the host
//     polls that word.  It also leaves the two counters at zero for the next call.
// A wait that does not end (it cannot, short of a fault elsewhere) gives up after kFindSpinLimit polls: the launch then
// reports kFindAborted instead of the sequence number and the host takes the five-launch path.
struct FindHost {                                    // one per context, hipHostMallocCoherent (fine-grained: visible while the kernel runs)
    uint32_t flag;                                   // sequence number of the call whose results are below / kFindAborted
    uint32_t pad[3];
    unsigned long long score[kTile];                 // out
    nhdfit_mapping maps[kTile];                      // out
    nhdfit_req reqs[kTile];                          // in
};
constexpr uint32_t kFindAborted = 0xFFFFFFFFu;
constexpr uint32_t kFindSpinLimit = 1u << 16;
struct FindArgs {
    StepArgs s;                                      // of it: digest, fit, finish_m / finish_h (the mapping tail), nb_digest, nb_fit, shapes_P, role_clock
    uint32_t wcls, want_map;
    uint32_t* sync;                                  // [0] digest blocks done, [1] fit tickets, [2] a wait gave up; zero between launches
    FindHost* host;
    uint32_t seq;
};
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_find(FindArgs a) {
    extern __shared__ __align__(16) uint8_t lds[];
    uint32_t blk = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const unsigned long long t0 = a.s.role_clock ? (unsigned long long)wall_clock64() : 0ull;
    if (blk < a.s.nb_digest) {
        role_digest<BLOCK>(a.s.digest, blk, lds);
        stamp(a.s.role_clock, 3, t0);
        (void)publish_and_count(&a.sync[0], true);                           // (rows, headers, zeroed scores: plain stores)
        return;
    }
    blk -= a.s.nb_digest;
    uint32_t* s_word = reinterpret_cast<uint32_t*>(lds);
    if (tid == 0) {
        const bool ok = poll_then_acquire(&a.sync[0], a.s.nb_digest, kFindSpinLimit);      // the digest's rows, past this CU's L1
        if (!ok) __hip_atomic_store(&a.sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_word = ok ? 1u : 0u;
    }
    __syncthreads();
    const bool go = *s_word != 0u;
    __syncthreads();                                                     // (the word's LDS is the fit role's from here on)
    const unsigned long long t1 = a.s.role_clock ? (unsigned long long)wall_clock64() : 0ull;
    if (go) {
        const uint32_t chunks = a.s.fit.chunks, nb = a.s.nb_fit;
        const FitItem it{0u, a.wcls, (uint32_t)((uint64_t)chunks * blk / nb), (uint32_t)((uint64_t)chunks * (blk + 1) / nb)};
        role_fit_item<BLOCK>(a.s.fit, a.s.fit.busy_from, it, lds);
    }
    stamp(a.s.role_clock, 4, t1);
    const uint32_t ticket = publish_and_count(&a.sync[1], false);         // (the sweep's only stores: atomicMax on the score words)
    if (tid == 0) {
        if (ticket == a.s.nb_fit - 1u) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");    // every block's scores
        *s_word = ticket;
    }
    __syncthreads();
    const bool last = *s_word == a.s.nb_fit - 1u;
    __syncthreads();
    if (!last) return;
    const bool aborted = __hip_atomic_load(&a.sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    if (a.want_map && !aborted) {
        const unsigned long long t2 = a.s.role_clock ? (unsigned long long)wall_clock64() : 0ull;
        map_one_tile<BLOCK>(a.s.finish_m, a.s.finish_h, a.wcls, lds);     // (stores the mappings into the host block)
        stamp(a.s.role_clock, 2, t2);
    }
    for (uint32_t p = tid; p < a.s.shapes_P; p += BLOCK) a.host->score[p] = a.s.fit.score[p];
    NHDFIT_DRAIN_VMEM();                                                  // every wave's stores into the host block, then the word behind them
    __syncthreads();
    if (tid == 0) {
        a.sync[0] = 0u; a.sync[1] = 0u; a.sync[2] = 0u;
        __hip_atomic_store(&a.host->flag, aborted ? kFindAborted : a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- one launch for a whole batch ---------------------------------------------------------------------
// nhdfit_find for MORE than one pod tile: the staged path costs three launches in a row behind the copy of the requests - digest,
// step, drain: each a latency chain of its own behind a launch gap - then two copies back and a stream wait: 170-205 us per call
// whatever the size (profiles/r05: 256 pods x 4 096 nodes 0.169 ms, 4 096 x 65 536 0.188), of which the device is busy for a third.
// Here the same roles run in ONE launch, tile by tile:
//   * blocks [0, nb_digest): the digests, dig_parts blocks per tile; each counts itself in its tile's word when its rows are out.
//     (The grid is padded to a multiple of eight blocks behind them so that fit block j still lands on XCD j % 8.)
//   * then the fit role's work items (step_fit.h; the host's list, widest tiles first).  A fit block waits for its tile's digests -
//     blocks of lower index: always running or done when it spins - sweeps its chunk range and takes a ticket of its tile.
//   * the block with a tile's last ticket sees the tile's scores final: it maps the tile's winners (map_one_tile), stores mappings
//     and scores into the fine-grained host block and counts the tile; the last tile's block stores the call's sequence number behind
//     everything (system scope) - the host polls that word - and leaves every counter at zero.
// Tiles finish at different times: the mapping of the early ones runs beside the sweep of the late ones.  A wait that does not end
// gives up as in k_find: the launch reports kFindAborted and the host takes the staged path.
struct FindNArgs {
    StepArgs s;                                      // of it: digest, fit (with its items), finish_m / finish_h (the mapping tail), nb_digest, nb_fit, shapes_P
    uint32_t want_map, tiles, dig_parts, nb_lead;    // nb_lead: nb_digest rounded up to a multiple of eight
    const uint8_t* tile_wcls;                        // [tiles]
    const uint32_t* tile_items;                      // [tiles] fit items per tile
    uint32_t* sync;                                  // [0] tiles done, [1] a wait gave up, [2 + t] digest blocks of tile t done, [2 + tiles + t] tickets of tile t; zero between launches
    unsigned long long* host_score;                  // [P] fine-grained host memory (finish_m.out: the mappings, likewise)
    uint32_t* host_flag;
    uint32_t seq;
};
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_findn(FindNArgs a) {
    extern __shared__ __align__(16) uint8_t lds[];
    uint32_t blk = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    if (blk < a.nb_lead) {
        if (blk >= a.s.nb_digest) return;                                    // padding
        role_digest<BLOCK>(a.s.digest, blk, lds);
        (void)publish_and_count(&a.sync[2u + blk / a.dig_parts], true);      // (rows, headers, zeroed scores: plain stores)
        return;
    }
    blk -= a.nb_lead;
    FitItem it = a.s.fit.items[blk];
    const uint32_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.tile);
    uint32_t* s_word = reinterpret_cast<uint32_t*>(lds);
    if (tid == 0) {
        const bool ok = poll_then_acquire(&a.sync[2u + tile], a.dig_parts, kFindSpinLimit);   // the digests' rows, past this CU's L1
        if (!ok) __hip_atomic_store(&a.sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_word = ok ? 1u : 0u;
    }
    __syncthreads();
    const bool go = *s_word != 0u;
    __syncthreads();                                                         // (the word's LDS is the fit role's from here on)
    if (go) role_fit_item<BLOCK>(a.s.fit, a.s.fit.busy_from, it, lds);
    const uint32_t ticket = publish_and_count(&a.sync[2u + a.tiles + tile], false);   // (the sweep's only stores: atomicMax on the score words)
    if (tid == 0) {
        const bool mine = ticket == a.tile_items[tile] - 1u;
        if (mine) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // every block's scores of this tile
        *s_word = mine ? 1u : 0u;
    }
    __syncthreads();
    const bool last = *s_word != 0u;
    __syncthreads();
    if (!last) return;
    const bool gave_up = __hip_atomic_load(&a.sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    if (a.want_map && !gave_up) map_one_tile<BLOCK>(a.s.finish_m, a.s.finish_h, a.tile_wcls[tile], lds, nullptr, tile);   // (stores the mappings into the host block)
    const uint32_t pod0 = tile * kTile;
    if (tid < (uint32_t)kTile && pod0 + tid < a.s.shapes_P) a.host_score[pod0 + tid] = a.s.fit.score[pod0 + tid];
    NHDFIT_DRAIN_VMEM();                                                      // every wave's stores into the host block
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");                         // (system scope: the host block) ... then the tile counts
        NHDFIT_DRAIN_VMEM();
        a.sync[2u + tile] = 0u; a.sync[2u + a.tiles + tile] = 0u;
        const uint32_t done = __hip_atomic_fetch_add(&a.sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        NHDFIT_DRAIN_VMEM();
        if (done == a.tiles - 1u) {                                           // every tile's results are out (each block released before it counted)
            const bool aborted = __hip_atomic_load(&a.sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            a.sync[0] = 0u; a.sync[1] = 0u;
            __hip_atomic_store(a.host_flag, aborted ? kFindAborted : a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- one launch for a LONE pod, no tables --------------------------------------------------------------
// nhdfit_find with ONE pod.  The tile image is overhead then (63 of 64 columns empty, every row a ballot, and the fit blocks
// wait for it): every block computes the pod's own masks over its 2^G assignments in LDS instead (fit_core.h LoneMasks - the
// 16-bit entries the digest would have bit-sliced; lane = signature for the reach families, read from the dictionary's 16-bit
// stream), waits for nobody, and sweeps its chunks with lane = node: the node's planes, popcounts, eight look-ups.  The block
// with the last ticket maps the winner (map_one_tile<., LONE>).  Host side as k_find.
struct Find1Args {
    MapArgs m;                                       // planes, detail, caps; reqs / out in the host block; score = the launch's score word
    ShapeArgs h;
    const nhdfit_plane4* p4;
    DictView d;                                      // caps, ncls, group_sets, flat / flat_words
    uint32_t nsig, fc_dim, fg_dim, ngs;
    uint32_t chunks, nb;
    double busy_from;
    const uint64_t* cand;
    uint32_t* sync;                                  // [1] fit tickets; zero between launches (as the score word)
    FindHost* host;
    uint32_t seq, want_map;
    unsigned long long* role_clock;
};
constexpr uint32_t kLoneMaxSigs = 4096;              // r0 / r1 in LDS: 2 x 8 KB
constexpr uint32_t kLoneGpuDim = NHDFIT_MAX_GPUS_PER_NUMA + 1, kLoneCoreDim = NHDFIT_MAX_CORES_PER_NUMA + 1;
constexpr size_t kLoneLds = lds_slice(sizeof(nhdfit_req)) + lds_slice(sizeof(PodSums)) + lds_slice(sizeof(PodHeader)) +
                            lds_slice(NHDFIT_MAX_CLASSES * (kMaxG + 1) * sizeof(uint16_t)) + 2 * lds_slice(kLoneGpuDim * sizeof(uint16_t)) +
                            2 * lds_slice(2 * kLoneCoreDim * 2 * sizeof(uint16_t)) + lds_slice(kDictLdsWords * sizeof(uint16_t)) +
                            2 * lds_slice(kLoneMaxSigs * sizeof(uint16_t)) + lds_slice(8 * sizeof(unsigned long long));
__device__ __forceinline__ void map_lone_pod_wave(const MapArgs& a, const ShapeArgs& h, const LoneMasks& t, const nhdfit_req& r,
                                                  const double* __restrict__ caps, uint8_t* lds);      // find1_wave_map.h, behind seq_kernel.h
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_find1(Find1Args a) {
    extern __shared__ __align__(16) uint8_t lds_all[];
    uint8_t* lds = lds_all;
    nhdfit_req* s_req = carve<nhdfit_req>(lds, 1);
    PodSums* s_sum = carve<PodSums>(lds, 1);
    PodHeader* s_hdr = carve<PodHeader>(lds, 1);
    uint16_t* s_cover = carve<uint16_t>(lds, NHDFIT_MAX_CLASSES * (kMaxG + 1));
    uint16_t* s_a0 = carve<uint16_t>(lds, kLoneGpuDim);
    uint16_t* s_a1 = carve<uint16_t>(lds, kLoneGpuDim);
    uint16_t* s_w0 = carve<uint16_t>(lds, 2 * kLoneCoreDim * 2);
    uint16_t* s_w1 = carve<uint16_t>(lds, 2 * kLoneCoreDim * 2);
    uint16_t* s_flat = carve<uint16_t>(lds, kDictLdsWords);
    uint16_t* s_r0 = carve<uint16_t>(lds, kLoneMaxSigs);
    uint16_t* s_r1 = carve<uint16_t>(lds, kLoneMaxSigs);
    unsigned long long* s_best = carve<unsigned long long>(lds, 8);
    uint8_t* lds_map = lds;                                              // the mapping tail's staging area
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const unsigned long long t0 = a.role_clock ? (unsigned long long)wall_clock64() : 0ull;

    // ---- the pod's masks
    if (tid < sizeof(nhdfit_req) / 16) reinterpret_cast<uint4*>(s_req)[tid] = reinterpret_cast<const uint4*>(a.m.reqs)[tid];
    for (uint32_t w = tid; w < a.d.flat_words / 2; w += BLOCK)          // (the stream is padded to an even word count)
        reinterpret_cast<uint32_t*>(s_flat)[w] = reinterpret_cast<const uint32_t*>(a.d.flat)[w];
    __syncthreads();
    const nhdfit_req& r = *s_req;
    const bool valid = req_valid(r);
    if (tid == 0) {
        *s_hdr = pod_header(r);
        s_sum->G = r.n_groups; s_sum->W = 1u << (r.n_groups & 7u); s_sum->full = s_sum->W - 1;
        s_sum->misc_smt = r.misc_smt; s_sum->misc_nosmt = r.misc_nosmt;
    }
    if (valid && tid < (1u << r.n_groups)) {                             // subset sums (pod_sums), one subset per thread
        uint32_t g = 0, x = 0, y = 0;
        for (uint32_t i = 0; i < r.n_groups; ++i)
            if (tid >> i & 1) { g += r.gpus[i]; x += r.cpu_smt[i]; y += r.cpu_nosmt[i]; }
        s_sum->gpu[tid] = g; s_sum->cpu_smt[tid] = x; s_sum->cpu_nosmt[tid] = y;
    }
    __syncthreads();
    const uint32_t W = s_sum->W, G = s_sum->G;
    if (valid && tid < a.d.ncls) class_cover(r, a.d.caps[tid], W, G, &s_cover[tid * (kMaxG + 1)]);
    for (uint32_t k = tid; k < 2 * a.fg_dim; k += BLOCK) {
        const uint32_t u = k >= a.fg_dim, f = u ? k - a.fg_dim : k;
        (u ? s_a1 : s_a0)[f] = valid ? (uint16_t)entry_a(*s_sum, u, f) : (uint16_t)0;
    }
    for (uint32_t k = tid; k < 2 * (2 * a.fc_dim * 2); k += BLOCK) {     // [u][smt * fc_dim + c][m]
        const uint32_t u = k >= 2 * a.fc_dim * 2, e = u ? k - 2 * a.fc_dim * 2 : k, m = e & 1, rec = e >> 1, smt = rec >= a.fc_dim, c = smt ? rec - a.fc_dim : rec;
        (u ? s_w1 : s_w0)[e] = valid ? (uint16_t)entry_w(*s_sum, u, smt != 0, c, m) : (uint16_t)0;
    }
    __syncthreads();
    for (uint32_t sig = tid; sig < a.nsig; sig += BLOCK) {               // lane = signature
        const uint32_t reach = valid ? sig_reach_flat(s_flat, a.nsig, sig, s_cover, W) : 0u;
        s_r0[sig] = (uint16_t)entry_r(reach, W, 0);
        s_r1[sig] = (uint16_t)entry_r(reach, W, 1);
    }
    __syncthreads();
    const LoneMasks t{s_a0, s_a1, s_w0, s_w1, s_r0, s_r1};
    const PodHeader h = *s_hdr;
    stamp(a.role_clock, 3, t0);

    // ---- lane = node
    const unsigned long long t1 = a.role_clock ? (unsigned long long)wall_clock64() : 0ull;
    constexpr uint32_t NW = BLOCK / 64;
    const uint32_t c_lo = (uint32_t)((uint64_t)a.chunks * blockIdx.x / a.nb), c_hi = (uint32_t)((uint64_t)a.chunks * (blockIdx.x + 1) / a.nb);
    const bool needs_gpu = (h.flags & kPodNeedGpu) != 0;
    uint32_t best_any = ~0u, best_pref = ~0u;
    for (uint32_t c = c_lo + wave; c < c_hi; c += NW) {
        const uint32_t i = c * 64 + lane;
        bool ok = false, nogpu = false;
        if (i < a.m.n) {
            const NodeIdx ni = node_index(a.m.p0[i], a.m.p1[i], a.m.p2[i], a.p4[i], a.fc_dim, a.fg_dim, a.ngs);
            nogpu = ni.nogpu != 0;
            ok = lone_pod_fits(t, h, ni, a.m.p3[i], a.p4[i].busy_time >= a.busy_from, a.d.group_sets);
            if (a.cand && !(a.cand[c] >> lane & 1)) ok = false;
        }
        const uint64_t word = __ballot(ok), pref = needs_gpu ? 0ull : word & __ballot(nogpu);
        if (word && best_any == ~0u) best_any = c * 64 + (uint32_t)__builtin_ctzll(word);
        if (pref && best_pref == ~0u) best_pref = c * 64 + (uint32_t)__builtin_ctzll(pref);
    }
    unsigned long long best = 0;
    if (best_pref != ~0u) best = score_of(true, a.m.global_base + best_pref);
    else if (best_any != ~0u) best = score_of(false, a.m.global_base + best_any);
    if (lane == 0) s_best[wave] = best;
    __syncthreads();
    if (tid == 0) {
        unsigned long long mx = s_best[0];
#pragma unroll
        for (int w = 1; w < (int)NW; ++w) mx = s_best[w] > mx ? s_best[w] : mx;
        if (mx) atomicMax(const_cast<unsigned long long*>(a.m.score), mx);
    }
    stamp(a.role_clock, 4, t1);
    // the block's only store is thread 0's atomicMax on the score word - performed at the coherence point: the ticket follows it on the
    // same lane, no fence; the last block reads the word with an agent-scope load (everything else it reads is older than the launch)
    if (tid == 0) {
        NHDFIT_DRAIN_VMEM();
        s_best[0] = __hip_atomic_fetch_add(&a.sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        NHDFIT_DRAIN_VMEM();
    }
    __syncthreads();
    const bool last = (uint32_t)s_best[0] == a.nb - 1u;
    __syncthreads();
    if (!last) return;
    if (a.want_map) {
        const unsigned long long t2 = a.role_clock ? (unsigned long long)wall_clock64() : 0ull;
        MapArgs m = a.m;
        m.reqs = s_req;                                                   // the request is in this block's LDS already: no second read of the host block
        // one wavefront, the lanes working together (find1_wave_map.h; round 5: -3 us per call against the tile machinery with one
        // live lane, profiles/r05/candidates.md); stores the mapping into the host block
        map_lone_pod_wave(m, a.h, t, *s_req, a.d.caps, lds_map);
        stamp(a.role_clock, 2, t2);
    }
    if (tid == 0) a.host->score[0] = __hip_atomic_load(a.m.score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    NHDFIT_DRAIN_VMEM();                                                  // every wave's stores into the host block, then the word behind them
    __syncthreads();
    if (tid == 0) {
        a.sync[1] = 0u;
        *const_cast<unsigned long long*>(a.m.score) = 0ull;
        __hip_atomic_store(&a.host->flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
