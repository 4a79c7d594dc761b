// nhdfit.hip - gfx950 kernels and the C-ABI of libnhdfit.so (include/nhdfit.h).
//
// ONE kernel launch per step (k_step).  Its grid is the union of five block ranges ("roles"), each working on a
// different step of a software pipeline over eight request-side buffer sets:
//   digest (step i+1)  per 64-pod tile: request records -> bit-sliced table image (CPU/GPU/NIC feasibility of every
//                      NUMA assignment as a function of a node's free-resource counts / NIC signature)
//   fit    (step i)    the P x N pass.  Block = (pod tile, node range).  The tile's table image is staged in LDS; a
//                      wavefront owns 64 consecutive nodes (lane = node, coalesced 16 B/lane loads of the five SoA
//                      planes, __popcll of the free-core bitmaps), ANDs the table rows of its node for all 64 pods
//                      at once, transposes the 64 x 64 verdict bits so that lane j holds pod j's 64-node word:
//                      coalesced bitmap store, first-fit score via ctz, max-reduced per block, one atomicMax per pod
//   shapes (step i-1), choose (step i-2), finish (step i-3)
//                      the winners' resource mappings (CPython set-order model): per-pod shape, the sequential
//                      set model once per distinct shape of a tile, per-pod NIC choice
// The side roles are long on latency and short on work; inside the fit role's launch they cost no stream time and
// no extra launches (the host does one launch per step).  Multi-GPU: ncclAllReduce(score, P, ncclUint64, ncclMax)
// on a second stream between the fit of step i and the shapes role of step i (one launch later).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "seq_core.h"
#include "wide_core.h"
#include "dict_stream.h"

using namespace nhdfit;

// Tuning / profiling knobs read from the environment exist in the tuning build only (-DNHDFIT_TUNING,
// `python -m nhd_amd.build --tuning` -> libnhdfit_tuning.so, used by tools/): the shipped library looks nothing up.
#ifdef NHDFIT_TUNING
static inline const char* tune_env(const char* name) { return getenv(name); }
#else
static inline const char* tune_env(const char*) { return nullptr; }
#endif

namespace {

// ------------------------------------------------------------------------------------------------
// device code
// ------------------------------------------------------------------------------------------------
#include "step_digest.h"
#include "step_fit.h"
#include "step_map.h"
#include "step_kernel.h"
#include "seq_kernel.h"
#include "find1_wave_map.h"
#include "seq2_kernel.h"
#include "wide_kernel.h"
#include "big_kernel.h"

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
thread_local std::string g_create_error;

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string& err) {
        if (handle) return true;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
        GetUniqueId = (decltype(GetUniqueId))dlsym(handle, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(handle, "ncclCommInitRank");
        AllReduce = (decltype(AllReduce))dlsym(handle, "ncclAllReduce");
        CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
        CommInitAll = (decltype(CommInitAll))dlsym(handle, "ncclCommInitAll");
        GroupStart = (decltype(GroupStart))dlsym(handle, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(handle, "ncclGroupEnd");
        Send = (decltype(Send))dlsym(handle, "ncclSend");
        Recv = (decltype(Recv))dlsym(handle, "ncclRecv");
        GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !AllReduce || !CommDestroy || !GetErrorString || !CommInitAll || !GroupStart || !GroupEnd || !Send || !Recv) {
            err = "librccl lacks a required symbol";
            return false;
        }
        return true;
    }
};
Rccl g_rccl;

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;   // elements
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc((void**)&p, n * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

template <class T>
struct PinBuf {                   // page-locked host staging: async copies in both directions, no bounce buffer inside the runtime
    T* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipHostMalloc((void**)&p, n * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

constexpr int kEventRing = 256;
constexpr int kBufs = 8;          // buffer sets: step s owns set s % kBufs from its digest (one launch before its fit)
                                  // to the end of its mapping (four launches after it, five when sharded)

constexpr int kPipes = 3;           // pipelines a context owns; a staged batch deals its steps to two of them, or to all three (nhdfit_enqueue_step)
struct Pipe {                            // one software pipeline of steps: its stream, its buffer sets, how far each phase got
    hipStream_t stream = nullptr;
    hipEvent_t ev_fit[kBufs] = {}, ev_red[kBufs] = {};   // stream <-> s_red hand-over (sharded runs only)
    hipEvent_t ev_staged = nullptr;      // pipe 0 records it behind what it staged (requests, work items, node records); pipe 1 waits for it
    uint64_t n_dig = 0, n_fit = 0, n_shaped = 0, n_chosen = 0, n_finished = 0;   // steps (since the last stage_requests) whose phase was launched
    uint64_t seen_gen = 0;               // the staging generation of pipe 0's stream this pipe has waited for (nhdfit_ctx::staged_gen)
    DevBuf<PodHeader> hdr[kBufs]; DevBuf<uint8_t> tabs[kBufs];
    DevBuf<unsigned long long> score[kBufs]; DevBuf<nhdfit_mapping> maps[kBufs];
    DevBuf<uint64_t> nm;                 // node-major verdict words [tiles][chunks*64] (one buffer per pipe: its fit roles run in stream order)
    DevBuf<unsigned long long> shape_keys[kBufs]; DevBuf<uint32_t> shape_res[kBufs]; DevBuf<int32_t> shape_slot[kBufs];   // mapping dedup tables
    DevBuf<uint32_t> shape_list[kBufs];  // distinct shapes per tile
    DevBuf<uint32_t> dig_count;          // [tiles] arrival counters of the digest blocks that share a tile's signature rows (zero between launches)
};

}  // namespace

struct nhdfit_ctx {
    int dev = -1;
    // Several pipes: the pipelined form (stage, enqueue, enqueue, ...) deals its steps round robin to independent software
    // pipelines on their own streams, so that one step's launch gap, table staging and tail are covered by the other steps'
    // blocks (two against one: 23 -> 17 us per step, profiles/r03; a third where the digest is a long chain, profiles/r04).
    // Everything else - single finds, mode B, uploads, deltas - runs on pipe 0, whose stream is `stream`; whatever changes the
    // mirror waits for all of them (sync_all).
    Pipe pipe[kPipes];
    hipStream_t stream = nullptr;        // = pipe[0].stream: uploads, deltas, commits, mode B, single finds
    hipStream_t s_red = nullptr;         // the all-reduce of sharded runs, overlapping the next step launch
    bool side_streams_used = true;       // something was enqueued on a pipe other than the first, or on s_red, since sync_all last waited for them
    double enq_us = 0, enq_launch_us = 0, enq_events_us = 0; uint64_t enq_n = 0;   // tuning aid (NHDFIT_ENQ_PROF): host time of nhdfit_enqueue_step, of the launch calls, of the event records
    bool known_idle = false;             // no HIP call of this context since its streams were last seen idle (sync_all; a single-launch find's polled word)
    bool dual = tune_env("NHDFIT_ONE_PIPE") == nullptr;   // tuning aid: NHDFIT_ONE_PIPE=1 keeps every step on pipe 0
    uint64_t n_enq = 0;                  // steps enqueued since the last stage_requests (step k runs on pipe k % 2)
    int last_pipe = 0;                   // the pipe of the most recent step (nhdfit_fetch reads its results)
    uint64_t staged_gen = 1;             // bumped whenever pipe 0's stream gets staging work (requests, work items, node records): the other pipes wait for it once
    int npipes = 2;                      // pipes the staged batch's steps are dealt to: 2, or 3 (nhdfit_enqueue_step decides per batch)
    bool geom_big = true;                // 512-thread step blocks (256 for small problems)
    uint32_t digest_parts = tune_env("NHDFIT_DIGEST_PARTS") ? (uint32_t)atoi(tune_env("NHDFIT_DIGEST_PARTS")) : 2;   // tuning aid
    uint32_t side_prio = tune_env("NHDFIT_SIDE_PRIO") ? (uint32_t)atoi(tune_env("NHDFIT_SIDE_PRIO")) : 1;   // tuning aid
    int seq_pods = tune_env("NHDFIT_SEQ_PODS") && atoi(tune_env("NHDFIT_SEQ_PODS")) == 8 ? 8 : 16;   // tuning aid: pods per round of the sequential kernel
    uint32_t choose_split = tune_env("NHDFIT_CHOOSE_SPLIT") ? (uint32_t)atoi(tune_env("NHDFIT_CHOOSE_SPLIT")) : 16;   // tuning aid: wavefronts per tile
    bool split = tune_env("NHDFIT_SPLIT") != nullptr;
    bool role_kernels = tune_env("NHDFIT_ROLE_KERNELS") != nullptr;   // profiling aid: every role as a kernel of its own (512-thread geometry only)
    DevBuf<unsigned long long> role_clock;            // profiling aid: NHDFIT_ROLE_TIMES=<step> prints the role windows of that step
    int64_t role_step = tune_env("NHDFIT_ROLE_TIMES") ? atoll(tune_env("NHDFIT_ROLE_TIMES")) : -1;   // profiling aid: launch the side roles apart from the fit role
    std::string err;
    hipDeviceProp_t prop;

    // node mirror
    DevBuf<nhdfit_plane0> p0; DevBuf<nhdfit_plane1> p1; DevBuf<nhdfit_plane2> p2;
    DevBuf<nhdfit_plane3> p3; DevBuf<nhdfit_plane4> p4; DevBuf<nhdfit_detail> det;
    DevBuf<nhdfit_origin> origin; uint32_t origin_hi = 0;   // nhdfit_upload_origin: records [0, origin_hi) are there
    DevBuf<nhdfit_delta> deltas; DevBuf<uint32_t> delta_run; DevBuf<uint8_t> delta_status;
    uint32_t n = 0, capacity = 0;
    uint64_t global_base = 0;

    // dictionary
    DevBuf<double> caps; DevBuf<uint32_t> sig_off, pool_off; DevBuf<uint8_t> pool_glimit; DevBuf<nhdfit_cc> cc;
    DevBuf<uint16_t> sig_flat; uint32_t flat_words = 0;
    DevBuf<uint16_t> sig_flat2; uint32_t flat2_words = 0;   // the same by pool type (DictView::flat2)
    uint32_t ncls = 0, nsig = 0;
    uint32_t max_cores = 1, max_gpus = 0, ngs = 0;
    DevBuf<uint64_t> group_sets;
    uint32_t lds_bytes = 0;       // largest hot section among the staged tiles (what a fit block stages in LDS)
    bool x_spill = false;
    uint32_t hot_staged[kWClasses] = {0, 0, 0, 0};   // per row width: bytes of the hot section staged in LDS (== hot_bytes unless the X rows outgrow LDS)
    Layout L[kWClasses] = {};     // tile-image layout per row width (dictionary, staged batch, provisioned X rows)
    uint32_t pitch = 0;           // bytes between tile images
    uint32_t hp_rows = 2;
    uint32_t n_big_pods = 0;      // staged pods with more than 3 proc groups
    uint32_t max_wcls = 0;        // widest tile class of the staged batch
    // pair form of the fit role's sweep (fit_core.h "pair rows"): per row width W = 2 / 4 the largest CPU demand among the staged
    // tiles of that width, and what refresh_layouts makes of it - table dimension D (0: off)
    uint32_t max_demand[2] = {0, 0};
    uint32_t pair_D[2] = {0, 0};
    uint32_t crow_D[2] = {0, 0};          // pair-table dimensions the records' C rows (NodeRec::flags) are written for; crow_stale: a staged
    bool crow_stale = false;              // batch of more than a tile asks for others - k_xcrow rewrites them in front of its first step
    bool pair_rows = !(tune_env("NHDFIT_PAIR") && atoi(tune_env("NHDFIT_PAIR")) == 0);   // tuning aid: NHDFIT_PAIR=0 keeps the six-fetch sweep

    // requests / results
    DevBuf<nhdfit_req> reqs; uint32_t P = 0;
    std::vector<uint32_t> perm;          // device (class-sorted) position -> caller's pod index
    PinBuf<nhdfit_req> pin_reqs; PinBuf<uint8_t> pin_wcls; PinBuf<uint64_t> pin_score; PinBuf<nhdfit_mapping> pin_maps;   // host staging of one call
    PinBuf<uint8_t> pin_items;           // the fit role's work items on their way to the device
    DevBuf<uint64_t> bitmap;             // pod-major rows [chunks][P], converted from `nm` on demand (fetch)
    DevBuf<uint64_t> rows_t;             // ... [P][chunks] for the sequential kernels (mode B)
    DevBuf<uint64_t> cand;               // [chunks] candidate nodes of the call
    DevBuf<uint8_t> tile_wcls;           // row width class per staged tile
    DevBuf<FitItem> items; uint32_t n_items = 0;   // work items of the fit role (blocks), heaviest tiles first
    std::vector<uint8_t> h_tile_wcls;
    DevBuf<AscEntry> asc;                // layouts of ascending-filled CPython sets (static table, built at creation)
    DevBuf<uint8_t> choose_tab;          // choose_tuples tabulated for U = 2, G <= 2 (static table, built at creation)
    // node records (fit_core.h NodeRec) + the class table behind their X rows; [rec_lo, rec_hi) = nodes whose records
    // are stale (uploads, commits), rec_all = every record (dictionary / capacity / node count changed)
    DevBuf<NodeRec> rec[kWClasses];
    DevBuf<double> rec_bt[kWClasses];    // busy times beside the records, in the records' (lane) order
    // lane order of the chunks' records (step_kernel.h k_xorder): built for the pair-table dimensions order_D; ord_all = every chunk is
    // (re)dealt at the next step (a staged batch changed a dimension); chunks k_xrecords rewrites are dealt right behind it
    bool lane_order = !(tune_env("NHDFIT_LANE_ORDER") && atoi(tune_env("NHDFIT_LANE_ORDER")) == 0);   // tuning aid: NHDFIT_LANE_ORDER=0 keeps node order
    uint32_t order_D[2] = {~0u, ~0u}; bool ord_all = false;
    DevBuf<unsigned long long> xkeys; DevBuf<uint32_t> xids; DevBuf<uint64_t> xcls; DevBuf<uint32_t> xnx;
    uint32_t rec_lo = 0, rec_hi = 0; bool rec_all = true;
    uint32_t nx = 0, x_cap = kMinXCap;   // interned classes (as of the last record update) / provisioned X rows
    // signatures some interned class refers to, ascending (DigestArgs::sig_list): rebuilt when classes are added; all_sigs: the
    // digest of the step being enqueued forms every signature's rows (mode B's snapshot pass reads them for committed nodes)
    DevBuf<uint16_t> sig_use; uint32_t n_sig_use = 0, sig_use_nx = 0; bool all_sigs = false;
    bool sig_use_on = tune_env("NHDFIT_ALL_SIGS") == nullptr;   // tuning aid: NHDFIT_ALL_SIGS=1 digests every signature in every step
    uint32_t fit_blocks = tune_env("NHDFIT_FIT_BLOCKS") ? (uint32_t)atoi(tune_env("NHDFIT_FIT_BLOCKS")) : 0;   // tuning aid: blocks of the fit role
    bool use_choose_tab = tune_env("NHDFIT_NO_CHOOSE_TABLE") == nullptr;   // tuning aid: run the set model for every shape
    // set-layout state machine for three-group pods (set_states.h): verified against the model on the host
    // (tests/test_pyset_emulation.py), parity-green and 10 % faster per step on the GPU (profiles/r02):
    // on by default, NHDFIT_NO_SET_STATES=1 runs the insertion-by-insertion model instead
    DevBuf<uint64_t> st_info; DevBuf<uint32_t> st_next, st_asc; uint32_t st_n = 0;
    bool use_set_states = tune_env("NHDFIT_NO_SET_STATES") == nullptr;
    // mode B
    DevBuf<uint64_t> nogpu, taken, tile_masks; DevBuf<int32_t> touched; DevBuf<uint16_t> gl_tiles; std::vector<SeqResult> seq_host; DevBuf<UndoRec> undo; DevBuf<SeqResult> seq_out; DevBuf<nhdfit_placement> seq_place;
    DevBuf<uint32_t> order, seq_counters;
    DevBuf<unsigned long long> seq_queue; DevBuf<uint32_t> seq_ctrl, seq_mat, seq_flags; DevBuf<uint4> seq_ent;   // decision-engine form of mode B (seq2_kernel.h)
    bool seq_general = tune_env("NHDFIT_SEQ_GENERAL") != nullptr;   // tuning aid: the one-block kernel for every batch
    DevBuf<uint64_t> sig_keys; DevBuf<uint32_t> sig_ids; uint32_t sig_mask = 0;   // canonical NIC-state key -> signature id (commit_core.h)
    bool use_cand = false, want_bitmap = true, want_map = true;
    // single-launch find (k_find, step_kernel.h): the fine-grained host block the launch reads the requests from and stores
    // its results into, the launch's counters, the sequence number of the last call, and what the device's candidate mask holds
    FindHost* find_host = nullptr;
    CommitHost* commit_host = nullptr; uint32_t commit_seq = 0;      // nhdfit_commit's result block (fine-grained host memory, polled)
    DevBuf<uint32_t> find_sync;
    DevBuf<unsigned long long> find_red;   // sharded single-launch find: the tile's scores on their way through the all-reduce
    uint32_t find_seq = 0;
    std::vector<uint64_t> cand_shadow;   // copy of the mask a small find last uploaded to `cand` (empty: unknown)
    bool fast_find = tune_env("NHDFIT_NO_FAST_FIND") == nullptr;   // tuning aid: every find through the staged five-launch path
    // single-launch find of a whole batch (k_findn): its host block (flag | scores | mappings, grown with the largest call), its counters
    uint8_t* findn_host = nullptr; size_t findn_cap = 0;
    DevBuf<uint32_t> findn_sync; uint32_t findn_sync_words = 0;
    bool batch_find = tune_env("NHDFIT_NO_BATCH_FIND") == nullptr;   // tuning aid: batches of more than one tile through the staged path
    PinBuf<uint32_t> pin_order; uint32_t tn_n = 0;   // mode B: [order | list] on their way to the device; pods without GPUs in the list
    bool reqs_deferred = false, wcls_deferred = false;   // the staged batch's request records / tile classes are in the page-locked block only (finish_deferred_copies)
    bool lone_pod = tune_env("NHDFIT_NO_LONE_POD") == nullptr;      // one pod: the table-free launch (k_find1); tuning aid: NHDFIT_NO_LONE_POD=1 takes k_find

    // nodes beyond the fast layout (wide_core.h): records sorted by index; the device copy is the truth once commits ran on it
    DevBuf<nhdfit_wide_node> wide; uint32_t n_wide = 0;
    DevBuf<nhdfit_wide_share> wide_share; bool sharing = false;   // ENABLE_SHARING arithmetic: one record beside every wide record (nhdfit_wide_share_upload)
    DevBuf<int16_t> wide_scratch; DevBuf<uint32_t> wide_flags; DevBuf<nhdfit_wide_placement> wide_place;
    std::vector<nhdfit_wide_placement> wide_places_last;   // placements the last nhdfit_schedule_batch made on wide nodes
    std::vector<uint32_t> wide_index;                      // host copy of the records' node indices (ascending)
    // big requests (5..8 processing groups, big_kernel.h): buffers of nhdfit_big_find / nhdfit_big_commit
    DevBuf<nhdfit_big_req> big_reqs; DevBuf<unsigned long long> big_score; DevBuf<nhdfit_big_mapping> big_maps;
    DevBuf<uint32_t> big_flags; DevBuf<nhdfit_big_placement> big_place; DevBuf<uint64_t> big_cand; DevBuf<int32_t> big_scratch;
    uint32_t wide_max_numa = 0;                            // most sockets among the wide records (sizes a big request's set tables)
    int wide_slot(uint32_t node) const {
        auto it = std::lower_bound(wide_index.begin(), wide_index.end(), node);
        return it != wide_index.end() && *it == node ? (int)(it - wide_index.begin()) : -1;
    }

    // timing
    hipEvent_t ev[kEventRing][2];        // start / end of sampled step launches
    uint8_t ev_kind[kEventRing] = {};    // 0 = step launch with a fit role, 1 = digest-only launch
    int ev_pending = 0;
    nhdfit_stats stats;

    // collective
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    bool warned_general_loop = false;
    DevBuf<uint8_t> xfer_send, xfer_recv;   // nhdfit_comm_sendrecv / nhdfit_comm_allreduce_sum_u8: device staging of the host buffers
};

namespace {

int fail(nhdfit_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(c, expr)                                                                         \
    do {                                                                                        \
        (c)->known_idle = false;          /* (whatever it is, it may put work on a stream) */   \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail((c), NHDFIT_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// Everything that puts work on a stream clears known_idle in one place: the HIP calls through HIPCHK above, the kernel launches through
// LAUNCH, the collectives through the comma form at their call sites (`(c->known_idle = false, g_rccl).AllReduce(...)`) - the
// idle shortcut of stage_requests must never see a stale `true`.
#define LAUNCH(c, ...)                                   \
    do {                                                 \
        (c)->known_idle = false;                         \
        hipLaunchKernelGGL(__VA_ARGS__);                 \
    } while (0)

// Waiting for a stream: the runtime's own wait parks the thread on the queue's interrupt, and the wake-up costs tens of
// microseconds - as much as a whole step of the pipelined form, half of what a 20-step region loses at its end, a third of a
// batch call through host buffers.  The scheduler's thread has nothing else to do while its one call is in flight (the reference
// calls FindNode from one thread, nhd/NHDScheduler.py:43,277), so it polls the stream first - for at most kSpinWaitUs, the length of
// the longest ordinary call (a mode-B batch) - and only then goes to sleep on it.
constexpr long kSpinWaitUs = 20000;
hipError_t wait_stream(hipStream_t s) {
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t polls = 0;; ++polls) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) {
            if (polls && e == hipSuccess) (void)hipGetLastError();      // ("not ready" must not be what the next launch's error check finds)
            return e;
        }
        if ((polls & 63u) == 63u && std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > kSpinWaitUs) break;
    }
    (void)hipGetLastError();
    return hipStreamSynchronize(s);
}

int drain_events(nhdfit_ctx* c) {
    for (int k = 0; k < c->ev_pending; ++k) {
        float f = 0;
        HIPCHK(c, hipEventSynchronize(c->ev[k][1]));
        HIPCHK(c, hipEventElapsedTime(&f, c->ev[k][0], c->ev[k][1]));
        if (c->ev_kind[k]) { c->stats.digest_ms_last = f; continue; }
        c->stats.launches++;
        c->stats.fit_ms_total += f;
        c->stats.fit_ms_last = f;
        c->stats.step_ms_last = f;
    }
    c->ev_pending = 0;
    return NHDFIT_OK;
}

constexpr size_t kLdsPerCu = 160 * 1024;      // gfx950
int flush_pipeline(nhdfit_ctx* c);
int refresh_layouts(nhdfit_ctx* c);

int sync_all(nhdfit_ctx* c) {
    { int rc_ = flush_pipeline(c); if (rc_) return rc_; }      // pending mapping phases of the last steps
    // the other pipes' streams and the reduce stream only ever carry the steps of a staged batch (launch_step, flush_pipeline) and calls
    // that wait for their own work: nothing was put on them since the last wait here -> nothing to ask (a query of an idle stream is ~3 us,
    // four of them were a tenth of a small nhdfit_find)
    for (Pipe& p : c->pipe)
        if (&p == &c->pipe[0] || c->side_streams_used) HIPCHK(c, wait_stream(p.stream));
    if (c->side_streams_used) HIPCHK(c, wait_stream(c->s_red));
    c->side_streams_used = false;
    c->known_idle = true;
    return NHDFIT_OK;
}

}  // namespace

extern "C" {

int nhdfit_abi_version(void) { return NHDFIT_ABI_VERSION; }

int nhdfit_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* nhdfit_last_error(nhdfit_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

void nhdfit_destroy(nhdfit_ctx* c);

int nhdfit_create(int device_id, nhdfit_ctx** out) {
    if (!out) return fail(nullptr, NHDFIT_E_INVAL, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, NHDFIT_E_NODEVICE, "no HIP device available (%s); libnhdfit has no CPU path",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, NHDFIT_E_INVAL, "device %d out of range [0,%d)", device_id, ndev);
    nhdfit_ctx* c = new (std::nothrow) nhdfit_ctx();
    if (!c) return fail(nullptr, NHDFIT_E_NOMEM, "out of host memory");
    c->dev = device_id;
    memset(&c->stats, 0, sizeof c->stats);
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipGetDeviceProperties(&c->prop, device_id)) != hipSuccess) {
        int rc = fail(nullptr, NHDFIT_E_HIP, "cannot open device %d: %s", device_id, hipGetErrorString(e));
        delete c;
        return rc;
    }
    if (strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
        int rc = fail(nullptr, NHDFIT_E_NODEVICE, "device %d is %s; libnhdfit is built for gfx950 only", device_id, c->prop.gcnArchName);
        delete c;
        return rc;
    }
    e = hipStreamCreateWithFlags(&c->s_red, hipStreamNonBlocking);
    for (Pipe& p : c->pipe) {
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&p.ev_staged, hipEventDisableTiming);
        for (int b = 0; b < kBufs && e == hipSuccess; ++b) {
            e = hipEventCreateWithFlags(&p.ev_fit[b], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&p.ev_red[b], hipEventDisableTiming);
        }
    }
    c->stream = c->pipe[0].stream;
    for (auto& q : c->ev)
        for (auto& x : q)
            if (e == hipSuccess) e = hipEventCreate(&x);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->find_host, sizeof(FindHost), hipHostMallocCoherent);
    if (e == hipSuccess) memset(c->find_host, 0, sizeof(FindHost));
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->commit_host, sizeof(CommitHost), hipHostMallocCoherent);
    if (e == hipSuccess) memset(c->commit_host, 0, sizeof(CommitHost));
    if (e == hipSuccess) e = c->find_sync.reserve(8);                  // [0..2] counters of k_find / k_find1, [4..5] k_find1's 64-bit score word
    if (e == hipSuccess) e = hipMemsetAsync(c->find_sync.p, 0, 8 * sizeof(uint32_t), c->stream);
    if (e == hipSuccess) e = c->xkeys.reserve(kXSlots);
    if (e == hipSuccess) e = c->xids.reserve(kXSlots);
    if (e == hipSuccess) e = c->xcls.reserve(kXSlots / 2);
    if (e == hipSuccess) e = c->xnx.reserve(2);
    if (e == hipSuccess) e = hipMemsetAsync(c->xkeys.p, 0, kXSlots * sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->xids.p, 0xFF, kXSlots * sizeof(uint32_t), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->xnx.p, 0, 2 * sizeof(uint32_t), c->stream);
    if (e == hipSuccess) e = c->asc.reserve(kAscEntries);
    if (e == hipSuccess) {
        LAUNCH(c, k_build_asc, dim3((kAscEntries + 255) / 256), dim3(256), 0, c->stream, c->asc.p);
        e = hipGetLastError();
        if (e == hipSuccess && c->use_set_states) {
            std::vector<uint64_t> info;
            std::vector<uint32_t> next, asc;
            build_set_states(info, next, asc);
            c->st_n = (uint32_t)info.size();
            e = c->st_info.reserve(info.size());
            if (e == hipSuccess) e = c->st_next.reserve(next.size());
            if (e == hipSuccess) e = c->st_asc.reserve(asc.size());
            if (e == hipSuccess) e = hipMemcpy(c->st_info.p, info.data(), info.size() * sizeof(uint64_t), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(c->st_next.p, next.data(), next.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(c->st_asc.p, asc.data(), asc.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        }
        if (e == hipSuccess) e = c->choose_tab.reserve(kChooseEntries);
        if (e == hipSuccess) {
            LAUNCH(c, k_build_choose, dim3((kChooseEntries + 255) / 256), dim3(256), 0, c->stream, c->asc.p, c->choose_tab.p);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = wait_stream(c->stream);
    }
    if (e != hipSuccess) {
        int rc = fail(nullptr, NHDFIT_E_HIP, "stream / event / table creation: %s", hipGetErrorString(e));
        nhdfit_destroy(c);
        return rc;
    }
    *out = c;
    return NHDFIT_OK;
}

void nhdfit_destroy(nhdfit_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->dev);
    (void)hipDeviceSynchronize();
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    c->p0.release(); c->p1.release(); c->p2.release(); c->p3.release(); c->p4.release(); c->det.release();
    c->origin.release(); c->deltas.release(); c->delta_run.release(); c->delta_status.release();
    c->wide.release(); c->wide_share.release(); c->wide_scratch.release(); c->wide_flags.release(); c->wide_place.release();
    c->big_reqs.release(); c->big_score.release(); c->big_maps.release(); c->big_flags.release(); c->big_place.release();
    c->big_cand.release(); c->big_scratch.release();
    if (c->find_host) (void)hipHostFree(c->find_host);
    if (c->commit_host) (void)hipHostFree(c->commit_host);
    c->commit_host = nullptr;
    if (c->findn_host) (void)hipHostFree(c->findn_host);
    c->findn_host = nullptr; c->findn_cap = 0; c->findn_sync.release();
    c->find_host = nullptr; c->find_sync.release(); c->find_red.release();
    c->pin_order.release(); c->pin_reqs.release(); c->pin_wcls.release(); c->pin_score.release(); c->pin_maps.release(); c->pin_items.release();
    c->caps.release(); c->sig_off.release(); c->pool_off.release(); c->pool_glimit.release(); c->cc.release(); c->sig_flat.release(); c->sig_flat2.release();
    c->reqs.release(); c->bitmap.release(); c->rows_t.release(); c->cand.release(); c->tile_wcls.release(); c->items.release(); c->xkeys.release(); c->xids.release(); c->xcls.release(); c->xnx.release(); c->sig_use.release(); for (auto& r : c->rec) r.release(); c->role_clock.release(); c->asc.release(); c->choose_tab.release(); c->st_info.release(); c->st_next.release(); c->st_asc.release(); c->group_sets.release();
    c->nogpu.release(); c->taken.release(); c->tile_masks.release(); c->touched.release(); c->gl_tiles.release(); c->seq_counters.release(); c->undo.release(); c->seq_out.release(); c->seq_place.release(); c->order.release(); c->seq_queue.release(); c->seq_ctrl.release(); c->seq_mat.release(); c->seq_flags.release(); c->seq_ent.release(); c->sig_keys.release(); c->sig_ids.release();
    for (Pipe& p : c->pipe) {
        p.nm.release(); p.dig_count.release();
        for (int b = 0; b < kBufs; ++b) {
            p.hdr[b].release(); p.tabs[b].release(); p.score[b].release(); p.maps[b].release();
            p.shape_keys[b].release(); p.shape_res[b].release(); p.shape_slot[b].release(); p.shape_list[b].release();
            if (p.ev_fit[b]) (void)hipEventDestroy(p.ev_fit[b]);
            if (p.ev_red[b]) (void)hipEventDestroy(p.ev_red[b]);
        }
        if (p.ev_staged) (void)hipEventDestroy(p.ev_staged);
        if (p.stream) (void)hipStreamDestroy(p.stream);
    }
    for (auto& q : c->ev)
        for (auto& x : q)
            if (x) (void)hipEventDestroy(x);
    if (c->s_red) (void)hipStreamDestroy(c->s_red);
    delete c;
}

int nhdfit_set_dictionary(nhdfit_ctx* c, uint32_t max_cores_per_numa, uint32_t max_gpus_per_numa,
                          const uint64_t* group_sets, uint32_t n_group_sets,
                          const double* caps, uint32_t ncls,
                          const uint32_t* sig_off, uint32_t nsig,
                          const uint32_t* pool_off, const uint8_t* pool_glimit, uint32_t npools,
                          const nhdfit_cc* cc, uint32_t ncc) {
    if (!c) return NHDFIT_E_INVAL;
    if (!sig_off || !pool_off || nsig < 1) return fail(c, NHDFIT_E_INVAL, "dictionary needs at least the empty signature");
    if (ncls > NHDFIT_MAX_CLASSES) return fail(c, NHDFIT_E_LIMIT, "%u capacity classes (max %d)", ncls, NHDFIT_MAX_CLASSES);
    if (sig_off[0] != 0 || sig_off[1] != 0) return fail(c, NHDFIT_E_INVAL, "signature 0 must be empty");
    if (sig_off[nsig] != npools || pool_off[npools] != ncc) return fail(c, NHDFIT_E_INVAL, "inconsistent dictionary offsets");
    for (uint32_t k = 0; k < ncc; ++k)
        if (cc[k].cls >= ncls) return fail(c, NHDFIT_E_INVAL, "class id %u out of range", cc[k].cls);
    if (max_gpus_per_numa > NHDFIT_MAX_GPUS_PER_NUMA)
        return fail(c, NHDFIT_E_LIMIT, "%u GPUs per NUMA node (max %d)", max_gpus_per_numa, NHDFIT_MAX_GPUS_PER_NUMA);
    if (max_cores_per_numa < 1 || max_cores_per_numa > NHDFIT_MAX_CORES_PER_NUMA)
        return fail(c, NHDFIT_E_LIMIT, "%u cores per socket (supported: 1..%d)", max_cores_per_numa, NHDFIT_MAX_CORES_PER_NUMA);
    if (n_group_sets && !group_sets) return fail(c, NHDFIT_E_INVAL, "NULL group set table");
    if (nsig > 0xFFFF) return fail(c, NHDFIT_E_LIMIT, "%u NIC signatures (max 65535)", nsig);
    HIPCHK(c, hipSetDevice(c->dev));
    { int rc_ = sync_all(c); if (rc_) return rc_; }
    c->rec_all = true;                                  // node records depend on the mirror and on the table layout
    HIPCHK(c, c->group_sets.reserve(n_group_sets ? n_group_sets : 1));
    if (n_group_sets) HIPCHK(c, hipMemcpy(c->group_sets.p, group_sets, n_group_sets * sizeof(uint64_t), hipMemcpyHostToDevice));
    else { const uint64_t zero = 0; HIPCHK(c, hipMemcpy(c->group_sets.p, &zero, sizeof zero, hipMemcpyHostToDevice)); }
    HIPCHK(c, c->caps.reserve(NHDFIT_MAX_CLASSES));           // the mapping roles stage all 16 entries
    HIPCHK(c, c->sig_off.reserve(nsig + 1));
    HIPCHK(c, c->pool_off.reserve(npools + 1));
    HIPCHK(c, c->pool_glimit.reserve(npools ? npools : 1));
    HIPCHK(c, c->cc.reserve(ncc ? ncc : 1));
    if (ncls) HIPCHK(c, hipMemcpy(c->caps.p, caps, ncls * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->sig_off.p, sig_off, (nsig + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->pool_off.p, pool_off, (npools + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (npools) HIPCHK(c, hipMemcpy(c->pool_glimit.p, pool_glimit, npools, hipMemcpyHostToDevice));
    if (ncc) HIPCHK(c, hipMemcpy(c->cc.p, cc, ncc * sizeof(nhdfit_cc), hipMemcpyHostToDevice));
    {   // canonical key of every signature's pool set -> id, for the device-side commit (commit_core.h sig_keys_of)
        uint32_t slots = 64;
        while (slots < 4 * nsig) slots <<= 1;
        std::vector<uint64_t> keys(slots, 0);
        std::vector<uint32_t> ids(slots, 0);
        for (uint32_t sg = 1; sg < nsig; ++sg) {
            uint64_t key = 0;
            for (uint32_t pl = sig_off[sg]; pl < sig_off[sg + 1]; ++pl) {
                uint8_t cnt[NHDFIT_MAX_CLASSES] = {0};
                for (uint32_t k = pool_off[pl]; k < pool_off[pl + 1]; ++k) cnt[cc[k].cls & 15u] = cc[k].cnt;
                key = sig_key_add(key, pool_key(pool_glimit[pl], cnt));
            }
            if (key == 0) continue;                                   // a signature made of no pool is the empty one
            uint32_t sl = (uint32_t)mix64(key) & (slots - 1);
            while (keys[sl] != 0 && keys[sl] != key) sl = (sl + 1) & (slots - 1);
            if (keys[sl] == key && ids[sl] != sg)
                return fail(c, NHDFIT_E_INVAL, "signatures %u and %u describe the same NIC pools", ids[sl], sg);
            keys[sl] = key; ids[sl] = sg;
        }
        HIPCHK(c, c->sig_keys.reserve(slots));
        HIPCHK(c, c->sig_ids.reserve(slots));
        HIPCHK(c, hipMemcpy(c->sig_keys.p, keys.data(), slots * sizeof(uint64_t), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->sig_ids.p, ids.data(), slots * sizeof(uint32_t), hipMemcpyHostToDevice));
        c->sig_mask = slots - 1;
    }
    {   // the dictionary as one stream of 16-bit words (DictView::flat)
        std::vector<uint16_t> flat(nsig + 1, 0);
        bool fits = true;
        for (uint32_t sg = 0; sg < nsig && fits; ++sg) {
            const size_t at = flat.size() - (nsig + 1);
            if (at > 0xFFFFu) { fits = false; break; }
            flat[sg] = (uint16_t)at;
            flat.push_back((uint16_t)(sig_off[sg + 1] - sig_off[sg]));
            for (uint32_t pl = sig_off[sg]; pl < sig_off[sg + 1]; ++pl) {
                const uint32_t ncc_pl = pool_off[pl + 1] - pool_off[pl];
                if (ncc_pl > 255u) { fits = false; break; }
                flat.push_back((uint16_t)(pool_glimit[pl] << 8 | ncc_pl));
                for (uint32_t k = pool_off[pl]; k < pool_off[pl + 1]; ++k) flat.push_back((uint16_t)((cc[k].cls & 0xFFu) << 8 | cc[k].cnt));
            }
        }
        if (flat.size() & 1) flat.push_back(0);
        c->flat_words = fits ? (uint32_t)flat.size() : 0;
        if (fits) {
            HIPCHK(c, c->sig_flat.reserve(flat.size()));
            HIPCHK(c, hipMemcpy(c->sig_flat.p, flat.data(), flat.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        }
    }
    {   // the dictionary by pool type (DictView::flat2; dict_stream.h): distinct pools, and every signature as (type, multiplicity) pairs
        static_assert(kPoolSlotsMax == kPoolSlots, "dict_stream.h and step_digest.h agree on the slot capacity");
        const SigDict hd{sig_off, pool_off, pool_glimit, cc, nsig};
        const std::vector<uint16_t> f2 = build_typed_stream(hd);
        static const bool typed_off = tune_env("NHDFIT_NO_POOL_TYPES") != nullptr;   // tuning aid: the digest walks pool by pool
        c->flat2_words = !typed_off ? (uint32_t)f2.size() : 0;
        if (c->flat2_words) {
            HIPCHK(c, c->sig_flat2.reserve(f2.size()));
            HIPCHK(c, hipMemcpy(c->sig_flat2.p, f2.data(), f2.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        }
    }
    c->ncls = ncls;
    c->nsig = nsig;
    c->ngs = n_group_sets ? n_group_sets : 1;
    c->max_cores = max_cores_per_numa;
    c->max_gpus = max_gpus_per_numa;
    c->P = 0;                                       // staged tables (if any) were built for the old dictionary
#ifdef NHDFIT_TUNING
    HIPCHK(c, hipFuncSetAttribute((const void*)k_fit_only<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_fit_only<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_role<512, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_role<512, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#endif
    HIPCHK(c, hipFuncSetAttribute((const void*)k_step<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_find<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_find1<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_step<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_step<512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_step<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return NHDFIT_OK;
}

int nhdfit_reserve_nodes(nhdfit_ctx* c, uint32_t capacity, uint64_t global_base) {
    if (!c) return NHDFIT_E_INVAL;
    HIPCHK(c, hipSetDevice(c->dev));
    { int rc_ = sync_all(c); if (rc_) return rc_; }
    if (capacity > c->capacity) {
        c->n = 0;                                   // growing drops the contents: the caller re-uploads
        c->n_wide = 0; c->wide_index.clear();
        c->rec_all = true;
        const size_t padded = ((size_t)capacity + 63) & ~size_t(63);       // the fit role reads whole 64-node chunks
        HIPCHK(c, c->p0.reserve(padded)); HIPCHK(c, c->p1.reserve(padded)); HIPCHK(c, c->p2.reserve(padded));
        HIPCHK(c, c->p3.reserve(padded)); HIPCHK(c, c->p4.reserve(padded)); HIPCHK(c, c->det.reserve(capacity));
        HIPCHK(c, c->origin.reserve(capacity)); c->origin_hi = 0;
        for (auto& r : c->rec) HIPCHK(c, r.reserve(padded));
        for (auto& r : c->rec_bt) HIPCHK(c, r.reserve(padded));
        HIPCHK(c, hipMemset(c->p4.p, 0, padded * sizeof(nhdfit_plane4)));   // busy times of the padding lanes: any finite value
        c->capacity = capacity;
    }
    c->global_base = global_base;
    return NHDFIT_OK;
}

int nhdfit_set_node_count(nhdfit_ctx* c, uint32_t n) {
    if (!c) return NHDFIT_E_INVAL;
    if (n > c->capacity) return fail(c, NHDFIT_E_INVAL, "node count %u exceeds reserved capacity %u", n, c->capacity);
    if (n != c->n) {
        HIPCHK(c, hipSetDevice(c->dev));
        { int rc_ = sync_all(c); if (rc_) return rc_; }     // mapping phases of steps in flight still read the old count
        c->rec_all = true;                                  // the padding records of the last chunk move
        c->n_items = 0;
        c->n = n;
        if (c->n_wide) {                                    // wide records past the new end go with their nodes
            std::vector<nhdfit_wide_node> h(c->n_wide);
            HIPCHK(c, hipMemcpy(h.data(), c->wide.p, h.size() * sizeof h[0], hipMemcpyDeviceToHost));
            uint32_t keep = 0;
            while (keep < c->n_wide && h[keep].index < n) ++keep;
            c->n_wide = keep;
            c->wide_index.resize(keep);
        }
    }
    return NHDFIT_OK;
}

int nhdfit_upload_nodes(nhdfit_ctx* c, uint32_t first, uint32_t count, const nhdfit_plane0* p0, const nhdfit_plane1* p1,
                        const nhdfit_plane2* p2, const nhdfit_plane3* p3, const nhdfit_plane4* p4, const nhdfit_detail* det) {
    if (!c) return NHDFIT_E_INVAL;
    if (!count) return NHDFIT_OK;
    if (!p0 || !p1 || !p2 || !p3 || !p4 || !det) return fail(c, NHDFIT_E_INVAL, "NULL plane");
    if ((uint64_t)first + count > c->capacity) return fail(c, NHDFIT_E_INVAL, "upload [%u,%u) exceeds capacity %u", first, first + count, c->capacity);
    HIPCHK(c, hipSetDevice(c->dev));
    { int rc_ = sync_all(c); if (rc_) return rc_; }     // a step in flight must not see a half-written record
    if (first + count > c->n) { c->rec_all = true; c->n_items = 0; }   // the node count changes: the last chunk's padding moves
    else if (c->rec_lo == c->rec_hi) { c->rec_lo = first; c->rec_hi = first + count; }
    else { c->rec_lo = std::min(c->rec_lo, first); c->rec_hi = std::max(c->rec_hi, first + count); }
    HIPCHK(c, hipMemcpy(c->p0.p + first, p0, count * sizeof *p0, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->p1.p + first, p1, count * sizeof *p1, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->p2.p + first, p2, count * sizeof *p2, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->p3.p + first, p3, count * sizeof *p3, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->p4.p + first, p4, count * sizeof *p4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->det.p + first, det, count * sizeof *det, hipMemcpyHostToDevice));
    if (first + count > c->n) c->n = first + count;
    return NHDFIT_OK;
}

namespace {
// Tile-image layouts for the current dictionary, staged batch and provisioned X rows; (re)sizes the image buffers.
int refresh_layouts(nhdfit_ctx* c) {
    uint32_t pitch = 0, hot = 0;
    for (uint32_t w = 0; w < (uint32_t)kWClasses; ++w) {
        c->L[w] = make_layout(2u << w, c->max_cores, c->max_gpus, c->nsig, c->ngs, c->hp_rows, c->x_cap);
        if (w <= c->max_wcls) { pitch = std::max(pitch, c->L[w].bytes); hot = std::max(hot, c->L[w].hot_bytes); }
    }
    // What a fit block stages in LDS: the whole hot section if it fits next to the winner scratch (8 wavefronts x 64
    // pods x 8 B); otherwise - more node classes than LDS has rows for - a prefix that leaves room for a second block
    // per CU, plus the HP rows; the X rows beyond the prefix are read from global memory (L2): slower, not wrong.
    const uint32_t scratch = 8 * 64 * sizeof(unsigned long long), hp_bytes = align16(c->hp_rows * 8);
    const bool spill = (size_t)hot + scratch > kLdsPerCu;
    uint32_t lds = 0;
    for (uint32_t w = 0; w < (uint32_t)kWClasses; ++w) {
        const Layout& L = c->L[w];
        if (w <= c->max_wcls && L.hot_hp + hp_bytes > 0xFFFFu * 8u)
            return fail(c, NHDFIT_E_LIMIT, "%u node classes: the table rows of a pod tile no longer fit 16-bit row offsets (512 KiB)", c->nx);
        uint32_t staged = L.hot_bytes;
        if (spill && w <= c->max_wcls && (size_t)L.hot_bytes + scratch > kLdsPerCu / 2) {
            staged = (uint32_t)(kLdsPerCu / 2 - scratch - hp_bytes) & ~15u;
            if (staged < L.hot_x + L.x_stride)
                return fail(c, NHDFIT_E_LIMIT, "the fixed table rows of a pod tile need %u bytes of LDS: %u node-group sets, %u hugepage rows, "
                            "%u cores per socket", L.hot_x + hp_bytes, c->ngs, c->hp_rows, c->max_cores);
            staged = std::min(staged, L.hot_hp);
        }
        c->hot_staged[w] = staged;
        if (w <= c->max_wcls) lds = std::max(lds, staged < L.hot_bytes ? staged + hp_bytes : L.hot_bytes);
    }
    // Pair table of the narrow tiles (fit_core.h "pair rows"), derived by every fit block behind its winner scratch: taken
    // while a block's LDS stays within a third of the CU's (three 512-thread blocks per CU is what the registers allow, and
    // the launch's dynamic LDS is one size for all of its blocks).
    c->pair_D[0] = c->pair_D[1] = 0;
    if (c->pair_rows && !spill) {
#ifndef NHDFIT_LDS_BLOCKS
#define NHDFIT_LDS_BLOCKS 3
#endif
        const uint32_t budget = (uint32_t)(kLdsPerCu / NHDFIT_LDS_BLOCKS) & ~1023u;
        for (uint32_t w = 0; w < 2 && w <= c->max_wcls; ++w) {
            const Layout& L = c->L[w];
            const uint32_t D = pair_dim(c->max_demand[w], L.fc_dim);
            const uint32_t base = (uint32_t)lds_slice(L.hot_bytes) + scratch, c_bytes = 2 * D * D * L.row;
            if (base + c_bytes > budget) continue;
            c->pair_D[w] = D;
            lds = std::max(lds, base + c_bytes - scratch);
        }
    }
    c->pitch = pitch;
    c->lds_bytes = lds;
    c->x_spill = spill;
    if (c->P) {
        const uint32_t tiles = (c->P + kTile - 1) / kTile;
        for (Pipe& p : c->pipe)
            for (int b = 0; b < kBufs; ++b) HIPCHK(c, p.tabs[b].reserve((size_t)tiles * pitch));
    }
    return NHDFIT_OK;
}
}  // namespace

namespace {
// The order a call's pods are staged in (perm[k] = caller's index of the pod at device position k): a function of the requests
// alone.  It is also the order in which EVERY form of a sharded find hands its score words to the all-reduce - the staged step and
// the single-launch find alike - so that ranks that take different forms for the same call (one shard holds a wide node, spilled
// class rows, a launch that gave up: rank-local facts) still reduce pod against pod.
// `seen` (optional): what the staging wants to know of every request anyway - one pass over the records instead of three
struct StagedSeen { int32_t hp_min = 0, hp_max = 0; uint32_t n_big = 0; };
void staged_order(const nhdfit_req* reqs, uint32_t P, std::vector<uint32_t>& perm, StagedSeen* seen = nullptr) {
    perm.resize(P);
    static thread_local std::vector<uint16_t> key;
    key.resize(P);
    uint32_t start[512 + 1] = {0};
    StagedSeen sn;
    for (uint32_t p = 0; p < P; ++p) {
        const int32_t hp = reqs[p].hugepages_gb;
        sn.hp_min = hp < sn.hp_min ? hp : sn.hp_min; sn.hp_max = hp > sn.hp_max ? hp : sn.hp_max;
        sn.n_big += reqs[p].n_groups > 3;
        const PodHeader h = pod_header(reqs[p]);
        // group count is the major key, descending: the tiles with the most assignments to sweep are the
        // first blocks of the fit grid (longest-first keeps the tail of the launch short)
        key[p] = (uint16_t)(((h.flags & kPodValid) ? 0u : 1u << 8) | ((reqs[p].n_groups > 3 ? 1u : 0u) << 7) |
                            ((15u - (reqs[p].n_groups & 15u)) << 3) | ((h.flags & (kPodNeedGpu | kPodPci | kPodFilter)) >> 1));
        start[key[p] + 1]++;
    }
    for (uint32_t k = 0; k < 512; ++k) start[k + 1] += start[k];          // stable counting sort: 9-bit keys
    for (uint32_t p = 0; p < P; ++p) perm[start[key[p]]++] = p;
    if (seen) *seen = sn;
}
}  // namespace

namespace {
// What the staging leaves in page-locked host memory goes to the device by copy commands - one per array, and every command costs the
// copy engine ~10 us before its first byte moves.  The single-launch find of a batch reads the small arrays (tile classes, work items)
// straight from the host block and, for a batch of a few tiles, the request records too; the copies it skipped are made up for here
// when the call takes the steps' path after all.
int stage_requests(nhdfit_ctx* c, const nhdfit_req* reqs, uint32_t P, bool defer_small_copies, bool defer_request_copy, size_t tail_bytes = 0);
int finish_deferred_copies(nhdfit_ctx* c) {
    const uint32_t tiles = (c->P + kTile - 1) / kTile;
    if (c->reqs_deferred) HIPCHK(c, hipMemcpyAsync(c->reqs.p, c->pin_reqs.p, (size_t)c->P * sizeof(nhdfit_req), hipMemcpyHostToDevice, c->stream));
    if (c->wcls_deferred) HIPCHK(c, hipMemcpyAsync(c->tile_wcls.p, c->pin_wcls.p, tiles, hipMemcpyHostToDevice, c->stream));
    c->reqs_deferred = c->wcls_deferred = false;
    return NHDFIT_OK;
}
}  // namespace

int nhdfit_stage_requests(nhdfit_ctx* c, const nhdfit_req* reqs, uint32_t P) {
    if (!c) return NHDFIT_E_INVAL;
    return stage_requests(c, reqs, P, false, false);
}

namespace {
int stage_requests(nhdfit_ctx* c, const nhdfit_req* reqs, uint32_t P, bool defer_small_copies, bool defer_request_copy, size_t tail_bytes) {
    if (!reqs || !P) return fail(c, NHDFIT_E_INVAL, "no requests");
    if (!c->nsig) return fail(c, NHDFIT_E_STATE, "set the dictionary first");
    static const bool prof = tune_env("NHDFIT_FIND_PROF") != nullptr;      // tuning aid: host-side phase times of the staging
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!prof) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[nhdfit]   stage P=%u %s %.1f us\n", P, what, std::chrono::duration<double, std::micro>(t1 - t_prev).count());
        t_prev = t1;
    };
    // (a single-launch find that saw its word left the streams idle, and nothing has been asked of the runtime since: asking the first
    // stream would only make it reap that launch now - ~12 us of every batch call)
    const bool idle = c->known_idle && c->ev_pending == 0;
    HIPCHK(c, hipSetDevice(c->dev));
    if (!idle) {
        { int rc_ = sync_all(c); if (rc_) return rc_; }
        { int rc_ = drain_events(c); if (rc_) return rc_; }
    }
    lap("streams idle, events read");
    for (Pipe& p : c->pipe) p.n_dig = p.n_fit = p.n_shaped = p.n_chosen = p.n_finished = 0;
    c->n_enq = 0;
    c->last_pipe = 0;
    c->staged_gen++;
    const uint32_t tiles = (P + kTile - 1) / kTile;
    // Pods are staged sorted by request class so that 64-pod tiles are homogeneous (narrow table rows, fast sweep of
    // the fit role) and the lanes of the mapping roles have similar group counts; results are un-permuted in fetch.
    StagedSeen seen;
    staged_order(reqs, P, c->perm, &seen);
    if (seen.hp_min < 0)
        for (uint32_t p = 0; p < P; ++p)
            if (reqs[p].hugepages_gb < 0) return fail(c, NHDFIT_E_INVAL, "pod %u asks for a negative number of hugepages", p);
    const int32_t hp_max = seen.hp_max;
    if (hp_max > kMaxHpRows - 2)
        return fail(c, NHDFIT_E_LIMIT, "a pod asks for %d GiB of hugepages (limit %d)", hp_max, kMaxHpRows - 2);
    const size_t tail_records = (tail_bytes + sizeof(nhdfit_req) - 1) / sizeof(nhdfit_req);   // (room behind the records for what rides the same copy)
    HIPCHK(c, c->reqs.reserve((size_t)P + tail_records));
    for (Pipe& p : c->pipe)
        if (tiles > p.dig_count.cap) {                 // (the digest role leaves its arrival counters at zero: cleared when the buffer is new)
            HIPCHK(c, p.dig_count.reserve(std::max<size_t>(tiles, 256)));
            HIPCHK(c, hipMemsetAsync(p.dig_count.p, 0, p.dig_count.cap * sizeof(uint32_t), c->stream));
        }
    for (Pipe& p : c->pipe)
        for (int b = 0; b < kBufs; ++b) {
            HIPCHK(c, p.hdr[b].reserve((size_t)tiles * kTile));
            HIPCHK(c, p.score[b].reserve(P));
            HIPCHK(c, p.maps[b].reserve(P));
        }
    lap("order, buffers");
    c->n_big_pods = seen.n_big;
    HIPCHK(c, c->pin_reqs.reserve((size_t)P + tail_records));
    nhdfit_req* sorted = c->pin_reqs.p;                                   // (free again: sync_all above waited for the last copy out of it)
    // tile by tile: the records gathered into the page-locked block, then - while they are in the core's cache - the tile's row width
    // class (2^(largest group count among its valid pods) assignments: the digest role derives the same class from the same records)
    // and, for the narrow tiles, the largest CPU demand (the pair table's dimension)
    c->h_tile_wcls.assign(tiles, 0);
    c->max_wcls = 0;
    c->max_demand[0] = c->max_demand[1] = 0;
    for (uint32_t t = 0; t < tiles; ++t) {
        const uint32_t lo = t * kTile, hi = std::min(P, lo + (uint32_t)kTile);
        uint8_t w = 0;
        for (uint32_t i = lo; i < hi; ++i) {
            sorted[i] = reqs[c->perm[i]];
            if (req_valid(sorted[i])) w = std::max(w, (uint8_t)wclass_of(sorted[i].n_groups));
        }
        c->h_tile_wcls[t] = w;
        if (w > c->max_wcls) c->max_wcls = w;
        if (w < 2)
            for (uint32_t i = lo; i < hi; ++i)
                if (req_valid(sorted[i])) c->max_demand[w] = std::max(c->max_demand[w], req_max_demand(sorted[i]));
    }
    lap("gather");
    // ONE copy: the batch in four pieces, each piece's transfer beside the next piece's gather, was measured and is slower - every copy
    // command costs the copy engine ~10 us before its first byte moves (config 4: 0.166 -> 0.193 ms per call, profiles/r05)
    c->reqs_deferred = defer_request_copy;
    if (!defer_request_copy) HIPCHK(c, hipMemcpyAsync(c->reqs.p, sorted, (size_t)P * sizeof *reqs, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, c->tile_wcls.reserve(tiles));
    HIPCHK(c, c->pin_wcls.reserve(tiles));
    memcpy(c->pin_wcls.p, c->h_tile_wcls.data(), tiles);
    c->wcls_deferred = defer_small_copies;
    if (!defer_small_copies) HIPCHK(c, hipMemcpyAsync(c->tile_wcls.p, c->pin_wcls.p, tiles, hipMemcpyHostToDevice, c->stream));
    c->P = P;
    c->hp_rows = (uint32_t)hp_max + 2;
    c->n_items = 0;                                 // the fit role's work items are rebuilt at the next step
    c->use_cand = false;
    lap("copies enqueued");
    { int rc_ = refresh_layouts(c); if (rc_) return rc_; }
    lap("layouts");
    // a node's C row depends on the pair table's dimension: a batch that changes it has the chunks' records dealt to the lanes again
    // (ensure_records, in front of the batch's first step) - a full tile and more only: smaller batches are latency, not throughput
    if (P > (uint32_t)kTile && (c->pair_D[0] != c->order_D[0] || c->pair_D[1] != c->order_D[1])) c->ord_all = true;
    // ... and so does the row itself, which the records carry ready-made (NodeRec::flags): rewritten likewise (k_xcrow); a smaller batch
    // of another dimension sweeps six rows per pair of assignments meanwhile (FitArgs::crow_ok = 0)
    if (P > (uint32_t)kTile && (c->pair_D[0] != c->crow_D[0] || c->pair_D[1] != c->crow_D[1])) c->crow_stale = true;
    return NHDFIT_OK;
}
}  // namespace

static int stage_cand(nhdfit_ctx* c, const uint64_t* cand) {
    const size_t chunks = (c->n + 63) / 64;
    HIPCHK(c, c->cand.reserve(chunks ? chunks : 1));
    HIPCHK(c, hipMemcpy(c->cand.p, cand, chunks * sizeof(uint64_t), hipMemcpyHostToDevice));
    c->cand_shadow.clear();
    c->use_cand = true;
    return NHDFIT_OK;
}

namespace {

// `every`: all chunks, for the pair-table dimensions of the staged batch (they become order_D); otherwise the chunks k_xrecords just
// rewrote, for the dimensions the others were dealt for
int order_chunks(nhdfit_ctx* c, uint32_t first_chunk, uint32_t n_chunks, bool every) {
    if (every) { c->order_D[0] = c->pair_D[0]; c->order_D[1] = c->pair_D[1]; }
    if (!n_chunks) return NHDFIT_OK;
    OrderArgs o;
    memset(&o, 0, sizeof o);
    for (int w = 0; w < kWClasses; ++w) { o.rec[w] = c->rec[w].p; o.bt[w] = c->rec_bt[w].p; }
    o.first_chunk = first_chunk; o.n_chunks = n_chunks;
    o.pair_D[0] = c->order_D[0] == ~0u ? 0u : c->order_D[0]; o.pair_D[1] = c->order_D[1] == ~0u ? 0u : c->order_D[1];
    LAUNCH(c, k_xorder, dim3((n_chunks * (uint32_t)kWClasses + 3) / 4), dim3(256), 0, c->stream, o);
    HIPCHK(c, hipGetLastError());
    return NHDFIT_OK;
}

// Bring the node records (and the class table behind their X rows) up to date with the mirror.  Cheap no-op when
// nothing changed; otherwise three small kernels over the touched nodes and one 8-byte read-back (the host sizes the
// tile images by the class count).  A grown class count or a changed dictionary re-does every record.
int ensure_records(nhdfit_ctx* c) {
    const uint32_t all_chunks = (c->n + 63) / 64;
    // dealing the records to the lanes pays where the fit role is more than a launch: from 128 chunks on
    const bool deal = c->lane_order && all_chunks >= 128;
    // the C rows the records carry, for the staged batch's pair-table dimensions: every record's flags word (unless all records are
    // about to be written anyway) - in front of whatever else rewrites records, which then writes for the new dimensions too
    const bool crow = c->crow_stale;
    if (crow) { c->crow_D[0] = c->pair_D[0]; c->crow_D[1] = c->pair_D[1]; c->crow_stale = false; }
    auto rewrite_crows = [&]() -> int {
        if (!c->n || c->rec_all || !c->rec[0].p || !c->rec[1].p) return NHDFIT_OK;
        CrowArgs k;
        k.rec[0] = c->rec[0].p; k.rec[1] = c->rec[1].p; k.npad = (c->n + 63) & ~63u; k.D[0] = c->crow_D[0]; k.D[1] = c->crow_D[1];
        LAUNCH(c, k_xcrow, dim3((k.npad + 255) / 256), dim3(256), 0, c->stream, k);
        HIPCHK(c, hipGetLastError());
        return NHDFIT_OK;
    };
    if (!c->rec_all && c->rec_lo == c->rec_hi) {
        if (c->n && (crow || (c->ord_all && deal))) {           // the records stand, a staged batch changed the pair table's dimension
            { int rc_ = sync_all(c); if (rc_) return rc_; }     // (steps in flight read the records)
            c->staged_gen++;
            if (crow) { int rc_ = rewrite_crows(); if (rc_) return rc_; }
            if (c->ord_all && deal) { int rc_ = order_chunks(c, 0, all_chunks, true); if (rc_) return rc_; }
        }
        c->ord_all = false;
        return NHDFIT_OK;
    }
    c->staged_gen++;                                            // (its kernels run on pipe 0's stream)
    if (!c->n) { c->rec_all = false; c->rec_lo = c->rec_hi = 0; c->ord_all = false; return NHDFIT_OK; }
    if (crow) { { int rc_ = sync_all(c); if (rc_) return rc_; } { int rc_ = rewrite_crows(); if (rc_) return rc_; } }
    const uint32_t npad = (c->n + 63) & ~63u;
    uint32_t dealt_first = 0, dealt_count = 0;
    for (int pass = 0; pass < 2; ++pass) {
        // whole chunks: k_xrecords writes the records in node order (whatever order the chunk's lanes had before)
        const uint32_t first = c->rec_all ? 0 : c->rec_lo & ~63u, count = c->rec_all ? npad : std::min(npad, (c->rec_hi + 63u) & ~63u) - first;
        dealt_first = first / 64; dealt_count = count / 64;
        RecArgs r;
        memset(&r, 0, sizeof r);
        r.p0 = c->p0.p; r.p1 = c->p1.p; r.p2 = c->p2.p; r.p3 = c->p3.p; r.p4 = c->p4.p;
        r.n = c->n; r.npad = npad; r.first = first; r.count = count;
        r.fc_dim = c->max_cores + 1; r.fg_dim = c->max_gpus + 1; r.ngs = c->ngs;
        r.x = XTable{c->xkeys.p, c->xids.p, c->xcls.p, c->xnx.p};
        for (int w = 0; w < kWClasses; ++w) {
            // records hold hot-section offsets: they depend on the dictionary and the provisioned X rows, not on the batch
            r.L[w] = make_layout(2u << w, c->max_cores, c->max_gpus, c->nsig, c->ngs, 2, c->x_cap);
            r.rec[w] = c->rec[w].p;
            r.bt[w] = c->rec_bt[w].p;
        }
        r.crow_D[0] = c->crow_D[0]; r.crow_D[1] = c->crow_D[1];
        const dim3 grid((count + 255) / 256), block(256);
        if (pass == 0) {
            LAUNCH(c, k_xkeys, grid, block, 0, c->stream, r);
            LAUNCH(c, k_xassign, dim3(1), dim3(1024), 0, c->stream, r.x);
        }
        LAUNCH(c, k_xrecords, grid, block, 0, c->stream, r);
        HIPCHK(c, hipGetLastError());
        if (pass == 1) break;
        uint32_t nx[2] = {0, 0};
        HIPCHK(c, hipMemcpyAsync(nx, c->xnx.p, sizeof nx, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, wait_stream(c->stream));
        if (nx[1] || nx[0] > kXSlots / 2) return fail(c, NHDFIT_E_LIMIT, "more than %u distinct (free GPUs, NIC signature) node classes", kXSlots / 2);
        if (nx[0] != c->nx)                                     // new classes: a table image digested ahead of its fit has no X rows for
            for (Pipe& p : c->pipe)                             // them - it is digested again (nhdfit_enqueue_step looks here first)
                if (p.n_dig > p.n_fit) p.n_dig = p.n_fit;
        c->nx = nx[0];
        if (c->sig_use_on && c->sig_use_nx != c->nx) {              // the signatures the classes refer to (classes are only ever added)
            std::vector<uint64_t> keys(c->nx);
            if (c->nx) HIPCHK(c, hipMemcpy(keys.data(), c->xcls.p, (size_t)c->nx * sizeof(uint64_t), hipMemcpyDeviceToHost));
            std::vector<uint8_t> used(0x10000u, 0);
            for (uint64_t k : keys) { used[xkey_sig_numa(k)] = 1; used[xkey_sig_pci(k)] = 1; }
            std::vector<uint16_t> list;
            for (uint32_t sg = 0; sg < 0x10000u; ++sg)
                if (used[sg]) list.push_back((uint16_t)sg);
            HIPCHK(c, c->sig_use.reserve(list.size() ? list.size() : 1));
            if (!list.empty()) HIPCHK(c, hipMemcpy(c->sig_use.p, list.data(), list.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            c->n_sig_use = (uint32_t)list.size();
            c->sig_use_nx = c->nx;
        }
        if (nx[0] <= c->x_cap) break;
        // more classes than provisioned rows: every hot-section offset moves -> staged tables and all records are redone
        { int rc_ = sync_all(c); if (rc_) return rc_; }
        c->x_cap = x_capacity(nx[0]);
        c->rec_all = true;
        for (Pipe& p : c->pipe) p.n_dig = std::min(p.n_dig, p.n_fit);   // every staged table image is redone
        int rc = refresh_layouts(c);
        if (rc) return rc;
    }
    if (deal) {
        // a few rewritten chunks (the scheduler's loop: one commit, then the next find) stay in node order until the next full deal -
        // a launch of its own in front of a 60 us call costs more than their bank conflicts
        const bool every = c->ord_all || c->rec_all;
        if (every || dealt_count >= 16) { int rc_ = order_chunks(c, every ? 0 : dealt_first, every ? all_chunks : dealt_count, every); if (rc_) return rc_; }
    }
    c->ord_all = false;
    c->rec_all = false;
    c->rec_lo = c->rec_hi = 0;
    return NHDFIT_OK;
}

// Work items of the fit role (one block each): the tiles' chunk ranges cut so that every block carries about the
// same cost - a chunk of a tile with W assignments costs ~(6 + W) - and the block count is a fixed multiple of what
// the chip holds at once (NHDFIT_FIT_BLOCKS overrides the target).  Wide tiles first: longest-first keeps the tail
// of the launch short.  Small problems simply get one wavefront-run per chunk.
int build_items(nhdfit_ctx* c, uint32_t nw, bool batch_find = false) {
    const uint32_t tiles = (c->P + kTile - 1) / kTile, chunks = (c->n + 63) / 64;
    const uint32_t cus = (uint32_t)c->prop.multiProcessorCount;
    // two of the three 512-thread blocks a CU holds (the side roles of the same launch live in the third slot)
    uint32_t target = c->fit_blocks ? c->fit_blocks : cus * (nw == 8 ? 2u : 5u);          // see k_step: the fit role leads the grid
    uint64_t total = 0;
    for (uint32_t t = 0; t < tiles; ++t) total += (uint64_t)chunks * (6u + (2u << c->h_tile_wcls[t]));
    std::vector<FitItem> items;
    // XCD-aware form (big problems): the fit role leads the grid and block b runs on XCD b % 8 (observed placement, used
    // for speed only), so every tile gets a multiple of 8 blocks and block j of a tile works inside eighth j % 8 of the
    // node axis: an XCD's L2 then only ever sees its eighth of the node records (0.7 MB at 65 536 nodes instead of all
    // 5.5 MB of the three row widths + busy times - more than the 4 MB an XCD has), re-read once per pod tile.
    static const bool xcd_items = !(tune_env("NHDFIT_XCD_ITEMS") && atoi(tune_env("NHDFIT_XCD_ITEMS")) == 0);
    // ... unless the shard's records fit any XCD's L2 several times over (config 5's shard: 32 768 nodes, 1.8 MB with all three row
    // widths and the busy times): then there is nothing to partition for, and a batch of many tiles is better served by FEW, LONG
    // blocks per tile - every fit block stages the tile's hot section and derives its pair tables first (config 5: 35-50 KB), and
    // 2 300 blocks of eight chunks per wavefront spent a fifth of the step doing that.
    const uint64_t rec_bytes = (uint64_t)chunks * 64u * (16u * (c->max_wcls + 1u) + 8u);
    const bool small_shard = rec_bytes <= (2u << 20);
    if (small_shard && nw == 8 && !c->fit_blocks) target = cus;          // (config 5 shard x 16 384 pods: 256 blocks 44.9 us per step, 512: 45.9, 1 024: 47.6, 2 048: 51.7)
    // (the single-launch find of a batch, k_findn, runs 256-thread blocks: the same cut into eighths of the node axis, two pieces each)
    const bool by_xcd = xcd_items && !c->fit_blocks && chunks >= 8u * 4u * nw && (nw == 8 || batch_find) && !small_shard;
    for (uint32_t t = 0; t < tiles; ++t) {           // staged order = widest tiles first
        const uint32_t w = c->h_tile_wcls[t];
        const uint64_t cost = (uint64_t)chunks * (6u + (2u << w));
        uint32_t nb = (uint32_t)((cost * target + total / 2) / total);
        nb = std::max(1u, std::min(nb, (chunks + nw - 1) / nw));          // at least one chunk per wavefront
        if (small_shard && nw == 8 && !c->fit_blocks) nb = std::min(nb, 8u);   // (few tiles: as the XCD form would cut them)
        if (by_xcd) {
            static const uint32_t force_k = tune_env("NHDFIT_XCD_K") ? (uint32_t)atoi(tune_env("NHDFIT_XCD_K")) : 0u;   // tuning aid
            // 8, 16 or 32 blocks per tile by its cost when one launch has the chip to itself; with two pipes the other launch's
            // blocks fill the gaps, and fewer, longer fit blocks (less staging, fewer tails) win: 8 per tile (-3 %, profiles/r03)
            const bool two_pipes = c->dual && !c->split && !c->role_kernels;
            static const uint32_t force_k8 = tune_env("NHDFIT_XCD_K8") ? (uint32_t)atoi(tune_env("NHDFIT_XCD_K8")) : 0u;   // tuning aid: the widest tiles only
            const uint32_t k = force_k8 && w >= 2 ? force_k8 : force_k ? force_k : batch_find ? 2u : two_pipes ? 1u : nb <= 11 ? 1u : nb <= 23 ? 2u : 4u;
            // tuning aid (NHDFIT_FIT_HALF = mask of row-width classes): FOUR blocks for a tile of such a class, a quarter of the node axis each -
            // half as many stagings and pair-table derivations per tile, twice the chunks per wavefront.  Two such tiles share eight
            // consecutive blocks (quarter q of the first on XCD q, of the second on XCD 4 + q); an odd one out gets its eight.
            static const uint32_t half_mask = tune_env("NHDFIT_FIT_HALF") ? (uint32_t)atoi(tune_env("NHDFIT_FIT_HALF")) : 0u;
            const bool next_same = t + 1 < tiles && c->h_tile_wcls[t + 1] == w;
            if (!batch_find && k == 1 && (half_mask >> w & 1u) && (items.size() % 8 == 4 || next_same)) {
                for (uint32_t q = 0; q < 4; ++q)
                    items.push_back(FitItem{t, w, (uint32_t)((uint64_t)chunks * q / 4), (uint32_t)((uint64_t)chunks * (q + 1) / 4)});
                continue;
            }
            for (uint32_t j = 0; j < 8 * k; ++j) {
                const uint32_t r = (j % 8) * k + j / 8;                     // range r of 8k: the (j / 8)-th piece of eighth j % 8
                const uint32_t lo = (uint32_t)((uint64_t)chunks * r / (8 * k)), hi = (uint32_t)((uint64_t)chunks * (r + 1) / (8 * k));
                items.push_back(FitItem{t, w, lo, hi});                     // (never empty: chunks >= 256; keeps b % 8 aligned)
            }
            continue;
        }
        for (uint32_t b = 0; b < nb; ++b) {
            const uint32_t lo = (uint32_t)((uint64_t)chunks * b / nb), hi = (uint32_t)((uint64_t)chunks * (b + 1) / nb);
            if (hi > lo) items.push_back(FitItem{t, w, lo, hi});
        }
    }
    HIPCHK(c, c->items.reserve(items.size() ? items.size() : 1));
    // page-locked staging, no wait: the list is only rebuilt after something drained the stream (stage_requests,
    // upload_nodes, set_node_count all sync first), so the previous copy out of this buffer is long done
    HIPCHK(c, c->pin_items.reserve(items.size() * sizeof(FitItem) + (size_t)tiles * sizeof(uint32_t) + 1));
    memcpy(c->pin_items.p, items.data(), items.size() * sizeof(FitItem));
    if (batch_find) {                                           // (k_findn reads the list where it is: no copy command in front of the launch)
        uint32_t* per_tile = reinterpret_cast<uint32_t*>(c->pin_items.p + items.size() * sizeof(FitItem));   // how many tickets each tile's last block waits for
        for (uint32_t t = 0; t < tiles; ++t) per_tile[t] = 0;
        for (const FitItem& it : items) per_tile[it.tile]++;
    } else
        HIPCHK(c, hipMemcpyAsync(c->items.p, c->pin_items.p, items.size() * sizeof(FitItem), hipMemcpyHostToDevice, c->stream));
    c->n_items = (uint32_t)items.size();
    c->staged_gen++;
    return NHDFIT_OK;
}

// Argument blocks of the roles for buffer set b of pipe p (shared by the step launch and the single-launch find).
MapArgs make_map_args(nhdfit_ctx* c, Pipe& p, int b) {
    MapArgs m;
    memset(&m, 0, sizeof m);
    m.p0 = c->p0.p; m.p1 = c->p1.p; m.p2 = c->p2.p; m.p3 = c->p3.p; m.det = c->det.p;
    m.tabs = p.tabs[b].p; m.pitch = c->pitch; m.tile_wcls = c->tile_wcls.p;
    for (int w = 0; w < kWClasses; ++w) m.L[w] = cold_view(c->L[w]);
    m.n = c->n; m.global_base = c->global_base; m.reqs = c->reqs.p; m.P = c->P;
    m.score = p.score[b].p; m.caps = c->caps.p; m.out = p.maps[b].p;
    return m;
}
ShapeArgs make_shape_args(nhdfit_ctx* c, Pipe& p, int b) {
    return ShapeArgs{p.shape_keys[b].p, p.shape_res[b].p, p.shape_slot[b].p, p.shape_list[b].p, c->asc.p,
                     c->use_choose_tab ? c->choose_tab.p : nullptr,
                     c->use_set_states ? SetStates{c->st_info.p, c->st_next.p, c->st_asc.p, c->st_n} : SetStates{nullptr, nullptr, nullptr, 0}};
}
void fill_digest_args(nhdfit_ctx* c, Pipe& p, int b, uint32_t wc_parts, uint32_t sig_parts, DigestArgs& d) {
    d.reqs = c->reqs.p; d.P = c->P;
    d.d = DictView{c->caps.p, c->ncls, c->group_sets.p, SigDict{c->sig_off.p, c->pool_off.p, c->pool_glimit.p, c->cc.p, c->nsig}, c->sig_flat.p, c->flat_words,
                   c->sig_flat2.p, c->flat2_words};
    for (int w = 0; w < kWClasses; ++w) d.L[w] = c->L[w];
    d.pitch = c->pitch; d.tabs = p.tabs[b].p; d.hdr = p.hdr[b].p; d.score = p.score[b].p;
    d.xcls = c->xcls.p; d.nx = c->xnx.p;
    d.wc_parts = wc_parts;
    d.sig_parts = sig_parts;
    const bool listed = c->sig_use_on && !c->all_sigs && c->sig_use_nx == c->nx && c->nx != 0;
    d.sig_list = listed ? c->sig_use.p : nullptr;
    d.n_sig_list = listed ? c->n_sig_use : 0;
    d.count = p.dig_count.p;
}
void fill_fit_args(nhdfit_ctx* c, Pipe& p, int bf, double now, FitArgs& f, bool pair = false) {
    for (int w = 0; w < 2; ++w) {
        f.pair_D[w] = pair ? c->pair_D[w] : 0u;
        f.crow_ok[w] = pair && c->pair_D[w] != 0u && c->crow_D[w] == c->pair_D[w] && !c->crow_stale ? 1u : 0u;
        f.hot_wc1[w] = c->L[w].hot_wc1;
    }
    f.fc_dim = c->max_cores + 1;
    for (int w = 0; w < kWClasses; ++w) {
        f.rec[w] = c->rec[w].p; f.bt[w] = c->rec_bt[w].p; f.off_hot[w] = c->L[w].off_hot; f.hot_bytes[w] = c->L[w].hot_bytes; f.hot_hp[w] = c->L[w].hot_hp; f.hot_staged[w] = c->hot_staged[w];
    }
    f.hp_last = c->hp_rows - 1; f.hp_bytes = align16(c->hp_rows * 8);
    f.n = c->n; f.chunks = (c->n + 63) / 64; f.global_base = c->global_base; f.busy_from = busy_threshold(now);
    f.tabs = p.tabs[bf].p; f.pitch = c->pitch; f.hdr = p.hdr[bf].p; f.P = c->P;
    f.cand = c->use_cand ? c->cand.p : nullptr;
    f.nm = c->want_bitmap ? p.nm.p : nullptr;
    f.score = p.score[bf].p;
    f.items = c->items.p;
    f.dbg_skip = tune_env("NHDFIT_FIT_SKIP") ? (uint32_t)atoi(tune_env("NHDFIT_FIT_SKIP")) : 0;
    f.clk = nullptr;
}

// One launch of the step kernel with every role that has work (see k_step).  `with_fit`: the fit role for step
// n_fit plus the digest of step n_fit + 1; `flushing`: nothing new will follow, drain the mapping phases.
int launch_step(nhdfit_ctx* c, Pipe& p, bool with_fit, bool with_digest, double now, bool flushing) {
    c->side_streams_used = true;
    const uint32_t P = c->P, tiles = (P + kTile - 1) / kTile;
    const uint32_t chunks = (c->n + 63) / 64;
    const bool big = c->geom_big;
    const uint32_t block = big ? 512 : 256, nw = block / 64;
    const bool small_map = c->want_map && c->n_big_pods < P;
    if (with_fit || with_digest) { int rc_ = ensure_records(c); if (rc_) return rc_; }
    if (with_fit && !c->n_items) { int rc_ = build_items(c, nw); if (rc_) return rc_; }
    if (&p != &c->pipe[0] && p.seen_gen != c->staged_gen) {
        // what pipe 0's stream staged since this pipe last looked (requests, work items, node records) is in front of this launch
        HIPCHK(c, hipEventRecord(p.ev_staged, c->pipe[0].stream));
        HIPCHK(c, hipStreamWaitEvent(p.stream, p.ev_staged, 0));
        p.seen_gen = c->staged_gen;
    }

    StepArgs a;
    memset(&a, 0, sizeof a);
    a.shapes_P = P;
    a.side_prio = c->side_prio;
    auto map_args = [&](int b) { return make_map_args(c, p, b); };
    auto shape_args = [&](int b) { return make_shape_args(c, p, b); };
    // mapping phases of earlier steps: each advances by at most one step per launch
    bool did_shapes = false, did_choose = false, did_finish = false;
    if (small_map) {
        if (p.n_finished < p.n_chosen) {
            const int b = (int)(p.n_finished % kBufs);
            a.finish_m = map_args(b); a.finish_h = shape_args(b); a.nb_finish = (P + block / 4 - 1) / (block / 4); did_finish = true;
        }
        if (p.n_chosen < p.n_shaped) {
            a.choose = shape_args((int)(p.n_chosen % kBufs));
            // wavefront = tile, lane = shape (NHDFIT_CHOOSE_LANES=0: a wavefront per shape, 16 x the blocks - tuning aid)
            static const bool lanes_off = tune_env("NHDFIT_CHOOSE_LANES") && atoi(tune_env("NHDFIT_CHOOSE_LANES")) == 0;
            const bool lanes = !lanes_off;
            a.choose_lanes = lanes ? 1u : 0u;
            a.nb_choose = lanes ? (tiles + nw - 1) / nw : (tiles * c->choose_split + nw - 1) / nw; did_choose = true;
        }
        // scores of step s are final once its fit launch (and, sharded, its all-reduce) is done; sharded runs give
        // the all-reduce one launch of slack so that it overlaps the next fit instead of stalling the stream
        const uint64_t ready = c->comm && !flushing && p.n_fit ? p.n_fit - 1 : p.n_fit;
        if (p.n_shaped < ready) {
            const int b = (int)(p.n_shaped % kBufs);
            if (c->comm) HIPCHK(c, hipStreamWaitEvent(p.stream, p.ev_red[b], 0));
            HIPCHK(c, p.shape_keys[b].reserve((size_t)tiles * kTile));
            HIPCHK(c, p.shape_res[b].reserve((size_t)tiles * kTile));
            HIPCHK(c, p.shape_slot[b].reserve(P));
            HIPCHK(c, p.shape_list[b].reserve(tiles));
            a.shapes_m = map_args(b); a.shapes_h = shape_args(b); a.nb_shapes = (P + block / 4 - 1) / (block / 4); did_shapes = true;
        }
    }
    with_digest = with_digest && p.n_dig <= p.n_fit + (with_fit ? 1 : 0) + (c->split ? 1 : 0);   // at most one step ahead of the fit
    if (with_digest) {
        const int b = (int)(p.n_dig % kBufs);                              // the next undigested step
        DigestArgs& d = a.digest;
        // (two pipes: two blocks per tile for the CPU rows instead of four - the digest's latency hides behind the other launch, its block slots do not: -3 %)
        static const uint32_t wc_env = tune_env("NHDFIT_WC_PARTS") && atoi(tune_env("NHDFIT_WC_PARTS")) >= 1 ? (uint32_t)atoi(tune_env("NHDFIT_WC_PARTS")) : 0u;   // tuning aid
        const uint32_t wc_parts = wc_env ? wc_env : (c->dual && !c->split && !c->role_kernels) ? 2u : kWcPartsDefault;
        // One block per tile for the signature rows.  Sharing them among up to four blocks with a last-arriver hand-over
        // (NHDFIT_SIG_PARTS=<n> in the tuning build) was measured on config 5's 272 signatures in round 3: the step went from 62
        // to 330 us - the last block derives the X rows from the others' rows past its L1, three dependent 8-byte loads per
        // (class, assignment), and that costs far more than the signature walk it parallelises (profiles/r03)
        static const uint32_t force_sp = tune_env("NHDFIT_SIG_PARTS") ? (uint32_t)atoi(tune_env("NHDFIT_SIG_PARTS")) : 0u;   // tuning aid
        fill_digest_args(c, p, b, wc_parts, force_sp ? std::min(force_sp, 8u) : 1u, d);
        a.nb_digest = tiles * (d.sig_parts + wc_parts);
    }
    uint32_t nb_fit = 0;
    int bf = -1;
    if (with_fit) {
        bf = (int)(p.n_fit % kBufs);
        if (c->want_bitmap) HIPCHK(c, p.nm.reserve((size_t)tiles * chunks * 64));
        fill_fit_args(c, p, bf, now, a.fit, !c->x_spill && !c->role_kernels && !c->split);   // (pair tables: the fused launch only)
        nb_fit = c->n_items;
    }
    a.nb_fit = nb_fit;
    const uint32_t grid = a.nb_choose + a.nb_shapes + a.nb_finish + a.nb_digest + nb_fit;
    if (!grid) return NHDFIT_OK;
    // dynamic LDS of the launch: the largest need among the roles present
    size_t lds = nb_fit ? lds_slice(c->lds_bytes) + (size_t)nw * 64 * sizeof(unsigned long long) : 0;
    if (a.nb_digest && kDigestLds > lds) lds = kDigestLds;
    const size_t map_lds = big ? map_lds_bytes<512>() : map_lds_bytes<256>();
    if ((a.nb_shapes || a.nb_finish) && map_lds > lds) lds = map_lds;

    // HIP-event timing is sampled (every 8th fit launch; every digest-only launch)
    if (with_fit && (int64_t)p.n_fit == c->role_step) {
        HIPCHK(c, c->role_clock.reserve(32));
        unsigned long long init[32] = {0};
        for (int k = 0; k < 5; ++k) { init[2 * k] = ~0ull; init[2 * k + 1] = 0; }
        HIPCHK(c, hipMemcpyAsync(c->role_clock.p, init, sizeof init, hipMemcpyHostToDevice, p.stream));
        HIPCHK(c, wait_stream(p.stream));
        a.role_clock = c->role_clock.p;
        a.fit.clk = c->role_clock.p + 16;                  // per-block phases of the fit role (FitArgs::clk)
    }
    const bool timed = (with_fit && (p.n_fit < 2 || (p.n_fit & 7) == 0)) || (!with_fit && with_digest);
    if (timed && c->ev_pending == kEventRing) { int rc = drain_events(c); if (rc) return rc; }
    static const bool enq_prof = tune_env("NHDFIT_ENQ_PROF") != nullptr;
    const auto t_ev0 = enq_prof ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    if (timed) HIPCHK(c, hipEventRecord(c->ev[c->ev_pending][0], p.stream));
    const auto t_l0 = enq_prof ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    if (enq_prof) c->enq_events_us += std::chrono::duration<double, std::micro>(t_l0 - t_ev0).count();
    if (c->x_spill) {               // more node classes than LDS rows: the variant whose fit role reads the rest from global memory
        if (big) LAUNCH(c, (k_step<512, true>), dim3(grid), dim3(512), lds, p.stream, a);
        else     LAUNCH(c, (k_step<256, true>), dim3(grid), dim3(256), lds, p.stream, a);
    } else
#ifdef NHDFIT_TUNING         // (role_kernels / split are switched by the tuning build's environment only: never set in libnhdfit.so)
    if (c->role_kernels) {
        const uint32_t nb[5] = {a.nb_choose, a.nb_shapes, a.nb_finish, a.nb_digest, nb_fit};
        if (nb[0]) LAUNCH(c, (k_role<512, 0>), dim3(nb[0]), dim3(512), 0, p.stream, a);
        if (nb[1]) LAUNCH(c, (k_role<512, 1>), dim3(nb[1]), dim3(512), map_lds_bytes<512>(), p.stream, a);
        if (nb[2]) LAUNCH(c, (k_role<512, 2>), dim3(nb[2]), dim3(512), map_lds_bytes<512>(), p.stream, a);
        if (nb[3]) LAUNCH(c, (k_role<512, 3>), dim3(nb[3]), dim3(512), kDigestLds, p.stream, a);
        if (nb[4]) LAUNCH(c, (k_role<512, 4>), dim3(nb[4]), dim3(512), lds, p.stream, a);
    } else
    if (grid == nb_fit && c->split) {
        if (big) LAUNCH(c, (k_fit_only<512>), dim3(grid), dim3(512), lds, p.stream, a.fit);
        else     LAUNCH(c, (k_fit_only<256>), dim3(grid), dim3(256), lds, p.stream, a.fit);
    } else
#endif
    if (big) LAUNCH(c, (k_step<512>), dim3(grid), dim3(512), lds, p.stream, a);
    else     LAUNCH(c, (k_step<256>), dim3(grid), dim3(256), lds, p.stream, a);
    HIPCHK(c, hipGetLastError());
    if (enq_prof) c->enq_launch_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_l0).count();
    if (with_fit && c->n_wide) {
        // the general path for the nodes beyond the fast layout: their verdict bits and scores join this step's (same stream,
        // behind the fit role; in front of the all-reduce of a sharded run and of every mapping phase)
        WideArgs wa;
        memset(&wa, 0, sizeof wa);
        wa.wide = c->wide.p; wa.n_wide = c->n_wide; wa.reqs = c->reqs.p; wa.P = P; wa.caps = c->caps.p; wa.busy_from = busy_threshold(now);
        wa.cand = c->use_cand ? c->cand.p : nullptr;
        wa.nm = c->want_bitmap ? reinterpret_cast<unsigned long long*>(p.nm.p) : nullptr; wa.chunks = chunks;
        wa.score = p.score[bf].p; wa.global_base = c->global_base; wa.share = c->sharing ? c->wide_share.p : nullptr;
        const uint64_t pairs = (uint64_t)c->n_wide * P;
        LAUNCH(c, k_wide_eval, dim3((uint32_t)((pairs + 255) / 256)), dim3(256), 0, p.stream, wa);
        HIPCHK(c, hipGetLastError());
    }
    if (timed) {
        const auto t_e1 = enq_prof ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        HIPCHK(c, hipEventRecord(c->ev[c->ev_pending][1], p.stream));
        c->ev_kind[c->ev_pending++] = with_fit ? 0 : 1;
        if (enq_prof) c->enq_events_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_e1).count();
    }
    if (a.role_clock) {
        unsigned long long t[32];
        HIPCHK(c, wait_stream(p.stream));
        HIPCHK(c, hipMemcpy(t, c->role_clock.p, sizeof t, hipMemcpyDeviceToHost));
        if (t[24]) {
            const double nb = (double)t[24];
            fprintf(stderr, "[nhdfit] step %lld fit blocks (%llu): mean / latest after the block's start - staged %.2f / %.2f, pair table %.2f / %.2f, sweep done %.2f / %.2f, "
                            "scores out %.2f / %.2f us\n", (long long)c->role_step, t[24], t[16] * 0.01 / nb, t[17] * 0.01, t[18] * 0.01 / nb, t[19] * 0.01,
                    t[20] * 0.01 / nb, t[21] * 0.01, t[22] * 0.01 / nb, t[23] * 0.01);
        }
        unsigned long long first = ~0ull;
        for (int k = 0; k < 5; ++k) first = t[2 * k] < first ? t[2 * k] : first;
        static const char* names[5] = {"choose", "shapes", "finish", "digest", "fit"};
        for (int k = 0; k < 5; ++k)
            if (t[2 * k + 1]) fprintf(stderr, "[nhdfit] step %lld role %-6s: first block starts +%.2f us, last block ends +%.2f us\n",
                                      (long long)c->role_step, names[k], (t[2 * k] - first) * 0.01, (t[2 * k + 1] - first) * 0.01);
    }
    p.n_finished += did_finish; p.n_chosen += did_choose; p.n_shaped += did_shapes;
    if (with_digest) p.n_dig++;
    if (with_fit) {
        hipStream_t after = p.stream;           // where the scores of this step become final
        if (c->comm) {      // one communicator -> its collectives stay on one stream (s_red), in step order
            HIPCHK(c, hipEventRecord(p.ev_fit[bf], p.stream));
            HIPCHK(c, hipStreamWaitEvent(c->s_red, p.ev_fit[bf], 0));
            ncclResult_t r = (c->known_idle = false, g_rccl).AllReduce(p.score[bf].p, p.score[bf].p, P, ncclUint64, ncclMax, c->comm, c->s_red);
            if (r != ncclSuccess) return fail(c, NHDFIT_E_RCCL, "ncclAllReduce: %s", g_rccl.GetErrorString(r));
            after = c->s_red;
        }
        if (c->want_map && c->n_big_pods) {      // pods with 4 proc groups: generic set model (scratch-heavy, kept out of k_step)
            const dim3 mg((P + kMapWaves - 1) / kMapWaves), mb(64 * kMapWaves);
            LAUNCH(c, k_map<true>, mg, mb, 0, after, map_args(bf));
            HIPCHK(c, hipGetLastError());
        }
        if (c->comm) HIPCHK(c, hipEventRecord(p.ev_red[bf], c->s_red));
        p.n_fit++;
        // no mapping roles for this step (output switched off, or only 4-group pods): nothing to catch up on later
        if (!small_map) p.n_shaped = p.n_chosen = p.n_finished = p.n_fit;
        c->stats.evals_last = (uint64_t)P * c->n;
        // algorithmic bytes of the step, SURVEY.md section 8(d): ceil(P/T) * N * B_node + P * B_req + P * N / 8 + 8 * P
        // with T = 64 pods per tile, B_node = 24 (the 16-byte node record + the 8-byte busy time every tile streams),
        // B_req = 128; the P * N / 8 term is the node-major verdict matrix (dropped when that output is switched off)
        c->stats.bytes_last = (uint64_t)tiles * c->n * 24ull + (uint64_t)P * sizeof(nhdfit_req) +
                              (c->want_bitmap ? (uint64_t)tiles * chunks * 64ull * 8ull : 0ull) + (uint64_t)P * 8ull;
        c->stats.nodes = c->n; c->stats.nsig = c->nsig; c->stats.ncls = c->ncls; c->stats.lds_bytes = c->lds_bytes;
        c->stats.pipes = c->dual && !c->split && !c->role_kernels ? (uint32_t)c->npipes : 1u;
    }
    return NHDFIT_OK;
}

// pod-major rows [chunks][P] of the last step's verdict matrix (stream-ordered after the step that produced it)
int convert_rows(nhdfit_ctx* c, Pipe& p) {
    const uint32_t chunks = (c->n + 63) / 64, tiles = (c->P + kTile - 1) / kTile;
    HIPCHK(c, c->bitmap.reserve((size_t)chunks * c->P));
    LAUNCH(c, k_rows, dim3((tiles * chunks + 3) / 4), dim3(256), 0, p.stream, p.nm.p, c->bitmap.p, chunks, c->P);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, wait_stream(p.stream));
    return NHDFIT_OK;
}

// the sequential kernels' copy, [P][chunks]; stream order is all they need
int convert_rows_t(nhdfit_ctx* c, Pipe& p) {
    const uint32_t chunks = (c->n + 63) / 64, tiles = (c->P + kTile - 1) / kTile;
    const uint32_t groups = (chunks + kRowsTChunks - 1) / kRowsTChunks;
    HIPCHK(c, c->rows_t.reserve((size_t)chunks * c->P));
    LAUNCH(c, k_rows_t, dim3((tiles * groups + 3) / 4), dim3(256), 0, p.stream, p.nm.p, c->rows_t.p, chunks, c->P);
    HIPCHK(c, hipGetLastError());
    return NHDFIT_OK;
}

int flush_pipeline(nhdfit_ctx* c) {
    if (!c->P || !c->want_map || c->n_big_pods >= c->P) return NHDFIT_OK;
    static const bool role_drain = tune_env("NHDFIT_ROLE_DRAIN") != nullptr;   // tuning aid: drain with role launches, as the steps ran
    const uint32_t tiles = (c->P + kTile - 1) / kTile;
    for (Pipe& p : c->pipe) {
        if (role_drain || c->role_kernels || c->split) {
            while (p.n_finished < p.n_fit) {
                int rc = launch_step(c, p, false, false, 0.0, true);
                if (rc) return rc;
            }
            continue;
        }
        // the steps whose mapping phases have not all run: mapped from their scores in one launch (k_map_tiles, step_map.h)
        while (p.n_finished < p.n_fit) {
            DrainArgs a;
            memset(&a, 0, sizeof a);
            a.tiles = tiles;
            a.h = make_shape_args(c, p, 0);
            while (a.nsteps < (uint32_t)kDrainSteps && p.n_finished + a.nsteps < p.n_fit) {
                const int b = (int)((p.n_finished + a.nsteps) % kBufs);
                if (c->comm) HIPCHK(c, hipStreamWaitEvent(p.stream, p.ev_red[b], 0));      // the step's scores are final behind its all-reduce
                a.m[a.nsteps++] = make_map_args(c, p, b);
            }
            c->side_streams_used = true;
            const bool drain_prof = tune_env("NHDFIT_DRAIN_PROF") != nullptr;      // tuning aid: where a drain launch's time goes
            if (drain_prof) {
                HIPCHK(c, c->role_clock.reserve(16));
                unsigned long long init[16] = {0};
                init[7] = ~0ull;
                HIPCHK(c, hipMemcpyAsync(c->role_clock.p, init, sizeof init, hipMemcpyHostToDevice, p.stream));
                a.clk = c->role_clock.p;
            }
            LAUNCH(c, k_map_tiles, dim3(a.nsteps * tiles), dim3(256), map_tile_lds_bytes<256>(), p.stream, a);
            HIPCHK(c, hipGetLastError());
            if (drain_prof) {
                unsigned long long t[16];
                HIPCHK(c, wait_stream(p.stream));
                HIPCHK(c, hipMemcpy(t, c->role_clock.p, sizeof t, hipMemcpyDeviceToHost));
                fprintf(stderr, "[nhdfit] drain of %u step(s) x %u tiles: staged +%.2f, NIC bits / masks / key +%.2f, shapes de-duplicated +%.2f, state machine +%.2f, "
                                "generic shapes +%.2f, finish +%.2f us after a block's start (latest block each); first block start to last block end %.2f us\n",
                        a.nsteps, tiles, t[0] * 0.01, t[1] * 0.01, t[2] * 0.01, t[3] * 0.01, t[4] * 0.01, t[5] * 0.01, (t[6] - t[7]) * 0.01);
            }
            p.n_finished += a.nsteps;
        }
        p.n_shaped = p.n_chosen = p.n_finished;
    }
    return NHDFIT_OK;
}

}  // namespace

static int enqueue_step(nhdfit_ctx* c, double now);
int nhdfit_enqueue_step(nhdfit_ctx* c, double now) {
    if (!c) return NHDFIT_E_INVAL;
    static const bool enq_prof = tune_env("NHDFIT_ENQ_PROF") != nullptr;
    if (!enq_prof) return enqueue_step(c, now);
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = enqueue_step(c, now);
    c->enq_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (++c->enq_n % 1000 == 0) {
        fprintf(stderr, "[nhdfit] 1000 enqueues: %.2f us each on the host, %.2f in the launch calls, %.2f in event records\n", c->enq_us / 1000, c->enq_launch_us / 1000, c->enq_events_us / 1000);
        c->enq_us = c->enq_launch_us = c->enq_events_us = 0;
    }
    return rc;
}
static int enqueue_step(nhdfit_ctx* c, double now) {
    if (!c->P) return fail(c, NHDFIT_E_STATE, "stage requests first");
    if (!c->n) return fail(c, NHDFIT_E_STATE, "no nodes uploaded");
    HIPCHK(c, hipSetDevice(c->dev));
    if (c->n_enq == 0) {
        // 512-thread blocks (8 waves; 3 co-resident blocks per CU at 70 VGPRs: one block's LDS fill overlaps the
        // others' sweep), 256-thread blocks for small problems so that the grid still covers the chip
        const uint32_t tiles = (c->P + kTile - 1) / kTile, chunks = (c->n + 63) / 64;
        c->geom_big = (uint64_t)tiles * ((chunks + 31) / 32) >= (uint32_t)c->prop.multiProcessorCount;
        if (const char* b = tune_env("NHDFIT_BLOCK")) c->geom_big = atoi(b) >= 512;      // tuning aid
        if (c->role_kernels) c->geom_big = true;
        // How many launches in flight (profiles/r04/pipes_*.log).  A third one covers long latency chains inside a launch: with a
        // large dictionary the digest role is one (config 5, 151 signatures: 15.9 -> 10.9 us per step of 2 048 pods).  Where the
        // fit role fills the launch (config 4) it buys 3 % in steady state and costs as much in a short run, whose end waits for
        // every launch in flight (20 steps: 19.1 -> 19.7 us each); problems too small to fill the chip are bound by the host's
        // ~11 us per enqueue whatever the count (config 2, config 3: flat from two to four).
        static const int force_pipes = tune_env("NHDFIT_PIPES") ? atoi(tune_env("NHDFIT_PIPES")) : 0;   // tuning aid
        c->npipes = force_pipes >= 1 && force_pipes <= kPipes ? force_pipes : (c->geom_big && c->nsig > 64) ? 3 : 2;
    }
    // step k of a staged batch runs on pipe k % 2 (sharded runs too: the all-reduces of both pipes go to the one reduce
    // stream in step order, the same order on every rank); the profiling forms stay on pipe 0
    const int which = c->dual && !c->split && !c->role_kernels ? (int)(c->n_enq % (uint64_t)c->npipes) : 0;
    Pipe& p = c->pipe[which];
    c->n_enq++;
    c->last_pipe = which;
    // the mirror may have changed since the last step (uploads, commits, deltas between two steps of one staged batch): node
    // records first - new node classes put the digests that ran ahead back (ensure_records), and they are redone below
    { int rc_ = ensure_records(c); if (rc_) return rc_; }
    if (p.n_dig <= p.n_fit) {                        // this pipe's first step after staging (or after such a change): its digest has not run yet
        int rc = launch_step(c, p, false, true, now, false);
        if (rc) return rc;
    }
    if (c->split) {                                  // profiling aid: side roles and fit role as two launches
        int rc = launch_step(c, p, false, true, now, false);
        if (rc) return rc;
        return launch_step(c, p, true, false, now, false);
    }
    return launch_step(c, p, true, true, now, false);
}

int nhdfit_sync(nhdfit_ctx* c) {
    if (!c) return NHDFIT_E_INVAL;
    HIPCHK(c, hipSetDevice(c->dev));
    // (the HIP events of the sampled launches are read where the statistics are asked for, nhdfit_get_stats - a few
    // hipEventElapsedTime calls are ~10 us, half a microsecond per step of a 20-step region, and nobody waiting here wants them)
    return sync_all(c);
}

int nhdfit_fetch(nhdfit_ctx* c, uint64_t* score_out, uint64_t* bitmap_out, nhdfit_mapping* map_out) {
    if (!c) return NHDFIT_E_INVAL;
    Pipe& p = c->pipe[c->last_pipe];
    if (!c->P || !p.n_fit) return fail(c, NHDFIT_E_STATE, "nothing staged / no step enqueued");
    c->side_streams_used = true;                                // (the copies below ride the last step's pipe)
    HIPCHK(c, hipSetDevice(c->dev));
    if (map_out && !c->want_map) return fail(c, NHDFIT_E_STATE, "mapping output is disabled");
    const uint32_t P = c->P;
    const int b = (int)((p.n_fit - 1) % kBufs);               // results of the most recent step
    // the launches that finish the mappings still in flight, the copies behind them on the same stream, ONE wait
    { int rc_ = flush_pipeline(c); if (rc_) return rc_; }
    if (c->comm) HIPCHK(c, wait_stream(c->s_red));
    if (map_out && c->n_wide) {
        // winners that are wide nodes: their mappings from the general set model, over what the mapping roles left for them
        constexpr uint32_t kWideMapThreads = 512;
        HIPCHK(c, c->wide_scratch.reserve((size_t)kWideMapThreads * kWideScratchWords));
        HIPCHK(c, c->wide_flags.reserve(4));
        HIPCHK(c, hipMemsetAsync(c->wide_flags.p, 0, 4 * sizeof(uint32_t), p.stream));
        WideMapArgs wm;
        memset(&wm, 0, sizeof wm);
        wm.wide = c->wide.p; wm.n_wide = c->n_wide; wm.reqs = c->reqs.p; wm.P = P; wm.caps = c->caps.p;
        wm.score = p.score[b].p; wm.global_base = c->global_base; wm.n = c->n; wm.out = p.maps[b].p;
        wm.scratch = c->wide_scratch.p; wm.flags = c->wide_flags.p; wm.share = c->sharing ? c->wide_share.p : nullptr;
        LAUNCH(c, k_wide_map, dim3(kWideMapThreads), dim3(64), 0, p.stream, wm);
        HIPCHK(c, hipGetLastError());
        uint32_t fl[4] = {0, 0, 0, 0};
        HIPCHK(c, hipMemcpyAsync(fl, c->wide_flags.p, sizeof fl, hipMemcpyDeviceToHost, p.stream));
        HIPCHK(c, wait_stream(p.stream));
        if (fl[0]) return fail(c, NHDFIT_E_LIMIT, "the set model of a wide node's mapping outgrew its table");
    }
    if (score_out) {
        HIPCHK(c, c->pin_score.reserve(P));
        HIPCHK(c, hipMemcpyAsync(c->pin_score.p, p.score[b].p, (size_t)P * 8, hipMemcpyDeviceToHost, p.stream));
    }
    if (map_out) {
        HIPCHK(c, c->pin_maps.reserve(P));
        HIPCHK(c, hipMemcpyAsync(c->pin_maps.p, p.maps[b].p, (size_t)P * sizeof(nhdfit_mapping), hipMemcpyDeviceToHost, p.stream));
    }
    int rc = nhdfit_sync(c);
    if (rc) return rc;
    if (score_out)
        for (uint32_t i = 0; i < P; ++i) score_out[c->perm[i]] = c->pin_score.p[i];
    if (map_out)
        for (uint32_t i = 0; i < P; ++i) map_out[c->perm[i]] = c->pin_maps.p[i];
    if (bitmap_out) {
        if (!c->want_bitmap) return fail(c, NHDFIT_E_STATE, "bitmap output is disabled");
        const size_t chunks = (c->n + 63) / 64;
        { int rc_ = convert_rows(c, p); if (rc_) return rc_; }
        std::vector<uint64_t> tmp(chunks * P);
        HIPCHK(c, hipMemcpy(tmp.data(), c->bitmap.p, chunks * P * 8, hipMemcpyDeviceToHost));
        for (size_t ch = 0; ch < chunks; ++ch)
            for (uint32_t i = 0; i < P; ++i) bitmap_out[ch * P + c->perm[i]] = tmp[ch * P + i];
    }
    return NHDFIT_OK;
}

namespace {
// nhdfit_find for at most one pod tile and no verdict matrix - the scheduler's pod-at-a-time FindNode - as ONE launch (k_find,
// step_kernel.h): the requests are read from, and the results stored into, a fine-grained host block; the host polls the
// sequence word the launch stores last.  Returns 1 when the call is not eligible (or the launch gave up): the caller then
// takes the staged path, which also words the errors; 0 on success; < 0 on a HIP error.
int find_small(nhdfit_ctx* c, const nhdfit_req* reqs, uint32_t P, double now, const uint64_t* cand, uint64_t* score_out, nhdfit_mapping* map_out) {
    if (!c->fast_find || !reqs || !P || P > (uint32_t)kTile || !c->nsig || !c->n || !c->find_host || c->n_wide) return 1;   // (wide nodes: the staged path carries the general pass)
    const auto t0 = std::chrono::steady_clock::now();
    if (map_out && !c->want_map) return 1;
    int32_t hp_max = 0;
    uint32_t wcls = 0;
    for (uint32_t p = 0; p < P; ++p) {
        if (reqs[p].hugepages_gb < 0) return 1;
        hp_max = reqs[p].hugepages_gb > hp_max ? reqs[p].hugepages_gb : hp_max;
        if (!req_valid(reqs[p])) continue;
        if (reqs[p].n_groups > 3) return 1;                     // the generic set model is a kernel of its own
        wcls = std::max(wcls, wclass_of(reqs[p].n_groups));
    }
    if (hp_max > kMaxHpRows - 2) return 1;
    HIPCHK(c, hipSetDevice(c->dev));
    if (c->P) {                                                 // a staged batch: its steps may be in flight on either pipe
        { int rc_ = sync_all(c); if (rc_) return rc_; }
        { int rc_ = drain_events(c); if (rc_) return rc_; }
    }
    Pipe& p = c->pipe[0];
    for (Pipe& q : c->pipe) q.n_dig = q.n_fit = q.n_shaped = q.n_chosen = q.n_finished = 0;
    c->n_enq = 0; c->last_pipe = 0; c->n_items = 0; c->n_big_pods = 0;
    // one pod: no table image at all (k_find1) when the dictionary's 16-bit stream and its signature count fit the block's LDS
    const bool lone = P == 1 && c->lone_pod && c->flat_words && c->flat_words <= kDictLdsWords && c->nsig <= kLoneMaxSigs;
    c->P = P;                                                   // (for the layout / argument helpers; nothing stays staged: reset below)
    c->hp_rows = (uint32_t)hp_max + 2;
    c->max_wcls = wcls;
    int rc = lone ? NHDFIT_OK : refresh_layouts(c);
    if (!rc && !lone) rc = ensure_records(c);
    if (rc || (!lone && c->x_spill)) { c->P = 0; return rc ? rc : 1; }
    hipError_t e = lone ? hipSuccess : p.hdr[0].reserve(kTile);
    if (e == hipSuccess) e = p.score[0].reserve(kTile);
    const uint32_t chunks = (c->n + 63) / 64;
    c->use_cand = cand != nullptr;
    if (e == hipSuccess && cand && (c->cand_shadow.size() != chunks || c->cand.cap < chunks ||
                                    memcmp(c->cand_shadow.data(), cand, (size_t)chunks * 8) != 0)) {
        // (consecutive pods of one node group come with the same mask: it is uploaded when it changes)
        c->cand_shadow.clear();
        e = c->cand.reserve(chunks);
        if (e == hipSuccess) e = hipMemcpyAsync(c->cand.p, cand, (size_t)chunks * 8, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) c->cand_shadow.assign(cand, cand + chunks);
    }
    if (e != hipSuccess) { c->P = 0; return fail(c, NHDFIT_E_HIP, "small find: %s", hipGetErrorString(e)); }

    FindHost* h = c->find_host;
    memcpy(h->reqs, reqs, (size_t)P * sizeof *reqs);
    uint32_t seq = ++c->find_seq;
    if (seq == 0u || seq == kFindAborted) seq = c->find_seq = 1u;
    Find1Args a1;
    memset(&a1, 0, sizeof a1);
    if (lone) {
        a1.m = make_map_args(c, p, 0);
        a1.m.reqs = h->reqs; a1.m.tile_wcls = nullptr; a1.m.tabs = nullptr; a1.m.out = h->maps;
        a1.m.score = reinterpret_cast<const unsigned long long*>(c->find_sync.p + 4);
        a1.h = make_shape_args(c, p, 0);
        a1.p4 = c->p4.p;
        a1.d = DictView{c->caps.p, c->ncls, c->group_sets.p, SigDict{c->sig_off.p, c->pool_off.p, c->pool_glimit.p, c->cc.p, c->nsig}, c->sig_flat.p, c->flat_words, nullptr, 0};
        a1.nsig = c->nsig; a1.fc_dim = c->max_cores + 1; a1.fg_dim = c->max_gpus + 1; a1.ngs = c->ngs;
        a1.chunks = chunks;
        static const uint32_t lone_nb = tune_env("NHDFIT_FIND_BLOCKS") ? (uint32_t)atoi(tune_env("NHDFIT_FIND_BLOCKS")) : 0u;   // tuning aid
        a1.nb = std::max(1u, std::min(chunks, lone_nb ? lone_nb : std::min((chunks + 7) / 8, (uint32_t)c->prop.multiProcessorCount)));   // two chunks per wavefront
        a1.busy_from = busy_threshold(now);
        a1.cand = c->use_cand ? c->cand.p : nullptr;
        a1.sync = c->find_sync.p; a1.host = h; a1.seq = seq; a1.want_map = map_out ? 1u : 0u;
    }
    FindArgs a;
    memset(&a, 0, sizeof a);
    a.s.shapes_P = P;
    static const uint32_t find_wc = tune_env("NHDFIT_FIND_WC_PARTS") && atoi(tune_env("NHDFIT_FIND_WC_PARTS")) >= 1 ? (uint32_t)atoi(tune_env("NHDFIT_FIND_WC_PARTS")) : kWcPartsDefault;   // tuning aid
    fill_digest_args(c, p, 0, find_wc, 1u, a.s.digest);
    a.s.digest.reqs = h->reqs;
    a.s.nb_digest = 1u + find_wc;
    fill_fit_args(c, p, 0, now, a.s.fit);
    a.s.fit.nm = nullptr; a.s.fit.items = nullptr; a.s.fit.dbg_skip = 0;
    constexpr uint32_t nw = 4;                                  // 256-thread blocks: a wavefront per chunk where the cluster is small enough
    // Few, longer fit blocks: every block stages the tile's table rows (tens of KB) before its first chunk and waits for the digest
    // on one counter - measured at 65 536 nodes (profiles/r03): 256 blocks 107 us per call, 64 blocks 54, 32 blocks 50.  Eight
    // chunks per wavefront, at most one block per CU.
    static const uint32_t force_nb = tune_env("NHDFIT_FIND_BLOCKS") ? (uint32_t)atoi(tune_env("NHDFIT_FIND_BLOCKS")) : 0u;   // tuning aid
    a.s.nb_fit = std::max(1u, std::min((chunks + nw - 1) / nw, force_nb ? force_nb : std::min((chunks + 8 * nw - 1) / (8 * nw), (uint32_t)c->prop.multiProcessorCount)));
    a.s.finish_m = make_map_args(c, p, 0);                      // the mapping tail (map_one_tile): requests in, mappings out of the host block
    a.s.finish_m.reqs = h->reqs; a.s.finish_m.tile_wcls = nullptr; a.s.finish_m.out = h->maps;
    a.s.finish_h = make_shape_args(c, p, 0);
    a.wcls = wcls; a.want_map = map_out ? 1u : 0u;
    a.sync = c->find_sync.p; a.host = h; a.seq = seq;
    size_t lds = lds_slice(c->lds_bytes) + (size_t)nw * 64 * sizeof(unsigned long long);
    lds = std::max(lds, std::max(kDigestLds, map_tile_lds_bytes<256>()));
    const bool clocks = kTuning && c->role_step >= 0;           // tuning aid (NHDFIT_ROLE_TIMES): the phases of the launch on the device clock
    if (clocks) {
        HIPCHK(c, c->role_clock.reserve(10));
        unsigned long long init[10];
        for (int k = 0; k < 5; ++k) { init[2 * k] = ~0ull; init[2 * k + 1] = 0; }
        HIPCHK(c, hipMemcpy(c->role_clock.p, init, sizeof init, hipMemcpyHostToDevice));
        a.s.role_clock = c->role_clock.p;
        a1.role_clock = c->role_clock.p;
    }
    const auto t_launch = std::chrono::steady_clock::now();
    c->P = 0;                                                   // nothing is staged for nhdfit_enqueue_step / nhdfit_fetch
    if (lone) LAUNCH(c, (k_find1<256>), dim3(a1.nb), dim3(256), kLoneLds + map_tile_lds_bytes<256>(), c->stream, a1);
    else LAUNCH(c, (k_find<256>), dim3(a.s.nb_digest + a.s.nb_fit), dim3(256), lds, c->stream, a);
    HIPCHK(c, hipGetLastError());
    uint32_t seen = 0;
    for (uint32_t spins = 1;; ++spins) {
        seen = __atomic_load_n(&h->flag, __ATOMIC_ACQUIRE);
        if (seen == seq || seen == kFindAborted) break;
        if ((spins & 255u) == 0 && std::chrono::steady_clock::now() - t_launch > std::chrono::microseconds(500)) {
            HIPCHK(c, wait_stream(c->stream));         // a long launch (or host memory the device does not write through): wait for its end
            seen = __atomic_load_n(&h->flag, __ATOMIC_ACQUIRE);
            break;
        }
        __builtin_ia32_pause();
    }
    if (seen != seq) {                                          // the launch gave up on a wait: counters back to zero, staged path
        HIPCHK(c, wait_stream(c->stream));
        HIPCHK(c, hipMemsetAsync(c->find_sync.p, 0, 8 * sizeof(uint32_t), c->stream));
        h->flag = 0;
        return 1;
    }
    if (clocks) {
        const double us_seen = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_launch).count();
        unsigned long long t[10];
        HIPCHK(c, wait_stream(c->stream));
        HIPCHK(c, hipMemcpy(t, c->role_clock.p, sizeof t, hipMemcpyDeviceToHost));
        unsigned long long first = ~0ull;
        for (int k = 0; k < 5; ++k) first = t[2 * k] < first ? t[2 * k] : first;
        static const char* names[5] = {"choose", "shapes", "finish", "digest", "fit"};
        fprintf(stderr, "[nhdfit] single-launch find%s: %u fit blocks, results seen %.1f us after the launch call began (host prep %.1f us)\n", lone ? " (lone pod, no tables)" : "", lone ? a1.nb : a.s.nb_fit, us_seen,
                std::chrono::duration<double, std::micro>(t_launch - t0).count());
        for (int k = 0; k < 5; ++k)
            if (t[2 * k + 1]) fprintf(stderr, "[nhdfit]   %-6s: +%.2f us .. +%.2f us\n", names[k], (t[2 * k] - first) * 0.01, (t[2 * k + 1] - first) * 0.01);
    }
    if (c->comm) {
        // sharded: every rank ran the launch on its shard and mapped its own winner; one all-reduce(max) of the packed scores
        // (<= 64 words) picks the cluster's winner, and a rank that does not own it drops its mapping - exactly what the
        // staged path returns (its mapping roles skip winners outside the shard).  All on the reduce stream: the launch is
        // over (the host saw its flag), and the communicator's collectives stay on one stream.
        // The words travel in the STAGED order of the call's pods (staged_order): whether a rank takes this form or the staged
        // step is decided by rank-local facts (a wide node in its shard, spilled class rows, a launch that gave up), so the two
        // forms must pair the same pods in the one collective they both issue per call.
        HIPCHK(c, c->find_red.reserve(kTile));
        HIPCHK(c, c->pin_score.reserve(2 * kTile));
        staged_order(reqs, P, c->perm);
        uint64_t* send = c->pin_score.p + kTile;
        for (uint32_t k = 0; k < P; ++k) send[k] = h->score[c->perm[k]];
        HIPCHK(c, hipMemcpyAsync(c->find_red.p, send, (size_t)P * 8, hipMemcpyHostToDevice, c->s_red));
        ncclResult_t r = (c->known_idle = false, g_rccl).AllReduce(c->find_red.p, c->find_red.p, P, ncclUint64, ncclMax, c->comm, c->s_red);
        if (r != ncclSuccess) return fail(c, NHDFIT_E_RCCL, "ncclAllReduce: %s", g_rccl.GetErrorString(r));
        HIPCHK(c, hipMemcpyAsync(c->pin_score.p, c->find_red.p, (size_t)P * 8, hipMemcpyDeviceToHost, c->s_red));
        HIPCHK(c, wait_stream(c->s_red));
        for (uint32_t k = 0; k < P; ++k) {
            const uint32_t i = c->perm[k];
            if (c->pin_score.p[k] != h->score[i]) memset(&h->maps[i], 0, sizeof(nhdfit_mapping));
            h->score[i] = c->pin_score.p[k];
        }
    }
    if (score_out) memcpy(score_out, h->score, (size_t)P * 8);
    if (map_out) memcpy(map_out, h->maps, (size_t)P * sizeof(nhdfit_mapping));
    c->stats.evals_last = (uint64_t)P * c->n;
    c->stats.bytes_last = (uint64_t)c->n * 24ull + (uint64_t)P * sizeof(nhdfit_req) + (uint64_t)P * 8ull;
    c->stats.nodes = c->n; c->stats.nsig = c->nsig; c->stats.ncls = c->ncls; c->stats.lds_bytes = c->lds_bytes;
    c->stats.small_finds++;
    return NHDFIT_OK;
}

// nhdfit_find for more than one pod tile as ONE launch (k_findn, step_kernel.h) behind the staging of the batch: the requests go to
// the device as the staged path sends them (sorted into tiles, one async copy), digest -> fit -> mapping run tile by tile inside the
// launch, scores and mappings arrive in a fine-grained host block, the host polls the sequence word stored last.
// Returns 0 = done (nothing stays staged), 1 = not eligible, nothing touched; 2 = the batch is STAGED but the launch did not run or
// gave up: the caller goes on with nhdfit_enqueue_step / nhdfit_fetch; < 0 = error (worded).
int find_batch(nhdfit_ctx* c, const nhdfit_req* reqs, uint32_t P, double now, const uint64_t* cand, uint64_t* score_out, nhdfit_mapping* map_out) {
    if (!c->batch_find || !c->fast_find || !reqs || P <= (uint32_t)kTile || !c->nsig || !c->n || c->n_wide || c->comm || c->role_kernels || c->split) return 1;
    if (map_out && !c->want_map) return 1;
    static const bool prof = tune_env("NHDFIT_FIND_PROF") != nullptr;      // tuning aid: host-side phase times of the call
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!prof) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[nhdfit] batch find P=%u %s %.1f us\n", P, what, std::chrono::duration<double, std::micro>(t1 - t_prev).count());
        t_prev = t1;
    };
    // sorted into tiles in the page-locked block.  What the launch reads once per block - tile classes, work items - it reads there; the
    // request records of a FEW tiles too (five digest blocks and the mapping read a tile's 8 KB over the link: cheaper than a copy
    // command's ~10 us up to a few hundred pods, dearer than the copy beyond)
    const bool host_reqs = P <= 512u;
    const uint32_t tiles_ = (P + kTile - 1) / kTile;
    // (a larger batch: the records, and behind them in the same block the work items, their per-tile counts and the tile classes - ONE copy command)
    const size_t tail_cap = host_reqs ? 0 : ((size_t)tiles_ * 32 + 4096) * sizeof(FitItem) + (size_t)tiles_ * 8 + 64;
    int rc = stage_requests(c, reqs, P, true, true, tail_cap);
    if (rc) return rc;
    lap("stage");
    auto to_steps = [&]() { const int r2 = finish_deferred_copies(c); return r2 ? r2 : 2; };
    if (c->n_big_pods) return to_steps();                       // four-group pods: their set model is a kernel of its own
    if (cand && (rc = stage_cand(c, cand))) return rc;
    if ((rc = ensure_records(c))) return rc;
    if (c->x_spill) return to_steps();
    constexpr uint32_t nw = 4;                                  // 256-thread blocks: a tile's pods are one wavefront of the mapping tail
    if ((rc = build_items(c, nw, true))) return rc;
    const uint32_t tiles = (P + kTile - 1) / kTile, chunks = (c->n + 63) / 64;
    Pipe& p = c->pipe[0];
    // the host block: flag word (16 bytes), P score words, P mappings
    if (P > c->findn_cap) {
        if (c->findn_host) (void)hipHostFree(c->findn_host);
        c->findn_host = nullptr; c->findn_cap = 0;
        const size_t cap = std::max<size_t>(4096, (size_t)P + P / 2);
        HIPCHK(c, hipHostMalloc((void**)&c->findn_host, 16 + cap * (8 + sizeof(nhdfit_mapping)), hipHostMallocCoherent));
        memset(c->findn_host, 0, 16);
        c->findn_cap = cap;
    }
    uint32_t* h_flag = reinterpret_cast<uint32_t*>(c->findn_host);
    unsigned long long* h_score = reinterpret_cast<unsigned long long*>(c->findn_host + 16);
    nhdfit_mapping* h_maps = reinterpret_cast<nhdfit_mapping*>(c->findn_host + 16 + c->findn_cap * 8);
    const uint32_t sync_words = 2u + 2u * tiles;
    if (sync_words > c->findn_sync_words) {
        const uint32_t words = std::max(sync_words, 2u + 2u * 256u);
        HIPCHK(c, c->findn_sync.reserve(words));
        HIPCHK(c, hipMemsetAsync(c->findn_sync.p, 0, (size_t)words * sizeof(uint32_t), c->stream));
        c->findn_sync_words = words;
    }
    uint32_t seq = ++c->find_seq;
    if (seq == 0u || seq == kFindAborted) seq = c->find_seq = 1u;

    FindNArgs a;
    memset(&a, 0, sizeof a);
    a.s.shapes_P = P;
    static const uint32_t wc_env = tune_env("NHDFIT_FIND_WC_PARTS") && atoi(tune_env("NHDFIT_FIND_WC_PARTS")) >= 1 ? (uint32_t)atoi(tune_env("NHDFIT_FIND_WC_PARTS")) : 0u;   // tuning aid
    const uint32_t wc_parts = wc_env ? wc_env : kWcPartsDefault;   // (the digest is on the call's critical path here: the CPU rows cut four ways, as the one-tile find cuts them)
    fill_digest_args(c, p, 0, wc_parts, 1u, a.s.digest);
    a.dig_parts = 1u + wc_parts;
    a.s.nb_digest = tiles * a.dig_parts;
    a.nb_lead = (a.s.nb_digest + 7u) & ~7u;
    fill_fit_args(c, p, 0, now, a.s.fit, true);
    a.s.fit.nm = nullptr; a.s.fit.dbg_skip = 0;                 // (no verdict matrix in this form)
    a.s.nb_fit = c->n_items;
    a.s.finish_m = make_map_args(c, p, 0);
    a.s.finish_m.out = h_maps;
    a.s.finish_h = make_shape_args(c, p, 0);
    a.want_map = map_out ? 1u : 0u; a.tiles = tiles;
    const size_t items_bytes = (size_t)c->n_items * sizeof(FitItem), counts_bytes = (size_t)tiles * sizeof(uint32_t);
    if (host_reqs || items_bytes + counts_bytes + tiles > tail_cap) {
        // everything where the staging left it: the launch reads the page-locked block (few tiles: a few dozen reads over the link)
        if (!host_reqs) { HIPCHK(c, hipMemcpyAsync(c->reqs.p, c->pin_reqs.p, (size_t)P * sizeof(nhdfit_req), hipMemcpyHostToDevice, c->stream)); c->reqs_deferred = false; }
        else { a.s.digest.reqs = c->pin_reqs.p; a.s.finish_m.reqs = c->pin_reqs.p; }
        a.s.fit.items = reinterpret_cast<const FitItem*>(c->pin_items.p);
        a.tile_items = reinterpret_cast<const uint32_t*>(c->pin_items.p + items_bytes);
        a.tile_wcls = c->pin_wcls.p;
    } else {
        uint8_t* tail_h = reinterpret_cast<uint8_t*>(c->pin_reqs.p + P);
        const uint8_t* tail_d = reinterpret_cast<const uint8_t*>(c->reqs.p + P);
        memcpy(tail_h, c->pin_items.p, items_bytes + counts_bytes);
        memcpy(tail_h + items_bytes + counts_bytes, c->pin_wcls.p, tiles);
        HIPCHK(c, hipMemcpyAsync(c->reqs.p, c->pin_reqs.p, (size_t)P * sizeof(nhdfit_req) + items_bytes + counts_bytes + tiles, hipMemcpyHostToDevice, c->stream));
        c->reqs_deferred = false;
        a.s.fit.items = reinterpret_cast<const FitItem*>(tail_d);
        a.tile_items = reinterpret_cast<const uint32_t*>(tail_d + items_bytes);
        a.tile_wcls = tail_d + items_bytes + counts_bytes;
    }
    a.sync = c->findn_sync.p; a.host_score = h_score; a.host_flag = h_flag; a.seq = seq;
    size_t lds = lds_slice(c->lds_bytes) + (size_t)nw * 64 * sizeof(unsigned long long);
    lds = std::max(lds, std::max(kDigestLds, map_tile_lds_bytes<256>()));
    lap("records, items, arguments");
    const auto t_launch = std::chrono::steady_clock::now();
    LAUNCH(c, (k_findn<256>), dim3(a.nb_lead + a.s.nb_fit), dim3(256), lds, c->stream, a);
    HIPCHK(c, hipGetLastError());
    lap("launch call");
    uint32_t seen = 0;
    for (uint32_t spins = 1;; ++spins) {
        seen = __atomic_load_n(h_flag, __ATOMIC_ACQUIRE);
        if (seen == seq || seen == kFindAborted) break;
        if ((spins & 255u) == 0 && std::chrono::steady_clock::now() - t_launch > std::chrono::microseconds(2000)) {
            HIPCHK(c, wait_stream(c->stream));                  // a long launch (or host memory the device does not write through): wait for its end
            seen = __atomic_load_n(h_flag, __ATOMIC_ACQUIRE);
            break;
        }
        __builtin_ia32_pause();
    }
    if (seen != seq) {                                          // the launch gave up on a wait: counters back to zero, staged path
        HIPCHK(c, wait_stream(c->stream));
        HIPCHK(c, hipMemsetAsync(c->findn_sync.p, 0, (size_t)c->findn_sync_words * sizeof(uint32_t), c->stream));
        *h_flag = 0;
        c->n_items = 0;                                         // (the step's own work items: 512-thread blocks)
        return to_steps();
    }
    lap("poll");
    if (score_out) for (uint32_t i = 0; i < P; ++i) score_out[c->perm[i]] = h_score[i];
    if (map_out) for (uint32_t i = 0; i < P; ++i) map_out[c->perm[i]] = h_maps[i];
    lap("results to the caller's order");
    c->stats.evals_last = (uint64_t)P * c->n;
    c->stats.bytes_last = (uint64_t)tiles * c->n * 24ull + (uint64_t)P * sizeof(nhdfit_req) + (uint64_t)P * 8ull;
    c->stats.nodes = c->n; c->stats.nsig = c->nsig; c->stats.ncls = c->ncls; c->stats.lds_bytes = c->lds_bytes;
    c->stats.batch_finds++;
    c->P = 0; c->n_items = 0;                                   // nothing stays staged for nhdfit_enqueue_step / nhdfit_fetch
    c->known_idle = true;                                       // (the word came behind everything this call put on the stream)
    (void)chunks;
    return NHDFIT_OK;
}
}  // namespace

int nhdfit_find(nhdfit_ctx* c, const nhdfit_req* reqs, uint32_t P, double now, const uint64_t* cand,
                uint64_t* score_out, uint64_t* bitmap_out, nhdfit_mapping* map_out) {
    static const bool prof = tune_env("NHDFIT_FIND_PROF") != nullptr;      // tuning aid: host-side phase times of the call
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!prof) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[nhdfit] find P=%u %s %.1f us\n", P, what, std::chrono::duration<double, std::micro>(t1 - t0).count());
        t0 = t1;
    };
    if (c && !bitmap_out) {
        const int rs = find_small(c, reqs, P, now, cand, score_out, map_out);
        if (rs <= 0) { lap("single launch"); return rs; }
    }
    int rc;
    bool staged = false;
    if (c && !bitmap_out) {
        const int rb = find_batch(c, reqs, P, now, cand, score_out, map_out);
        if (rb <= 0) { lap("single launch, batch"); return rb; }
        staged = rb == 2;                                       // (the launch did not run: the staged batch takes the steps' path)
    }
    if (!staged) {
        rc = nhdfit_stage_requests(c, reqs, P);
        if (rc) return rc;
    }
    lap("stage");
    if (cand && (rc = stage_cand(c, cand))) return rc;
    if ((rc = nhdfit_enqueue_step(c, now))) return rc;
    lap("enqueue");
    if (prof) { if ((rc = flush_pipeline(c))) return rc; lap("flush-launches"); if ((rc = nhdfit_sync(c))) return rc; lap("sync"); }
    rc = nhdfit_fetch(c, score_out, bitmap_out, map_out);
    lap("fetch");
    return rc;
}

namespace {
MapTables map_tables(nhdfit_ctx* c) {
    return MapTables{c->asc.p, c->use_choose_tab ? c->choose_tab.p : nullptr,
                     c->use_set_states ? SetStates{c->st_info.p, c->st_next.p, c->st_asc.p, c->st_n} : SetStates{nullptr, nullptr, nullptr, 0}};
}
SigTable sig_table(nhdfit_ctx* c) { return SigTable{c->sig_keys.p, c->sig_ids.p, c->sig_mask}; }
}  // namespace

// ---- nodes beyond the fast layout (wide_core.h / wide_kernel.h) -------------------------------------------------------------
int nhdfit_wide_count(nhdfit_ctx* c, uint32_t* n_wide) {
    if (!c || !n_wide) return NHDFIT_E_INVAL;
    *n_wide = c->n_wide;
    return NHDFIT_OK;
}

int nhdfit_wide_download(nhdfit_ctx* c, nhdfit_wide_node* out, uint32_t cap, uint32_t* n_wide) {
    if (!c || !n_wide) return NHDFIT_E_INVAL;
    *n_wide = c->n_wide;
    if (!c->n_wide || !out) return NHDFIT_OK;
    if (cap < c->n_wide) return fail(c, NHDFIT_E_INVAL, "%u wide records, room for %u", c->n_wide, cap);
    HIPCHK(c, hipSetDevice(c->dev));
    HIPCHK(c, wait_stream(c->stream));
    HIPCHK(c, hipMemcpy(out, c->wide.p, (size_t)c->n_wide * sizeof *out, hipMemcpyDeviceToHost));
    return NHDFIT_OK;
}

int nhdfit_wide_upload(nhdfit_ctx* c, uint32_t first, uint32_t count, const nhdfit_wide_node* wide, uint32_t n_wide) {
    if (!c) return NHDFIT_E_INVAL;
    if (n_wide && !wide) return fail(c, NHDFIT_E_INVAL, "NULL wide records");
    if ((uint64_t)first + count > c->capacity) return fail(c, NHDFIT_E_INVAL, "wide upload [%u,%u) exceeds capacity %u", first, first + count, c->capacity);
    for (uint32_t k = 0; k < n_wide; ++k) {
        if (wide[k].index < first || wide[k].index >= first + count || (k && wide[k].index <= wide[k - 1].index))
            return fail(c, NHDFIT_E_INVAL, "wide record %u: index %u outside [%u,%u) or not ascending", k, wide[k].index, first, first + count);
        if (!wide_shape_ok(wide[k])) return fail(c, NHDFIT_E_LIMIT, "wide record %u: %u NUMA nodes of %u cores, %u GPUs (general path: <= %d of <= %d, <= %d)", k,
                                                 (unsigned)wide[k].numa_nodes, (unsigned)wide[k].cores_per_proc, (unsigned)wide[k].n_gpus,
                                                 NHDFIT_WIDE_MAX_NUMA, NHDFIT_WIDE_MAX_CORES_PER_NUMA, NHDFIT_MAX_GPUS);
        for (uint32_t u = 0; u < (uint32_t)NHDFIT_WIDE_MAX_NUMA; ++u)
            if (wide[k].nic_cnt[u] > NHDFIT_MAX_NICS_PER_NUMA || (u >= wide[k].numa_nodes && wide[k].nic_cnt[u]))
                return fail(c, NHDFIT_E_LIMIT, "wide record %u: %u NICs on NUMA node %u", k, (unsigned)wide[k].nic_cnt[u], u);
    }
    if (!n_wide && !c->n_wide) return NHDFIT_OK;
    HIPCHK(c, hipSetDevice(c->dev));
    { int rc_ = sync_all(c); if (rc_) return rc_; }
    std::vector<nhdfit_wide_node> cur(c->n_wide), next;
    if (c->n_wide) HIPCHK(c, hipMemcpy(cur.data(), c->wide.p, cur.size() * sizeof cur[0], hipMemcpyDeviceToHost));   // (commits may have changed them)
    size_t k = 0;
    for (; k < cur.size() && cur[k].index < first; ++k) next.push_back(cur[k]);
    for (uint32_t j = 0; j < n_wide; ++j) next.push_back(wide[j]);
    for (; k < cur.size(); ++k)
        if (cur[k].index >= first + count) next.push_back(cur[k]);
    HIPCHK(c, c->wide.reserve(next.size() ? next.size() : 1));
    if (!next.empty()) HIPCHK(c, hipMemcpy(c->wide.p, next.data(), next.size() * sizeof next[0], hipMemcpyHostToDevice));
    c->n_wide = (uint32_t)next.size();
    c->sharing = false;                                          // (speed_used records go with the wide records: sent again after them)
    c->wide_max_numa = 0;
    for (const auto& w : next) c->wide_max_numa = std::max<uint32_t>(c->wide_max_numa, w.numa_nodes);
    c->wide_index.resize(next.size());
    for (size_t j = 0; j < next.size(); ++j) c->wide_index[j] = next[j].index;
    return NHDFIT_OK;
}

int nhdfit_wide_share_upload(nhdfit_ctx* c, const nhdfit_wide_share* share, uint32_t n_wide) {
    if (!c) return NHDFIT_E_INVAL;
    HIPCHK(c, hipSetDevice(c->dev));
    { int rc_ = sync_all(c); if (rc_) return rc_; }
    if (!share) { c->sharing = false; return NHDFIT_OK; }
    if (n_wide != c->n_wide) return fail(c, NHDFIT_E_INVAL, "%u speed_used records for %u wide records", n_wide, c->n_wide);
    if (c->n_wide != c->n) return fail(c, NHDFIT_E_STATE, "ENABLE_SHARING: every node of the mirror must be a wide record (%u of %u are)", c->n_wide, c->n);
    HIPCHK(c, c->wide_share.reserve(n_wide ? n_wide : 1));
    if (n_wide) HIPCHK(c, hipMemcpy(c->wide_share.p, share, (size_t)n_wide * sizeof *share, hipMemcpyHostToDevice));
    c->sharing = true;
    return NHDFIT_OK;
}

int nhdfit_wide_share_download(nhdfit_ctx* c, nhdfit_wide_share* out, uint32_t cap, uint32_t* n_wide) {
    if (!c || !n_wide) return NHDFIT_E_INVAL;
    *n_wide = c->sharing ? c->n_wide : 0;
    if (!c->sharing || !c->n_wide) return NHDFIT_OK;
    if (!out || cap < c->n_wide) return fail(c, NHDFIT_E_INVAL, "room for %u speed_used records, the mirror holds %u", cap, c->n_wide);
    HIPCHK(c, hipSetDevice(c->dev));
    { int rc_ = sync_all(c); if (rc_) return rc_; }
    HIPCHK(c, hipMemcpy(out, c->wide_share.p, (size_t)c->n_wide * sizeof *out, hipMemcpyDeviceToHost));
    return NHDFIT_OK;
}

int nhdfit_wide_commit(nhdfit_ctx* c, uint32_t node, const nhdfit_req* req, const nhdfit_mapping* map, double busy_time,
                       nhdfit_wide_placement* place_out) {
    if (!c || !req || !map || !place_out) return NHDFIT_E_INVAL;
    const int slot = c->wide_slot(node);
    if (slot < 0) return fail(c, NHDFIT_E_INVAL, "node %u is not a wide node", node);
    if (!map->valid) return fail(c, NHDFIT_E_INVAL, "the mapping is not valid");
    if (!req_valid(*req)) return fail(c, NHDFIT_E_INVAL, "the request is not valid (map type NUMA / PCI, 1..%d proc groups)", NHDFIT_MAX_GROUPS);
    for (uint32_t g = 0; g <= req->n_groups; ++g) {
        if (map->cpu[g] < 0 || map->cpu[g] >= NHDFIT_WIDE_MAX_NUMA) return fail(c, NHDFIT_E_INVAL, "mapping: cpu[%u] = %d is not a NUMA node", g, (int)map->cpu[g]);
        if (g == req->n_groups) break;
        if (map->gpu[g] < 0 || map->gpu[g] >= NHDFIT_WIDE_MAX_NUMA || map->nic_numa[g] < 0 || map->nic_numa[g] >= NHDFIT_WIDE_MAX_NUMA)
            return fail(c, NHDFIT_E_INVAL, "mapping: group %u sits on NUMA node %d / its NIC on %d", g, (int)map->gpu[g], (int)map->nic_numa[g]);
        if (map->nic_idx[g] < 0 || map->nic_idx[g] >= NHDFIT_MAX_NICS_PER_NUMA)
            return fail(c, NHDFIT_E_INVAL, "mapping: group %u uses NIC ordinal %d", g, (int)map->nic_idx[g]);
    }
    HIPCHK(c, hipSetDevice(c->dev));
    { int rc_ = sync_all(c); if (rc_) return rc_; }             // steps in flight read the wide records
    HIPCHK(c, c->wide_place.reserve(1));
    WideCommitArgs wa;
    memset(&wa, 0, sizeof wa);
    wa.wide = c->wide.p; wa.slot = (uint32_t)slot; wa.req = *req; wa.map = *map; wa.busy_time = busy_time; wa.out = c->wide_place.p;
    wa.share = c->sharing ? c->wide_share.p : nullptr;
    LAUNCH(c, k_wide_commit, dim3(1), dim3(64), 0, c->stream, wa);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(place_out, c->wide_place.p, sizeof *place_out, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, wait_stream(c->stream));
    return NHDFIT_OK;
}

// ---- big requests: pods with 5..8 processing groups, the general path over every node (big_kernel.h) ------------------------
int nhdfit_big_find(nhdfit_ctx* c, const nhdfit_big_req* reqs, uint32_t P, double now, const uint64_t* cand,
                    uint64_t* score_out, nhdfit_big_mapping* map_out) {
    if (!c || !reqs || !P || !score_out) return NHDFIT_E_INVAL;
    if (P > 65535u) return fail(c, NHDFIT_E_LIMIT, "%u big requests in one call (<= 65535)", P);
    if (c->n && !c->ncls) return fail(c, NHDFIT_E_STATE, "set the dictionary first (nhdfit_set_dictionary: the NIC capacity classes)");
    HIPCHK(c, hipSetDevice(c->dev));
    { int rc_ = sync_all(c); if (rc_) return rc_; }             // (a rare call: no need to run beside steps in flight)
    const size_t chunks = (c->n + 63) / 64;
    HIPCHK(c, c->big_reqs.reserve(P));
    HIPCHK(c, c->big_score.reserve(P));
    HIPCHK(c, c->big_maps.reserve(P));
    HIPCHK(c, c->big_flags.reserve(4));
    HIPCHK(c, hipMemcpyAsync(c->big_reqs.p, reqs, (size_t)P * sizeof *reqs, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->big_score.p, 0, (size_t)P * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(c->big_flags.p, 0, 4 * sizeof(uint32_t), c->stream));
    if (cand && chunks) {
        HIPCHK(c, c->big_cand.reserve(chunks));
        HIPCHK(c, hipMemcpyAsync(c->big_cand.p, cand, chunks * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    }
    const uint32_t units = c->n + c->n_wide;
    if (units) {
        BigEvalArgs ea;
        memset(&ea, 0, sizeof ea);
        ea.p0 = c->p0.p; ea.p1 = c->p1.p; ea.p2 = c->p2.p; ea.p3 = c->p3.p; ea.p4 = c->p4.p; ea.det = c->det.p; ea.n = c->n;
        ea.wide = c->wide.p; ea.n_wide = c->n_wide; ea.reqs = c->big_reqs.p; ea.P = P; ea.caps = c->caps.p; ea.busy_from = busy_threshold(now);
        ea.share = c->sharing ? c->wide_share.p : nullptr;
        ea.cand = cand && chunks ? c->big_cand.p : nullptr; ea.score = c->big_score.p; ea.global_base = c->global_base; ea.flags = c->big_flags.p;
        LAUNCH(c, k_big_eval, dim3((units + 63) / 64, P), dim3(64), 0, c->stream, ea);
        HIPCHK(c, hipGetLastError());
    }
    if (c->comm) {      // sharded: one all-reduce(max) of the P packed scores picks the cluster's winners; the owner maps (k_big_map skips the rest)
        HIPCHK(c, wait_stream(c->stream));
        ncclResult_t r = (c->known_idle = false, g_rccl).AllReduce(c->big_score.p, c->big_score.p, P, ncclUint64, ncclMax, c->comm, c->s_red);
        if (r != ncclSuccess) return fail(c, NHDFIT_E_RCCL, "ncclAllReduce: %s", g_rccl.GetErrorString(r));
        HIPCHK(c, wait_stream(c->s_red));
    }
    if (map_out && c->n) {
        // set tables of one mapping: sized for the call's largest group count on the mirror's widest node (2 sockets, 8 groups:
        // 6 x 512 + 2 x 2 048 words; 4 sockets, 8 groups: 6 x 262 144 + 2 x 1 048 576); as many mappings at a time as 1 GiB holds
        uint32_t gmax = 1;
        for (uint32_t i = 0; i < P; ++i) gmax = std::max(gmax, std::min<uint32_t>(reqs[i].n_groups, NHDFIT_BIG_MAX_GROUPS));
        const uint32_t umax = std::max<uint32_t>(NHDFIT_MAX_NUMA, c->n_wide ? c->wide_max_numa : 0);
        const size_t stride = big_scratch_words(umax, gmax);
        const bool lds_tables = stride * sizeof(int32_t) <= 96 * 1024;
        const uint32_t workers = (uint32_t)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(P, 128), ((size_t)1 << 28) / stride));
        HIPCHK(c, c->big_scratch.reserve(lds_tables ? 1 : (size_t)workers * stride));
        BigMapArgs ma;
        memset(&ma, 0, sizeof ma);
        ma.p0 = c->p0.p; ma.p1 = c->p1.p; ma.p2 = c->p2.p; ma.p3 = c->p3.p; ma.p4 = c->p4.p; ma.det = c->det.p; ma.n = c->n;
        ma.wide = c->wide.p; ma.n_wide = c->n_wide; ma.reqs = c->big_reqs.p; ma.P = P; ma.caps = c->caps.p; ma.share = c->sharing ? c->wide_share.p : nullptr;
        ma.score = c->big_score.p; ma.global_base = c->global_base; ma.out = c->big_maps.p; ma.scratch = c->big_scratch.p; ma.flags = c->big_flags.p;
        ma.stride = stride; ma.slots_g = (int32_t)wide_table_slots(wide_ipow(umax, gmax)); ma.slots_c = (int32_t)wide_table_slots(wide_ipow(umax, gmax + 1)); ma.workers = workers;
        ma.lds_tables = lds_tables ? 1u : 0u;
        if (lds_tables) HIPCHK(c, hipFuncSetAttribute((const void*)k_big_map, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        LAUNCH(c, k_big_map, dim3(workers), dim3(64), lds_tables ? stride * sizeof(int32_t) : 0, c->stream, ma);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(map_out, c->big_maps.p, (size_t)P * sizeof *map_out, hipMemcpyDeviceToHost, c->stream));
    } else if (map_out) {
        memset(map_out, 0, (size_t)P * sizeof *map_out);
    }
    uint32_t fl[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(score_out, c->big_score.p, (size_t)P * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(fl, c->big_flags.p, sizeof fl, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, wait_stream(c->stream));
    c->stats.big_nic_steps_max = std::max(c->stats.big_nic_steps_max, fl[2]);
    if (fl[1]) return fail(c, NHDFIT_E_LIMIT, "a big request's NIC stage ran out of search budget on some node (%u steps per pod and node)", (unsigned)NHDFIT_BIG_NIC_BUDGET);
    if (fl[0]) return fail(c, NHDFIT_E_LIMIT, "the set model of a big request's mapping outgrew its table");
    return NHDFIT_OK;
}

int nhdfit_big_commit(nhdfit_ctx* c, uint32_t node, const nhdfit_big_req* req, const nhdfit_big_mapping* map, double busy_time,
                      nhdfit_big_placement* place_out) {
    if (!c || !req || !map || !place_out) return NHDFIT_E_INVAL;
    if (node >= c->n) return fail(c, NHDFIT_E_INVAL, "node %u out of range (%u nodes)", node, c->n);
    if (!map->valid) return fail(c, NHDFIT_E_INVAL, "the mapping is not valid");
    if (!req_valid(*req)) return fail(c, NHDFIT_E_INVAL, "the request is not valid (map type NUMA / PCI, 1..%d proc groups)", NHDFIT_BIG_MAX_GROUPS);
    const int slot = c->wide_slot(node);
    const int numa_lim = slot >= 0 ? NHDFIT_WIDE_MAX_NUMA : NHDFIT_MAX_NUMA;
    for (uint32_t g = 0; g <= req->n_groups; ++g) {
        if (map->cpu[g] < 0 || map->cpu[g] >= numa_lim) return fail(c, NHDFIT_E_INVAL, "mapping: cpu[%u] = %d is not a NUMA node", g, (int)map->cpu[g]);
        if (g == req->n_groups) break;
        if (map->gpu[g] < 0 || map->gpu[g] >= numa_lim || map->nic_numa[g] < 0 || map->nic_numa[g] >= numa_lim)
            return fail(c, NHDFIT_E_INVAL, "mapping: group %u sits on NUMA node %d / its NIC on %d", g, (int)map->gpu[g], (int)map->nic_numa[g]);
        if (map->nic_idx[g] < 0 || map->nic_idx[g] >= NHDFIT_MAX_NICS_PER_NUMA)
            return fail(c, NHDFIT_E_INVAL, "mapping: group %u uses NIC ordinal %d", g, (int)map->nic_idx[g]);
    }
    HIPCHK(c, hipSetDevice(c->dev));
    { int rc_ = sync_all(c); if (rc_) return rc_; }             // the commit writes the mirror: steps in flight on either pipe read it
    HIPCHK(c, c->big_place.reserve(1));
    BigCommitArgs ba;
    memset(&ba, 0, sizeof ba);
    ba.p0 = c->p0.p; ba.p1 = c->p1.p; ba.p2 = c->p2.p; ba.p3 = c->p3.p; ba.p4 = c->p4.p; ba.det = c->det.p;
    ba.wide = c->wide.p; ba.slot = slot; ba.node = node; ba.req = *req; ba.map = *map; ba.busy_time = busy_time; ba.sigs = sig_table(c);
    ba.share = c->sharing ? c->wide_share.p : nullptr;
    ba.out = c->big_place.p;
    LAUNCH(c, k_big_commit, dim3(1), dim3(64), 0, c->stream, ba);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(place_out, c->big_place.p, sizeof *place_out, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, wait_stream(c->stream));
    if (slot < 0) {                                             // the node's records (X class, free-core counts) follow its planes
        if (c->rec_lo == c->rec_hi) { c->rec_lo = node; c->rec_hi = node + 1; }
        else { c->rec_lo = std::min(c->rec_lo, node); c->rec_hi = std::max(c->rec_hi, node + 1); }
    }
    return NHDFIT_OK;
}

int nhdfit_wide_placements(nhdfit_ctx* c, nhdfit_wide_placement* out, uint32_t cap, uint32_t* n) {
    if (!c || !n) return NHDFIT_E_INVAL;
    *n = (uint32_t)c->wide_places_last.size();
    if (out)
        for (uint32_t k = 0; k < *n && k < cap; ++k) out[k] = c->wide_places_last[k];
    return NHDFIT_OK;
}

namespace {
// Mode B while the mirror holds wide nodes: the scheduler's loop as it stands (nhd/NHDScheduler.py:425-437) - FindNode for
// pod k (table pass + general pass, one score word), the commit step on whichever mirror holds the winner, then pod k + 1.
// No decision engine here: a cluster with wide nodes is served exactly, pod by pod.
int schedule_batch_general(nhdfit_ctx* c, const nhdfit_req* reqs, uint32_t P, double now, const uint64_t* cand, int apply,
                           int64_t* node_out, nhdfit_mapping* map_out, nhdfit_placement* place_out, int32_t* status_out, uint32_t* n_done) {
    struct Saved { uint32_t node; nhdfit_plane0 p0; nhdfit_plane1 p1; nhdfit_plane2 p2; nhdfit_plane3 p3; nhdfit_plane4 p4; nhdfit_detail det; };
    std::vector<Saved> saved;                                   // first-touch copies of ordinary nodes (apply = 0)
    std::set<uint32_t> saved_nodes;
    if (!c->warned_general_loop) {                              // (once per context: the whole cluster pays for its wide nodes here)
        c->warned_general_loop = true;
        fprintf(stderr, "[nhdfit] the mirror holds %u node(s) beyond the fast layout: nhdfit_schedule_batch runs the scheduler's loop pod by pod "
                        "(a find and a commit each) instead of the decision engine\n", c->n_wide);
    }
    std::vector<nhdfit_wide_node> wide_before(c->n_wide);
    std::vector<nhdfit_wide_share> share_before(c->sharing ? c->n_wide : 0);
    if (!apply && c->n_wide) {
        uint32_t nw = 0;
        int rc = nhdfit_wide_download(c, wide_before.data(), (uint32_t)wide_before.size(), &nw);
        if (!rc && !share_before.empty()) rc = nhdfit_wide_share_download(c, share_before.data(), (uint32_t)share_before.size(), &nw);
        if (rc) return rc;
    }
    uint32_t done = 0;
    int rc = NHDFIT_OK;
    for (uint32_t i = 0; i < P && rc == NHDFIT_OK; ++i) {
        uint64_t score = 0;
        nhdfit_mapping mp;
        memset(&mp, 0, sizeof mp);
        node_out[i] = -1;
        if (map_out) memset(&map_out[i], 0, sizeof map_out[i]);
        if (place_out) memset(&place_out[i], 0, sizeof place_out[i]);
        if (status_out) status_out[i] = 0;
        done = i + 1;
        if (!req_valid(reqs[i])) continue;
        if ((rc = nhdfit_find(c, reqs + i, 1, now, cand, &score, nullptr, &mp))) break;
        if (!score) continue;
        const uint64_t gi = NHDFIT_SCORE_INDEX(score);
        const uint32_t v = (uint32_t)(gi - c->global_base);
        node_out[i] = (int64_t)gi;
        if (map_out) map_out[i] = mp;
        if (!mp.valid) { rc = fail(c, NHDFIT_E_STATE, "pod %u: no mapping for its feasible node %u", i, v); break; }
        if (c->wide_slot(v) >= 0) {
            nhdfit_wide_placement wp;
            if ((rc = nhdfit_wide_commit(c, v, reqs + i, &mp, now, &wp))) break;
            wp.pod = i; wp.node = v;
            c->wide_places_last.push_back(wp);
            if (place_out) place_out[i].status = NHDFIT_COMMIT_WIDE;
            if (status_out) status_out[i] = wp.status == NHDFIT_COMMIT_WOULD_RAISE ? NHDFIT_COMMIT_WOULD_RAISE : 0;
            continue;
        }
        if (!apply) {
            if (saved_nodes.insert(v).second) {
                Saved sv; sv.node = v;
                if ((rc = nhdfit_download_nodes(c, v, 1, &sv.p0, &sv.p1, &sv.p2, &sv.p3, &sv.p4, &sv.det))) break;
                saved.push_back(sv);
            }
        }
        nhdfit_placement pl;
        if ((rc = nhdfit_commit(c, v, reqs + i, &mp, now, &pl))) break;
        if (place_out) place_out[i] = pl;
        if (status_out) status_out[i] = pl.status;
        if (pl.status == NHDFIT_COMMIT_NEW_SIG) break;          // the caller interns the state and submits the rest (include/nhdfit.h)
    }
    if (!apply) {                                               // the mirror as it was
        for (const Saved& sv : saved) {
            const int r2 = nhdfit_upload_nodes(c, sv.node, 1, &sv.p0, &sv.p1, &sv.p2, &sv.p3, &sv.p4, &sv.det);
            if (r2 && !rc) rc = r2;
        }
        if (!wide_before.empty()) {
            HIPCHK(c, wait_stream(c->stream));
            HIPCHK(c, hipMemcpy(c->wide.p, wide_before.data(), wide_before.size() * sizeof wide_before[0], hipMemcpyHostToDevice));
            if (!share_before.empty()) HIPCHK(c, hipMemcpy(c->wide_share.p, share_before.data(), share_before.size() * sizeof share_before[0], hipMemcpyHostToDevice));
        }
    }
    if (rc) return rc;
    *n_done = done;
    return NHDFIT_OK;
}
}  // namespace

int nhdfit_schedule_batch(nhdfit_ctx* c, const nhdfit_req* reqs, uint32_t P, double now, const uint64_t* cand, int apply,
                          int64_t* node_out, nhdfit_mapping* map_out, nhdfit_placement* place_out, int32_t* status_out,
                          uint32_t* n_done) {
    if (!c) return NHDFIT_E_INVAL;
    if (!node_out || !n_done) return fail(c, NHDFIT_E_INVAL, "node_out / n_done is NULL");
    // A sequential batch is decided on THIS context's nodes alone - with a communicator attached too: it is the per-shard pass of
    // mode B across shards (nhd_amd/sharding.py hands the pods a shard could not place to the next rank), so its snapshot step
    // must not all-reduce its scores with ranks that are working on other slices.  (Round 4 refused the call outright; the ring's
    // first run on real ranks would have ended there - found by the rank-to-rank GPU test of round 5.)
    struct CommOff {
        nhdfit_ctx* c; ncclComm_t saved;
        explicit CommOff(nhdfit_ctx* c_) : c(c_), saved(c_->comm) { c->comm = nullptr; }
        ~CommOff() { c->comm = saved; }
    };
    if (c->comm) { int rc_ = sync_all(c); if (rc_) return rc_; }   // (steps of a staged batch still carry their all-reduce)
    CommOff comm_off(c);
    if (!std::isfinite(now)) return fail(c, NHDFIT_E_INVAL, "now must be finite (a placed node is busy at `now`)");
    HIPCHK(c, hipSetDevice(c->dev));
    c->wide_places_last.clear();
    if (c->n_wide) return schedule_batch_general(c, reqs, P, now, cand, apply, node_out, map_out, place_out, status_out, n_done);
    hipStream_t sm = c->stream;
    Pipe& p = c->pipe[0];                                       // (the one step a freshly staged batch enqueues runs on pipe 0)
    const uint32_t chunks = (c->n + 63) / 64;
    const uint32_t tiles = (P + kTile - 1) / kTile;
    int rc;
    {
        // snapshot pass: digest + fit (verdict matrix and first-fit scores; the mapping roles are not needed - every
        // placement of the batch is mapped against the node's state at ITS turn)
        const bool wb = c->want_bitmap, wm = c->want_map;
        c->want_bitmap = true; c->want_map = false;
        c->all_sigs = true;                              // the decisions read R rows for states no node is in yet (a committed node's new signature)
        rc = nhdfit_stage_requests(c, reqs, P);
        if (!rc && cand) rc = stage_cand(c, cand);
        if (!rc) rc = nhdfit_enqueue_step(c, now);
        c->all_sigs = false;
        c->want_bitmap = wb; c->want_map = wm;
        if (rc) return rc;
        HIPCHK(c, c->nogpu.reserve(chunks ? chunks : 1));
        HIPCHK(c, c->taken.reserve(chunks ? chunks : 1));
        HIPCHK(c, c->tile_masks.reserve((size_t)tiles * 2));
        HIPCHK(c, c->touched.reserve(c->n ? c->n : 1));
        HIPCHK(c, c->gl_tiles.reserve(tiles));
        HIPCHK(c, c->seq_counters.reserve(4));
        HIPCHK(c, c->undo.reserve(P));
        HIPCHK(c, c->seq_out.reserve(P));
        HIPCHK(c, c->seq_place.reserve(P));
        HIPCHK(c, c->seq_ctrl.reserve(32));
        HIPCHK(c, c->seq_ent.reserve(P));
        HIPCHK(c, c->seq_mat.reserve(c->n ? c->n : 1));
        HIPCHK(c, c->seq_flags.reserve(4));
        HIPCHK(c, c->rows_t.reserve((size_t)chunks * P));         // (convert_rows_t's: the arguments below hold its address before the first pass fills it)
        // caller's pod -> staged (class-sorted) position, and behind it the decision engine's list [the pods without GPUs | every other
        // pod] (caller's indices ascending): one page-locked block, ONE copy command (each costs the copy engine ~10 us; out of pageable
        // memory the runtime stages it besides)
        HIPCHK(c, c->order.reserve(2 * (size_t)P));
        HIPCHK(c, c->pin_order.reserve(2 * (size_t)P));
        uint32_t* order_h = c->pin_order.p;
        uint32_t* tn_h = c->pin_order.p + P;
        for (uint32_t i = 0; i < P; ++i) order_h[c->perm[i]] = i;
        {
            uint32_t at = 0;
            c->tn_n = 0;
            for (int pass = 0; pass < 2; ++pass) {
                for (uint32_t i = 0; i < P; ++i) {
                    uint32_t g = 0;
                    const bool valid = req_valid(reqs[i]);
                    if (valid) for (uint32_t k = 0; k < reqs[i].n_groups; ++k) g += reqs[i].gpus[k];
                    if ((valid && g == 0) == (pass == 0)) tn_h[at++] = i;
                }
                if (pass == 0) c->tn_n = at;
            }
        }
        HIPCHK(c, hipMemcpyAsync(c->order.p, order_h, 2 * (size_t)P * sizeof(uint32_t), hipMemcpyHostToDevice, sm));
        LAUNCH(c, k_nogpu, dim3(chunks), dim3(64), 0, sm, c->p2.p, c->n, c->nogpu.p);
        const int b0 = (int)((p.n_fit - 1) % kBufs);
        LAUNCH(c, k_tile_masks, dim3(tiles), dim3(64), 0, sm, p.hdr[b0].p, tiles, c->tile_masks.p);
        HIPCHK(c, hipGetLastError());
    }
    // pod-major verdict rows of the snapshot + empty taken / first-touch state (again before a fallback pass)
    // (one launch clears everything a pass starts from - seven fill commands in a row cost more than the fills: `engine` adds the
    // decision engine's words)
    uint32_t queue_len = 0;
    auto reset_scan_state = [&](bool engine) -> int {
        int rc_ = convert_rows_t(c, p);
        if (rc_) return rc_;
        SeqResetArgs ra;
        memset(&ra, 0, sizeof ra);
        ra.taken = c->taken.p; ra.chunks = chunks ? chunks : 1; ra.touched = c->touched.p; ra.n = c->n;
        ra.counters = c->seq_counters.p; ra.flags = c->seq_flags.p;
        if (engine) { ra.ctrl = c->seq_ctrl.p; ra.mat = c->seq_mat.p; ra.queue = c->seq_queue.p; ra.queue_len = queue_len; }
        const uint32_t most = std::max(std::max(ra.chunks, ra.n), ra.queue_len);
        LAUNCH(c, k_seq_reset, dim3(std::min<uint32_t>((most + 255) / 256, 1024u)), dim3(256), 0, sm, ra);
        HIPCHK(c, hipGetLastError());
        return NHDFIT_OK;
    };
    const int b = (int)((p.n_fit - 1) % kBufs);
    SeqArgs sa;
    memset(&sa, 0, sizeof sa);
    sa.p0 = c->p0.p; sa.p1 = c->p1.p; sa.p2 = c->p2.p; sa.p3 = c->p3.p; sa.p4 = c->p4.p; sa.det = c->det.p;
    sa.n = c->n; sa.chunks = chunks; sa.global_base = c->global_base; sa.now = now;
    sa.reqs = c->reqs.p; sa.score = p.score[b].p; sa.P = P; sa.order = c->order.p;
    sa.tabs = p.tabs[b].p; sa.pitch = c->pitch; sa.tile_wcls = c->tile_wcls.p;
    for (int w = 0; w < kWClasses; ++w) sa.L[w] = c->L[w];
    sa.rows = c->rows_t.p; sa.taken = c->taken.p; sa.nogpu = c->nogpu.p; sa.tile_masks = c->tile_masks.p;
    sa.caps = c->caps.p; sa.sigs = sig_table(c); sa.fc_dim = c->max_cores + 1; sa.fg_dim = c->max_gpus + 1; sa.ngs = c->ngs;
    sa.mt = map_tables(c);
    sa.undo = c->undo.p; sa.touched = c->touched.p; sa.counters = c->seq_counters.p; sa.keep_undo = 1;
    sa.out = c->seq_out.p; sa.place = c->seq_place.p; sa.n_done = c->seq_counters.p + 1; sa.gl_tiles = c->gl_tiles.p;

    // The general kernel: one block walks `n_list` pods (all P when list is null) in order, kSeqPods per round.
    auto run_general = [&](const uint32_t* list_dev, uint32_t n_list, uint32_t& done) -> int {
        SeqArgs g = sa;
        g.list = list_dev; g.n_list = n_list;
        size_t seq_lds = lds_slice((size_t)tiles * 16) + lds_slice(((size_t)c->sig_mask + 1) * 8) + lds_slice(((size_t)c->sig_mask + 1) * 4) + lds_slice((size_t)P * 4) + lds_slice(tiles);
        g.lds_tables = seq_lds <= 96 * 1024;
        if (!g.lds_tables) seq_lds = 0;
        const int seq_pods = c->seq_pods;
        HIPCHK(c, hipFuncSetAttribute(seq_pods == 16 ? (const void*)k_seq<16> : (const void*)k_seq<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        const bool seq_prof = tune_env("NHDFIT_SEQ_PROF") != nullptr;
        if (seq_prof) { HIPCHK(c, c->role_clock.reserve(16)); g.prof = c->role_clock.p; }
        if (seq_pods == 16) LAUNCH(c, k_seq<16>, dim3(1), dim3(1024), seq_lds, sm, g);
        else LAUNCH(c, k_seq<8>, dim3(1), dim3(512), seq_lds, sm, g);
        HIPCHK(c, hipGetLastError());
        if (seq_prof) {
            unsigned long long t[16];
            HIPCHK(c, wait_stream(sm));
            HIPCHK(c, hipMemcpy(t, c->role_clock.p, sizeof t, hipMemcpyDeviceToHost));
            const double per = 0.01 / (double)(t[3] ? t[3] : 1);
            fprintf(stderr, "[nhdfit] k_seq wave 0 per round: node load %.1f, NIC bits %.1f, mapping %.1f, commit %.1f, write-back %.1f us\n",
                    t[5] * per, t[6] * per, t[7] * per, t[8] * per, t[9] * per);
            fprintf(stderr, "[nhdfit] k_seq: %llu pods in %llu rounds; scan %.1f us, pick %.1f us, map+commit %.1f us, columns %.1f us per round\n", t[4], t[3],
                    t[0] * per, t[2] * per, t[1] * per, t[10] * per);
            fprintf(stderr, "[nhdfit] k_seq columns: evaluate %.1f, patch %.1f, fence %.1f, barrier %.1f us per round; %.1f (node, tile) pairs per round\n",
                    t[11] * per, t[12] * per, t[13] * per, t[14] * per, (double)t[15] / (double)(t[3] ? t[3] : 1));
        }
        uint32_t counters[4] = {0, 0, 0, 0};
        HIPCHK(c, hipMemcpyAsync(counters, c->seq_counters.p, sizeof counters, hipMemcpyDeviceToHost, sm));
        HIPCHK(c, wait_stream(sm));
        done = counters[1];
        return NHDFIT_OK;
    };
    auto undo_all = [&]() -> int {                              // every node this batch touched goes back to its first-touch copy
        LAUNCH(c, k_undo, dim3(P), dim3(64), 0, sm, sa);
        HIPCHK(c, hipGetLastError());
        return NHDFIT_OK;
    };

    uint32_t decided = 0;                                       // pods [0, decided) of the caller's order are decided
    // The decision engine (seq2_kernel.h) takes every batch its block 0 has the LDS for: two bit maps over the nodes, one entry per
    // GPU-less pod (the multiset of their commits), one bit per pod; the optional tables go in after those.
    const uint32_t n_gpu_less = c->tn_n;
    const uint32_t hash_slots = decide_hash_slots(n_gpu_less);
    // (the two node bit maps may stop short of the mirror - DecideArgs::span: config 5's whole cluster is 4 096 chunks, 64 KB on their own;
    // they then cover what the 64 KB leave, at least 512 chunks, and a decision past them sends the batch to the general kernel)
    const size_t dyn_fixed = lds_slice((size_t)hash_slots * 4) + lds_slice((size_t)((P + 31) / 32) * 4);
    uint32_t span = chunks;
    if (2 * lds_slice((size_t)chunks * 8) + dyn_fixed > 64 * 1024)
        span = dyn_fixed + 2 * 512 * 8 <= 64 * 1024 ? (uint32_t)((64 * 1024 - dyn_fixed) / 16) & ~15u : 0u;
    static const uint32_t force_span = tune_env("NHDFIT_SEQ_SPAN") ? (uint32_t)atoi(tune_env("NHDFIT_SEQ_SPAN")) : 0u;   // tuning aid (tests of the fallback)
    if (force_span && force_span < span) span = force_span;
    const size_t dyn_base = 2 * lds_slice((size_t)span * 8) + dyn_fixed;
    bool any_g4 = false;                                        // (four-group pods: the instantiation that carries the generic set model)
    for (uint32_t i = 0; i < P && !any_g4; ++i) any_g4 = reqs[i].n_groups > 3;
    // What the block may ask for is the CU's 160 KB less the kernel's own static LDS (read from the code object: it moves with every edit
    // of seq2_kernel.h - the optional tables were once admitted against fixed marks that assumed 45 KB of it and a batch whose tables all
    // fitted those marks was refused by hipFuncSetAttribute, found by tools/soak_mode_b_gpu.py).
    static size_t decide_static[2] = {0, 0};
    if (!decide_static[any_g4]) {
        hipFuncAttributes fa;
        HIPCHK(c, hipFuncGetAttributes(&fa, any_g4 ? (const void*)k_decide<true> : (const void*)k_decide<false>));
        decide_static[any_g4] = fa.sharedSizeBytes ? fa.sharedSizeBytes : 1;
    }
    const size_t dyn_room = decide_static[any_g4] < 160 * 1024 ? 160 * 1024 - decide_static[any_g4] : 0;
    bool fast = !c->seq_general && span > 0 && c->n > 0 && P < (1u << 26) && dyn_base <= dyn_room;
    if (fast) {
        queue_len = P * 4u;                                     // a commit per pod + up to three patch items per commit of a GPU-less pod
        HIPCHK(c, c->seq_queue.reserve(queue_len));
    }
    if ((rc = reset_scan_state(fast))) return rc;
    if (fast) {
        // The decision engine (seq2_kernel.h): one block decides, the rest of the grid commits.
        const uint32_t* list_dev = c->order.p + P;              // [the pods without GPUs | every other pod] (uploaded behind the order)
        const uint32_t n_n = c->tn_n, n_g = P - c->tn_n;
        LAUNCH(c, k_decide_prep, dim3((P + 255) / 256), dim3(256), 0, sm, list_dev, P, c->order.p, p.score[b].p, c->global_base, c->seq_ent.p);
        HIPCHK(c, hipGetLastError());
        DecideArgs qa;
        memset(&qa, 0, sizeof qa);
        qa.list_n = list_dev; qa.n_n = n_n; qa.list_g = list_dev + n_n; qa.n_g = n_g; qa.ent_n = c->seq_ent.p; qa.ent_g = c->seq_ent.p + n_n; qa.queue_len = queue_len; qa.ncls = c->ncls;
        qa.hash_slots = hash_slots;
        qa.span = span;
        qa.dbg = tune_env("NHDFIT_SEQ_SKIP") ? (uint32_t)atoi(tune_env("NHDFIT_SEQ_SKIP")) : 0u;
        qa.s = sa; qa.queue = c->seq_queue.p; qa.ctrl = c->seq_ctrl.p; qa.mat = c->seq_mat.p; qa.flags = c->seq_flags.p;
        size_t dyn = dyn_base;
        const size_t sig_bytes = lds_slice(((size_t)c->sig_mask + 1) * 8) + lds_slice(((size_t)c->sig_mask + 1) * 4);
        const size_t st_bytes = lds_slice((size_t)c->st_n * 8) + lds_slice((size_t)c->st_n * 32) + lds_slice(256 * 4);
        const size_t room96 = dyn_room < 96 * 1024 ? dyn_room : 96 * 1024, room112 = dyn_room < 112 * 1024 ? dyn_room : 112 * 1024;
        if (dyn + sig_bytes <= room96) { qa.lds_sigs = 1; dyn += sig_bytes; }
        if (c->use_set_states && c->st_n && dyn + st_bytes <= room96) { qa.lds_states = 1; dyn += st_bytes; }
        if (c->use_choose_tab && dyn + lds_slice(kChooseEntries) <= room112) { qa.lds_choose = 1; dyn += lds_slice(kChooseEntries); }
        HIPCHK(c, hipFuncSetAttribute(any_g4 ? (const void*)k_decide<true> : (const void*)k_decide<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        static const uint32_t workers = tune_env("NHDFIT_SEQ_WORKERS") ? (uint32_t)atoi(tune_env("NHDFIT_SEQ_WORKERS")) : (uint32_t)kWorkerBlocks;   // tuning aid
        if (any_g4) LAUNCH(c, k_decide<true>, dim3(1 + (workers ? workers : 1u)), dim3(64 * kDecideWaves), dyn, sm, qa);
        else LAUNCH(c, k_decide<false>, dim3(1 + (workers ? workers : 1u)), dim3(64 * kDecideWaves), dyn, sm, qa);
        HIPCHK(c, hipGetLastError());
        uint32_t flags[4] = {0, 0, 0, 0};
        HIPCHK(c, hipMemcpyAsync(flags, c->seq_flags.p, sizeof flags, hipMemcpyDeviceToHost, sm));
        HIPCHK(c, wait_stream(sm));
        if (tune_env("NHDFIT_SEQ_PROF")) {
            uint32_t ctl[32];
            HIPCHK(c, hipMemcpy(ctl, c->seq_ctrl.p, sizeof ctl, hipMemcpyDeviceToHost));
            fprintf(stderr, "[nhdfit] k_decide: %u queue items; GPU-less pods: %u verifications failed, %u looked at a node in LDS, %u at a published one, "
                            "%u at an untouched one, %u waited for an earlier pod's target, %u window rescans, %u sent back by the sequencer\n",
                    ctl[1] ? ctl[1] - 1 : 0, ctl[4], ctl[5], ctl[6], ctl[7], ctl[8], ctl[14], ctl[15]);
            fprintf(stderr, "[nhdfit] k_decide sequencer: waiting for fetchers %.2f ms, pods with GPUs %.2f ms, GPU-less pods: waiting for their speculators %.2f ms, "
                            "validation %.2f ms\n", ctl[9] * 1e-5, ctl[10] * 1e-5, ctl[11] * 1e-5, ctl[12] * 1e-5);
            fprintf(stderr, "[nhdfit] k_decide fetcher 0: waiting for list entries / windows %.2f ms, for the ring %.2f ms, issue + park %.2f ms\n", ctl[2] * 1e-5, ctl[3] * 1e-5, ctl[13] * 1e-5);
            fprintf(stderr, "[nhdfit] k_decide sequencer, a pod with GPUs: window + pick %.2f ms, take %.2f ms, queue entry %.2f ms\n", ctl[28] * 1e-5, ctl[29] * 1e-5, ctl[30] * 1e-5);
            fprintf(stderr, "[nhdfit] k_decide speculators, finer: candidate pick %.2f ms, pre-checks %.2f ms, NIC bits %.2f ms, (mapping = verification), before post %.2f ms, "
                            "post + first-touch + result %.2f ms, (summary = stage 1)\n", ctl[23] * 1e-5, ctl[24] * 1e-5, ctl[25] * 1e-5, ctl[26] * 1e-5, ctl[27] * 1e-5);
            fprintf(stderr, "[nhdfit] k_decide speculators (%d, summed): set-up %.2f ms, node state %.2f ms, verification %.2f ms, commit stage 1 %.2f ms, waiting (sequencer, earlier pods) %.2f ms, "
                            "commit stage 2 %.2f ms, publication %.2f ms\n", kSpecWaves, ctl[16] * 1e-5, ctl[17] * 1e-5, ctl[18] * 1e-5, ctl[19] * 1e-5, ctl[20] * 1e-5, ctl[22] * 1e-5, ctl[21] * 1e-5);
        }
        if (flags[1] || flags[3]) {
            // a NIC state without a signature id (or a wait that ran out): start over with the kernel whose stop / intern /
            // resume protocol the caller knows
            if ((rc = undo_all())) return rc;
            if ((rc = reset_scan_state(false))) return rc;
            fast = false;
        } else decided = P;
    }
    if (!fast) {
        if ((rc = run_general(nullptr, 0, decided))) return rc;
    }
    if (!apply) { if ((rc = undo_all())) return rc; }
    c->seq_host.resize(P);
    HIPCHK(c, hipMemcpyAsync(c->seq_host.data(), c->seq_out.p, P * sizeof(SeqResult), hipMemcpyDeviceToHost, sm));
    if (place_out) HIPCHK(c, hipMemcpyAsync(place_out, c->seq_place.p, (size_t)P * sizeof(nhdfit_placement), hipMemcpyDeviceToHost, sm));
    rc = nhdfit_sync(c);
    if (rc) return rc;
    *n_done = decided;
    int64_t lo = -1, hi = -1;
    for (uint32_t i = 0; i < decided; ++i) {
        const SeqResult& o = c->seq_host[i];
        node_out[i] = o.node;
        if (map_out) map_out[i] = o.map;
        if (status_out) status_out[i] = o.status;
        if (o.node >= 0) {
            const int64_t v = o.node - (int64_t)c->global_base;
            lo = lo < 0 || v < lo ? v : lo;
            hi = v + 1 > hi ? v + 1 : hi;
        }
    }
    if (apply && lo >= 0) {                                     // records of the committed nodes are stale
        if (c->rec_lo == c->rec_hi) { c->rec_lo = (uint32_t)lo; c->rec_hi = (uint32_t)hi; }
        else { c->rec_lo = std::min(c->rec_lo, (uint32_t)lo); c->rec_hi = std::max(c->rec_hi, (uint32_t)hi); }
    }
    return NHDFIT_OK;
}

int nhdfit_find_sequential(nhdfit_ctx* c, const nhdfit_req* reqs, uint32_t P, double now, const uint64_t* cand,
                           int64_t* node_out, nhdfit_mapping* map_out, int32_t* status_out) {
    // the mirror is left as it was (k_undo): a NIC state without a signature cannot be patched in, the batch fails
    uint32_t done = 0;
    int rc = nhdfit_schedule_batch(c, reqs, P, now, cand, 0, node_out, map_out, nullptr, status_out, &done);
    if (rc) return rc;
    if (done < P) return fail(c, NHDFIT_E_STATE, "pod %u left its node in a NIC state the dictionary has no signature for: "
                              "use nhdfit_schedule_batch (apply) and intern it", done - 1);
    return NHDFIT_OK;
}

int nhdfit_commit(nhdfit_ctx* c, uint32_t node, const nhdfit_req* req, const nhdfit_mapping* map, double busy_time,
                  nhdfit_placement* place_out) {
    if (!c || !req || !map || !place_out) return NHDFIT_E_INVAL;
    if (node >= c->n) return fail(c, NHDFIT_E_INVAL, "node %u out of range (%u nodes)", node, c->n);
    if (!map->valid) return fail(c, NHDFIT_E_INVAL, "the mapping is not valid");
    if (!req_valid(*req)) return fail(c, NHDFIT_E_INVAL, "the request is not valid (map type NUMA / PCI, 1..%d proc groups)", NHDFIT_MAX_GROUPS);
    if (c->wide_slot(node) >= 0) return fail(c, NHDFIT_E_INVAL, "node %u is a wide node: its commit step is nhdfit_wide_commit", node);
    for (uint32_t g = 0; g <= req->n_groups; ++g) {          // entries index two-element arrays / 16-entry NIC tables on the device
        if (map->cpu[g] < 0 || map->cpu[g] >= NHDFIT_MAX_NUMA) return fail(c, NHDFIT_E_INVAL, "mapping: cpu[%u] = %d is not a NUMA node", g, (int)map->cpu[g]);
        if (g == req->n_groups) break;
        if (map->gpu[g] < 0 || map->gpu[g] >= NHDFIT_MAX_NUMA || map->nic_numa[g] < 0 || map->nic_numa[g] >= NHDFIT_MAX_NUMA)
            return fail(c, NHDFIT_E_INVAL, "mapping: group %u sits on NUMA node %d / its NIC on %d", g, (int)map->gpu[g], (int)map->nic_numa[g]);
        if (map->nic_idx[g] < 0 || map->nic_idx[g] >= NHDFIT_MAX_NICS_PER_NUMA)
            return fail(c, NHDFIT_E_INVAL, "mapping: group %u uses NIC ordinal %d", g, (int)map->nic_idx[g]);
    }
    HIPCHK(c, hipSetDevice(c->dev));
    // the commit writes the planes: steps in flight on EITHER pipe (their fit roles, digests running ahead, pending mapping
    // phases) read them - wait for both, as every other writer of the mirror does (nhdfit_upload_nodes)
    // ... unless nothing can be in flight anywhere but on this very stream: no batch is staged (the single-launch finds leave none) and
    // the other pipes' streams and the reduce stream have carried nothing since they were last waited for - then stream order is all
    // the commit needs (the scheduler's pod-at-a-time loop: asking the stream whether the find's launch has retired cost ~10 us per pod)
    if (c->P || c->side_streams_used) { int rc_ = sync_all(c); if (rc_) return rc_; }
    if (!c->commit_host) return fail(c, NHDFIT_E_STATE, "no host block for the commit's result");
    CommitArgs ca;
    memset(&ca, 0, sizeof ca);
    ca.p0 = c->p0.p; ca.p1 = c->p1.p; ca.p2 = c->p2.p; ca.p3 = c->p3.p; ca.p4 = c->p4.p; ca.det = c->det.p;
    ca.node = node; ca.req = *req; ca.map = *map; ca.busy_time = busy_time; ca.sigs = sig_table(c);
    const uint32_t seq = ++c->commit_seq ? c->commit_seq : ++c->commit_seq;                 // (never 0: the block's resting value)
    ca.host = c->commit_host; ca.seq = seq; ca.ncls = c->ncls;
    const auto t_launch = std::chrono::steady_clock::now();
    LAUNCH(c, k_commit, dim3(1), dim3(64), 0, c->stream, ca);       // (both pipes are idle: sync_all above)
    HIPCHK(c, hipGetLastError());
    // the placement arrives in the fine-grained host block behind the sequence number: poll it (a launch that takes longer than
    // half a millisecond - it cannot, short of a fault - is waited for on its stream)
    for (uint32_t spins = 1; __atomic_load_n(&c->commit_host->flag, __ATOMIC_ACQUIRE) != seq; ++spins) {
        if ((spins & 255u) == 0 && std::chrono::steady_clock::now() - t_launch > std::chrono::microseconds(500)) {
            HIPCHK(c, wait_stream(c->stream));
            if (__atomic_load_n(&c->commit_host->flag, __ATOMIC_ACQUIRE) != seq) return fail(c, NHDFIT_E_HIP, "the commit kernel ended without publishing its placement");
            break;
        }
        __builtin_ia32_pause();
    }
    *place_out = c->commit_host->place;
    if (c->rec_lo == c->rec_hi) { c->rec_lo = node; c->rec_hi = node + 1; }
    else { c->rec_lo = std::min(c->rec_lo, node); c->rec_hi = std::max(c->rec_hi, node + 1); }
    return NHDFIT_OK;
}

int nhdfit_upload_origin(nhdfit_ctx* c, uint32_t first, uint32_t count, const nhdfit_origin* origin) {
    if (!c) return NHDFIT_E_INVAL;
    if (!count) return NHDFIT_OK;
    if (!origin) return fail(c, NHDFIT_E_INVAL, "NULL origin");
    if ((uint64_t)first + count > c->capacity) return fail(c, NHDFIT_E_INVAL, "upload [%u,%u) exceeds capacity %u", first, first + count, c->capacity);
    if (first > c->origin_hi) return fail(c, NHDFIT_E_INVAL, "origin records [%u,%u) are missing", c->origin_hi, first);
    HIPCHK(c, hipSetDevice(c->dev));
    HIPCHK(c, wait_stream(c->stream));                  // a delta kernel in flight may still write its records
    HIPCHK(c, hipMemcpy(c->origin.p + first, origin, count * sizeof *origin, hipMemcpyHostToDevice));
    c->origin_hi = std::max(c->origin_hi, first + count);
    return NHDFIT_OK;
}

int nhdfit_apply_deltas(nhdfit_ctx* c, const nhdfit_delta* deltas, uint32_t n, uint8_t* status_out) {
    if (!c) return NHDFIT_E_INVAL;
    if (!n) return NHDFIT_OK;
    if (!deltas || !status_out) return fail(c, NHDFIT_E_INVAL, "deltas / status_out is NULL");
    if (c->origin_hi < c->n) return fail(c, NHDFIT_E_STATE, "upload the origin records first (nhdfit_upload_origin)");
    // runs of one node, in array order inside a node (stable sort by node)
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) {
        if (deltas[i].node >= c->n) return fail(c, NHDFIT_E_INVAL, "delta %u: node %u out of range (%u nodes)", i, deltas[i].node, c->n);
        if (deltas[i].op < NHDFIT_DELTA_TAKE || deltas[i].op > NHDFIT_DELTA_SET_HUGEPAGES) return fail(c, NHDFIT_E_INVAL, "delta %u: unknown op %u", i, deltas[i].op);
        if (deltas[i].nic_n > NHDFIT_DELTA_MAX_NICS) return fail(c, NHDFIT_E_INVAL, "delta %u: %u NIC entries", i, (unsigned)deltas[i].nic_n);
        if (c->n_wide && c->wide_slot(deltas[i].node) >= 0)
            return fail(c, NHDFIT_E_INVAL, "delta %u: node %u is a wide node - its record is re-uploaded (nhdfit_wide_upload), deltas are for the planes", i, deltas[i].node);
        order[i] = i;
    }
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return deltas[x].node < deltas[y].node; });
    std::vector<nhdfit_delta> sorted(n);
    std::vector<uint32_t> run;
    uint32_t lo = ~0u, hi = 0;
    for (uint32_t i = 0; i < n; ++i) {
        sorted[i] = deltas[order[i]];
        if (i == 0 || sorted[i].node != sorted[i - 1].node) run.push_back(i);
        lo = std::min(lo, sorted[i].node); hi = std::max(hi, sorted[i].node + 1);
    }
    const uint32_t n_runs = (uint32_t)run.size();
    run.push_back(n);
    HIPCHK(c, hipSetDevice(c->dev));
    { int rc_ = sync_all(c); if (rc_) return rc_; }             // steps in flight on either pipe (fit, digest, the mapping phases flushed here) read
                                                                // the nodes as they were matched: the deltas wait for both pipes
    HIPCHK(c, c->deltas.reserve(n)); HIPCHK(c, c->delta_run.reserve(run.size())); HIPCHK(c, c->delta_status.reserve(n));
    HIPCHK(c, hipMemcpyAsync(c->deltas.p, sorted.data(), n * sizeof(nhdfit_delta), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->delta_run.p, run.data(), run.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    DeltaArgs da;
    memset(&da, 0, sizeof da);
    da.p0 = c->p0.p; da.p1 = c->p1.p; da.p2 = c->p2.p; da.p3 = c->p3.p; da.p4 = c->p4.p; da.det = c->det.p; da.origin = c->origin.p;
    da.deltas = c->deltas.p; da.run = c->delta_run.p; da.n_runs = n_runs; da.sigs = sig_table(c); da.status = c->delta_status.p;
    LAUNCH(c, k_delta, dim3((n_runs + 63) / 64), dim3(64), 0, c->stream, da);
    HIPCHK(c, hipGetLastError());
    std::vector<uint8_t> st(n);
    HIPCHK(c, hipMemcpyAsync(st.data(), c->delta_status.p, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, wait_stream(c->stream));                  // (also keeps `sorted` / `run` alive until the copies are done)
    for (uint32_t i = 0; i < n; ++i) status_out[order[i]] = st[i];
    if (c->rec_lo == c->rec_hi) { c->rec_lo = lo; c->rec_hi = hi; }
    else { c->rec_lo = std::min(c->rec_lo, lo); c->rec_hi = std::max(c->rec_hi, hi); }
    return NHDFIT_OK;
}

int nhdfit_download_nodes(nhdfit_ctx* c, uint32_t first, uint32_t count, nhdfit_plane0* p0, nhdfit_plane1* p1, nhdfit_plane2* p2,
                          nhdfit_plane3* p3, nhdfit_plane4* p4, nhdfit_detail* det) {
    if (!c) return NHDFIT_E_INVAL;
    if ((uint64_t)first + count > c->n) return fail(c, NHDFIT_E_INVAL, "download [%u,%u) exceeds the %u nodes of the mirror", first, first + count, c->n);
    HIPCHK(c, hipSetDevice(c->dev));
    HIPCHK(c, wait_stream(c->stream));
    if (p0) HIPCHK(c, hipMemcpy(p0, c->p0.p + first, count * sizeof *p0, hipMemcpyDeviceToHost));
    if (p1) HIPCHK(c, hipMemcpy(p1, c->p1.p + first, count * sizeof *p1, hipMemcpyDeviceToHost));
    if (p2) HIPCHK(c, hipMemcpy(p2, c->p2.p + first, count * sizeof *p2, hipMemcpyDeviceToHost));
    if (p3) HIPCHK(c, hipMemcpy(p3, c->p3.p + first, count * sizeof *p3, hipMemcpyDeviceToHost));
    if (p4) HIPCHK(c, hipMemcpy(p4, c->p4.p + first, count * sizeof *p4, hipMemcpyDeviceToHost));
    if (det) HIPCHK(c, hipMemcpy(det, c->det.p + first, count * sizeof *det, hipMemcpyDeviceToHost));
    return NHDFIT_OK;
}

int nhdfit_set_outputs(nhdfit_ctx* c, int want_bitmap, int want_map) {
    if (!c) return NHDFIT_E_INVAL;
    if ((want_bitmap != 0) != c->want_bitmap || (want_map != 0) != c->want_map) {   // steps in flight keep the old setting
        HIPCHK(c, hipSetDevice(c->dev));
        int rc_ = sync_all(c);
        if (rc_) return rc_;
    }
    c->want_bitmap = want_bitmap != 0;
    c->want_map = want_map != 0;
    return NHDFIT_OK;
}

int nhdfit_comm_unique_id(void* id128) {
    std::string err;
    if (!id128) return NHDFIT_E_INVAL;
    if (!g_rccl.load(err)) return fail(nullptr, NHDFIT_E_RCCL, "%s", err.c_str());
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, NHDFIT_E_RCCL, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r));
    memcpy(id128, &id, 128);
    return NHDFIT_OK;
}

int nhdfit_comm_init(nhdfit_ctx* c, int nranks, int rank, const void* id128) {
    if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, NHDFIT_E_INVAL, "bad communicator arguments");
    std::string err;
    if (!g_rccl.load(err)) return fail(c, NHDFIT_E_RCCL, "%s", err.c_str());
    if (c->comm) return fail(c, NHDFIT_E_STATE, "communicator already attached");
    HIPCHK(c, hipSetDevice(c->dev));
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) { c->comm = nullptr; return fail(c, NHDFIT_E_RCCL, "ncclCommInitRank: %s", g_rccl.GetErrorString(r)); }
    c->nranks = nranks;
    c->rank = rank;
    return NHDFIT_OK;
}

int nhdfit_comm_destroy(nhdfit_ctx* c) {
    if (!c) return NHDFIT_E_INVAL;
    if (c->comm) {
        { int rc_ = sync_all(c); if (rc_) return rc_; }
        g_rccl.CommDestroy(c->comm);
        c->comm = nullptr;
    }
    c->nranks = 1;
    c->rank = 0;
    return NHDFIT_OK;
}

int nhdfit_comm_rank(nhdfit_ctx* c, int* rank, int* nranks) {
    if (!c) return NHDFIT_E_INVAL;
    if (rank) *rank = c->comm ? c->rank : 0;
    if (nranks) *nranks = c->comm ? c->nranks : 1;
    return NHDFIT_OK;
}

int nhdfit_comm_sendrecv(nhdfit_ctx* c, const void* send_buf, size_t send_bytes, int dst, void* recv_buf, size_t recv_bytes, int src) {
    if (!c) return NHDFIT_E_INVAL;
    const int nranks = c->comm ? c->nranks : 1, me = c->comm ? c->rank : 0;
    const bool sending = dst >= 0 && send_bytes, receiving = src >= 0 && recv_bytes;
    if ((sending && (!send_buf || dst >= nranks)) || (receiving && (!recv_buf || src >= nranks))) return fail(c, NHDFIT_E_INVAL, "sendrecv: bad peer or buffer");
    if (!sending && !receiving) return NHDFIT_OK;
    if (!c->comm) {                                             // one GPU, no communicator: the self-exchange is a copy
        if (sending != receiving || send_bytes != recv_bytes) return fail(c, NHDFIT_E_INVAL, "sendrecv without a communicator: only a self-exchange of equal sizes");
        memmove(recv_buf, send_buf, recv_bytes);
        return NHDFIT_OK;
    }
    if (sending && receiving && dst == me && src == me && send_bytes != recv_bytes) return fail(c, NHDFIT_E_INVAL, "self-exchange of unequal sizes");
    HIPCHK(c, hipSetDevice(c->dev));
    // staged through two device buffers: RCCL moves device memory over xGMI; the reduce stream carries every collective of the
    // communicator in call order (the same order on every rank)
    HIPCHK(c, c->xfer_send.reserve(send_bytes ? send_bytes : 1));
    HIPCHK(c, c->xfer_recv.reserve(recv_bytes ? recv_bytes : 1));
    if (sending) HIPCHK(c, hipMemcpyAsync(c->xfer_send.p, send_buf, send_bytes, hipMemcpyHostToDevice, c->s_red));
    ncclResult_t r = g_rccl.GroupStart();
    if (r == ncclSuccess && sending) r = (c->known_idle = false, g_rccl).Send(c->xfer_send.p, send_bytes, ncclUint8, dst, c->comm, c->s_red);
    if (r == ncclSuccess && receiving) r = (c->known_idle = false, g_rccl).Recv(c->xfer_recv.p, recv_bytes, ncclUint8, src, c->comm, c->s_red);
    const ncclResult_t r2 = g_rccl.GroupEnd();
    if (r != ncclSuccess || r2 != ncclSuccess) return fail(c, NHDFIT_E_RCCL, "ncclSend / ncclRecv: %s", g_rccl.GetErrorString(r != ncclSuccess ? r : r2));
    if (receiving) HIPCHK(c, hipMemcpyAsync(recv_buf, c->xfer_recv.p, recv_bytes, hipMemcpyDeviceToHost, c->s_red));
    HIPCHK(c, wait_stream(c->s_red));
    return NHDFIT_OK;
}

int nhdfit_comm_allreduce_sum_u8(nhdfit_ctx* c, void* buf, size_t bytes) {
    if (!c) return NHDFIT_E_INVAL;
    if (!c->comm || !bytes) return NHDFIT_OK;
    if (!buf) return fail(c, NHDFIT_E_INVAL, "allreduce: no buffer");
    HIPCHK(c, hipSetDevice(c->dev));
    HIPCHK(c, c->xfer_send.reserve(bytes));
    HIPCHK(c, hipMemcpyAsync(c->xfer_send.p, buf, bytes, hipMemcpyHostToDevice, c->s_red));
    ncclResult_t r = (c->known_idle = false, g_rccl).AllReduce(c->xfer_send.p, c->xfer_send.p, bytes, ncclUint8, ncclSum, c->comm, c->s_red);
    if (r != ncclSuccess) return fail(c, NHDFIT_E_RCCL, "ncclAllReduce(uint8, sum): %s", g_rccl.GetErrorString(r));
    HIPCHK(c, hipMemcpyAsync(buf, c->xfer_send.p, bytes, hipMemcpyDeviceToHost, c->s_red));
    HIPCHK(c, wait_stream(c->s_red));
    return NHDFIT_OK;
}

// ---- one process, several GPUs: the reference's single scheduler thread behind FindNode on all devices ------------
struct nhdfit_group {
    std::vector<nhdfit_ctx*> ctx;
    std::vector<ncclComm_t> comm;        // empty: scores are max-reduced on the host (NHDFIT_GROUP_REDUCE=host, or one device)
    std::string err;
};

int nhdfit_group_create(const int* devices, int n, nhdfit_group** out) {
    if (!devices || n < 1 || !out) return fail(nullptr, NHDFIT_E_INVAL, "bad device list");
    *out = nullptr;
    nhdfit_group* g = new (std::nothrow) nhdfit_group();
    if (!g) return fail(nullptr, NHDFIT_E_NOMEM, "out of host memory");
    for (int k = 0; k < n; ++k) {
        nhdfit_ctx* c = nullptr;
        int rc = nhdfit_create(devices[k], &c);
        if (rc) { for (auto* x : g->ctx) nhdfit_destroy(x); delete g; return rc; }
        g->ctx.push_back(c);
    }
    const char* mode = tune_env("NHDFIT_GROUP_REDUCE");
    if (n > 1 && !(mode && !strcmp(mode, "host"))) {
        std::string err;
        if (!g_rccl.load(err)) { for (auto* x : g->ctx) nhdfit_destroy(x); delete g; return fail(nullptr, NHDFIT_E_RCCL, "%s", err.c_str()); }
        g->comm.resize(n);
        ncclResult_t r = g_rccl.CommInitAll(g->comm.data(), n, devices);      // one communicator per device, one process
        if (r != ncclSuccess) {
            for (auto* x : g->ctx) nhdfit_destroy(x);
            delete g;
            return fail(nullptr, NHDFIT_E_RCCL, "ncclCommInitAll: %s", g_rccl.GetErrorString(r));
        }
    }
    *out = g;
    return NHDFIT_OK;
}

void nhdfit_group_destroy(nhdfit_group* g) {
    if (!g) return;
    for (size_t k = 0; k < g->ctx.size(); ++k) {
        (void)hipSetDevice(g->ctx[k]->dev);
        (void)hipDeviceSynchronize();
        if (k < g->comm.size() && g->comm[k]) g_rccl.CommDestroy(g->comm[k]);
    }
    for (auto* c : g->ctx) nhdfit_destroy(c);
    delete g;
}

int nhdfit_group_size(nhdfit_group* g) { return g ? (int)g->ctx.size() : 0; }
nhdfit_ctx* nhdfit_group_ctx(nhdfit_group* g, int k) { return g && k >= 0 && k < (int)g->ctx.size() ? g->ctx[k] : nullptr; }
const char* nhdfit_group_last_error(nhdfit_group* g) { return g ? g->err.c_str() : ""; }

// Mode A over every shard of the group: requests replicated, digest + fit per device, ONE all-reduce(max) of the P packed
// scores over xGMI (ncclGroupStart / ncclAllReduce per device / ncclGroupEnd), then the owner of each winner maps it.
// cand[k]: optional candidate mask of shard k.  map_out[p] is taken from the owner; owner_out[p] = its shard or -1.
int nhdfit_group_find(nhdfit_group* g, const nhdfit_req* reqs, uint32_t P, double now, const uint64_t* const* cand,
                      uint64_t* score_out, nhdfit_mapping* map_out, int32_t* owner_out) {
    if (!g || !reqs || !P || !score_out) return NHDFIT_E_INVAL;
    const size_t n = g->ctx.size();
    auto gfail = [&](nhdfit_ctx* c, int rc) { g->err = c ? c->err : std::string("group error"); return rc; };
    for (size_t k = 0; k < n; ++k) {                                     // digest + fit on every device, no host wait in between
        nhdfit_ctx* c = g->ctx[k];
        int rc = nhdfit_stage_requests(c, reqs, P);
        if (!rc && cand && cand[k]) rc = stage_cand(c, cand[k]);
        if (!rc && c->n) rc = nhdfit_enqueue_step(c, now);
        if (rc) return gfail(c, rc);
    }
    std::vector<std::vector<uint64_t>> host_scores;
    if (!g->comm.empty()) {
        ncclResult_t r = g_rccl.GroupStart();
        for (size_t k = 0; k < n && r == ncclSuccess; ++k) {
            nhdfit_ctx* c = g->ctx[k];
            Pipe& p = c->pipe[0];                                        // (a freshly staged batch's one step runs on pipe 0)
            if (hipSetDevice(c->dev) != hipSuccess) { r = ncclSystemError; break; }
            const int b = p.n_fit ? (int)((p.n_fit - 1) % kBufs) : 0;
            if (!c->n) {                                                 // a shard without nodes contributes "no feasible node"
                if (p.score[b].reserve(P) != hipSuccess || hipMemsetAsync(p.score[b].p, 0, (size_t)P * 8, c->stream) != hipSuccess) { r = ncclSystemError; break; }
            }
            r = (c->known_idle = false, g_rccl).AllReduce(p.score[b].p, p.score[b].p, P, ncclUint64, ncclMax, g->comm[k], c->stream);
        }
        ncclResult_t r2 = g_rccl.GroupEnd();
        if (r != ncclSuccess || r2 != ncclSuccess) { g->err = std::string("ncclAllReduce (group): ") + g_rccl.GetErrorString(r != ncclSuccess ? r : r2); return NHDFIT_E_RCCL; }
    } else if (n > 1) {                                                  // host max-reduce (debug / test path): D2H, max, H2D
        host_scores.resize(n);
        std::vector<uint64_t> best(P, 0), tmp(P);
        for (size_t k = 0; k < n; ++k) {
            nhdfit_ctx* c = g->ctx[k];
            if (!c->n) continue;
            Pipe& p = c->pipe[0];
            HIPCHK(c, hipSetDevice(c->dev));
            HIPCHK(c, wait_stream(c->stream));
            const int b = (int)((p.n_fit - 1) % kBufs);
            HIPCHK(c, hipMemcpy(tmp.data(), p.score[b].p, (size_t)P * 8, hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < P; ++i) best[i] = std::max(best[i], tmp[i]);
        }
        for (size_t k = 0; k < n; ++k) {
            nhdfit_ctx* c = g->ctx[k];
            if (!c->n) continue;
            Pipe& p = c->pipe[0];
            HIPCHK(c, hipSetDevice(c->dev));
            const int b = (int)((p.n_fit - 1) % kBufs);
            HIPCHK(c, hipMemcpy(p.score[b].p, best.data(), (size_t)P * 8, hipMemcpyHostToDevice));
        }
    }
    // mapping roles per device (each maps the winners it owns), then collect
    std::vector<nhdfit_mapping> maps(P);
    std::vector<uint64_t> sc(P);
    bool have_score = false;
    if (map_out) memset(map_out, 0, (size_t)P * sizeof(nhdfit_mapping));
    if (owner_out) for (uint32_t i = 0; i < P; ++i) owner_out[i] = -1;
    for (size_t k = 0; k < n; ++k) {
        nhdfit_ctx* c = g->ctx[k];
        if (!c->n) continue;
        int rc = nhdfit_fetch(c, sc.data(), nullptr, map_out ? maps.data() : nullptr);
        if (rc) return gfail(c, rc);
        if (!have_score) { memcpy(score_out, sc.data(), (size_t)P * 8); have_score = true; }
        for (uint32_t i = 0; i < P; ++i) {
            if (!sc[i]) continue;
            const uint64_t gi = NHDFIT_SCORE_INDEX(sc[i]);
            if (gi >= c->global_base && gi < c->global_base + c->n) {
                if (map_out) map_out[i] = maps[i];
                if (owner_out) owner_out[i] = (int32_t)k;
            }
        }
    }
    if (!have_score) memset(score_out, 0, (size_t)P * 8);
    return NHDFIT_OK;
}

int nhdfit_get_stats(nhdfit_ctx* c, nhdfit_stats* out) {
    if (!c || !out) return NHDFIT_E_INVAL;
    if (c->ev_pending) {                                    // sampled launches whose events have not been read yet (waits for them if need be)
        HIPCHK(c, hipSetDevice(c->dev));
        int rc = drain_events(c);
        if (rc) return rc;
    }
    *out = c->stats;
    return NHDFIT_OK;
}

int nhdfit_reset_stats(nhdfit_ctx* c) {
    if (!c) return NHDFIT_E_INVAL;
    int rc = nhdfit_sync(c);
    if (!rc) rc = drain_events(c);
    if (rc) return rc;
    c->stats.launches = 0;
    c->stats.fit_ms_total = 0;
    c->stats.big_nic_steps_max = 0;
    return NHDFIT_OK;
}

}  // extern "C"
