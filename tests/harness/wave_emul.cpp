// wave_emul.cpp - TEST INFRASTRUCTURE.  Runs DEVICE SOURCE TEXT on the host: the wavefront form of the commit step
// (nhd_amd/csrc/seq2_kernel.h, "the commit step with the wavefront's lanes": lowest_bits_wave, take_batch_wave, pool_key_wave,
// sig_keys_wave, commit_node_wave - what k_decide's speculators and workers execute) is cut out of the kernel header at build
// time (tests/harness/__init__.py: the lines between the section's heading and the next one, unmodified) and compiled here with
// a 64-lane wavefront emulated by 64 threads: __ballot is a barrier + an OR of the lanes' predicates, the wavefront barrier is a
// barrier, `lane` is the thread's number; the LDS copies (`s`, `d`, `out`) are shared by the threads as they are by the lanes.
// The functions' control flow is wave-uniform (every lane reaches every ballot), which is what the emulation relies on and, by
// not dead-locking, checks.  tests/test_wave_commit_emulation.py compares the result with the scalar form the host twin and
// nhdfit_commit's k_commit use (commit_core.h commit_node) - state, detail, placement record and status, byte for byte.
// NOT part of libnhdfit.so; nothing in nhd_amd/ loads it.
#include <atomic>
#include <barrier>
#include <cstddef>
#include <cstring>
#include <thread>
#include <vector>
#include "../../nhd_amd/csrc/seq_core.h"
#include "../../nhd_amd/csrc/wide_core.h"

using namespace nhdfit;

namespace emu {
constexpr int kLanes = 64;
std::barrier<> bar(kLanes);
std::atomic<uint64_t> acc[2];
thread_local uint32_t t_lane = 0, t_count = 0;
inline uint64_t ballot(bool p) {
    const uint32_t k = t_count++ & 1u;
    if (p) acc[k].fetch_or(1ull << t_lane, std::memory_order_acq_rel);
    bar.arrive_and_wait();
    const uint64_t m = acc[k].load(std::memory_order_acquire);
    bar.arrive_and_wait();
    if (t_lane == 0) acc[k].store(0, std::memory_order_release);     // (next written two ballots from now: a barrier lies between)
    return m;
}
inline void wave_barrier() { bar.arrive_and_wait(); }
int slots[kLanes];
inline int readlane(int v, int l) {                                  // every lane calls it (wave-uniform control flow)
    slots[t_lane] = v;
    bar.arrive_and_wait();
    const int r = slots[l & (kLanes - 1)];
    bar.arrive_and_wait();
    return r;
}
}  // namespace emu

#define __device__
#define __forceinline__ inline
#define __ballot(p) emu::ballot(p)
#define __popcll(x) __builtin_popcountll(x)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_wave_barrier() emu::wave_barrier()
#define __builtin_amdgcn_readlane(v, l) emu::readlane((v), (l))
#define __builtin_amdgcn_readfirstlane(v) emu::readlane((v), 0)      // (all 64 lanes are active in every routine run here)
#define __noinline__ __attribute__((noinline))

namespace {
#include "_wave_map_block.inc"       // seq_kernel.h: "wave-cooperative forms of the mapping arithmetic" (map_on_state_wave and its helpers)
#include "_wave_commit_block.inc"    // seq2_kernel.h: "the commit step with the wavefront's lanes"
#include "_wave_bigmap_block.inc"    // big_kernel.h: "the winner's mapping with the wavefront's lanes" (wide_map_wave)

// the mapping tables as the device builds them (k_build_asc / k_build_choose / the set-layout state machine)
const AscEntry* asc_table() {
    static std::vector<AscEntry> t;
    if (t.empty()) {
        t.resize(kAscEntries);
        for (int len = 1; len <= 4; ++len)
            for (uint32_t sub = 0; sub < (1u << (1u << len)); ++sub) t[kAscOffset[len] + sub] = asc_entry_build(len, sub);
    }
    return t.data();
}
const uint8_t* choose_table() {
    static std::vector<uint8_t> t;
    if (t.empty()) {
        t.resize(kChooseEntries);
        for (uint32_t e = 0; e < kChooseEntries; ++e) t[e] = choose_entry_build(asc_table(), e);
    }
    return t.data();
}
const SetStates& set_states() {
    static std::vector<uint64_t> info;
    static std::vector<uint32_t> next, asc;
    static SetStates t{};
    if (info.empty()) {
        build_set_states(info, next, asc);
        t = SetStates{info.data(), next.data(), asc.data(), (uint32_t)info.size()};
    }
    return t;
}
}

extern "C" {

// nhdfit_commit's arithmetic in its wavefront form on one node record (same arguments as host_harness.cpp hh_commit + the
// dictionary's class count); returns the status every lane agreed on, -100 if the lanes disagreed
int we_commit(nhdfit_plane0* p0, nhdfit_plane1* p1, nhdfit_plane2* p2, nhdfit_plane3* p3, nhdfit_plane4* p4, nhdfit_detail* det,
              const nhdfit_req* req, const nhdfit_mapping* map, double busy_time,
              const uint32_t* sig_off, uint32_t nsig, const uint32_t* pool_off, const uint8_t* pool_glimit, const nhdfit_cc* cc,
              uint32_t ncls, nhdfit_placement* out, int form) {
    uint32_t slots = 64;
    while (slots < 4 * nsig) slots <<= 1;
    std::vector<uint64_t> skeys(slots, 0);
    std::vector<uint32_t> sids(slots, 0);
    for (uint32_t sg = 1; sg < nsig; ++sg) {
        uint64_t key = 0;
        for (uint32_t pl = sig_off[sg]; pl < sig_off[sg + 1]; ++pl) {
            uint8_t cnt[NHDFIT_MAX_CLASSES] = {0};
            for (uint32_t k = pool_off[pl]; k < pool_off[pl + 1]; ++k) cnt[cc[k].cls & 15u] = cc[k].cnt;
            key = sig_key_add(key, pool_key(pool_glimit[pl], cnt));
        }
        if (!key) continue;
        uint32_t sl = (uint32_t)mix64(key) & (slots - 1);
        while (skeys[sl] != 0 && skeys[sl] != key) sl = (sl + 1) & (slots - 1);
        skeys[sl] = key; sids[sl] = sg;
    }
    const SigTable sigs{skeys.data(), sids.data(), slots - 1};
    NodeState st{*p0, *p1, *p2, *p3, *p4};                       // the wavefront's LDS copies
    alignas(16) nhdfit_detail dd = *det;
    nhdfit_placement pl;
    std::memset(&pl, 0xA5, sizeof pl);                           // (the wavefront form initialises the record itself)
    int status[emu::kLanes];
    emu::acc[0] = emu::acc[1] = 0;
    bool with_gpus = false;
    for (uint32_t g = 0; g < req->n_groups && g < (uint32_t)kMaxG; ++g) with_gpus = with_gpus || req->gpus[g] != 0;
    std::vector<std::thread> lanes;
    for (int i = 0; i < emu::kLanes; ++i)
        lanes.emplace_back([&, i] {
            emu::t_lane = (uint32_t)i; emu::t_count = 0;
            if (form == 2 && !with_gpus) {
                // the two-stage form k_decide's speculators run for a pod without GPUs: stage 1 (summary) in place, then stage 2 (picks)
                // on the free sets stage 1 found; thread 1 loses the bits stage 2 returns (the kernel: an AND into the mirror)
                uint64_t f0 = 0, f1 = 0, c0 = 0, c1 = 0;
                const bool smt_node = (st.p2.flags & NHDFIT_NF_SMT) != 0;
                int s1 = commit_summary_wave(st, dd, *req, *map, busy_time, sigs, ncls, (uint32_t)i, f0, f1);
                const int s2 = commit_picks_wave(f0, f1, smt_node, *req, *map, pl, (uint32_t)i, c0, c1);
                if (s2 == kCommitWouldRaise) s1 = kCommitWouldRaise;
                if (i == 0) { st.p1.t1[0] &= ~c0; st.p1.t1[1] &= ~c1; pl.status = (uint8_t)s1; }
                emu::wave_barrier();
                status[i] = s1;
            } else
            status[i] = commit_node_wave(st, dd, *req, *map, busy_time, sigs, ncls, pl, (uint32_t)i);
        });
    for (auto& t : lanes) t.join();
    for (int i = 1; i < emu::kLanes; ++i) if (status[i] != status[0]) return -100;
    *p0 = st.p0; *p1 = st.p1; *p2 = st.p2; *p3 = st.p3; *p4 = st.p4;
    *det = dd;
    *out = pl;
    return status[0];
}

// The mapping FindNode returns for pod `req` on the node in state (planes, det): seq_core.h map_on_state (scalar: the host twin's)
// against seq_kernel.h map_on_state_wave (the sequential kernels' wavefront form, emulated lanes).  nic_bits: the NIC-feasible
// NUMA assignments, here from the scalar NIC walk itself (on the device: the cold R rows of the pod's tile image).
// nic_bits_in >= 0: use these bits instead.  tables: 0 = the set model alone, 1 = with the ascending-set table, 2 = with every table the device uses.
// Returns 0 when both forms agree (ok flag; mapping when ok), 1 otherwise, -100 if the lanes disagree among themselves.
int we_map_on_state(const nhdfit_plane0* p0, const nhdfit_plane1* p1, const nhdfit_plane2* p2, const nhdfit_plane3* p3, const nhdfit_plane4* p4,
                    const nhdfit_detail* det, const nhdfit_req* req, const double* caps, int tables, int64_t nic_bits_in,
                    nhdfit_mapping* scalar_out, nhdfit_mapping* wave_out, int* ok_out, int form) {
    const NodeState st{*p0, *p1, *p2, *p3, *p4};
    const nhdfit_detail dd = *det;
    const WinnerState w = state_view(st, dd, caps);
    const int G = (int)req->n_groups, U = w.U;
    uint32_t nic_bits = 0;
    if (nic_bits_in >= 0) nic_bits = (uint32_t)nic_bits_in;            // the caller's (the lone-pod form's masks: fit_core.h lone_nic_bits)
    else if (G >= 1 && G <= kMaxG)
        for (uint32_t p = 0; p < (1u << G); ++p) {
            if (U == 1 && p) break;
            int8_t idx[kMaxG];
            if (first_nic_choice(*req, w, p, req->map_type == NHDFIT_MAP_PCI, idx)) nic_bits |= 1u << p;
        }
    MapTables mt{nullptr, nullptr, SetStates{nullptr, nullptr, nullptr, 0}};
    if (tables >= 1) mt.asc = asc_table();
    if (tables >= 2) { mt.choose_tab = choose_table(); mt.st = set_states(); }
    nhdfit_mapping ms;
    std::memset(&ms, 0, sizeof ms);
    const bool ok_s = map_on_state(*req, st, dd, caps, nic_bits, mt, ms);
    nhdfit_mapping mw[emu::kLanes];
    bool ok_w[emu::kLanes];
    emu::acc[0] = emu::acc[1] = 0;
    std::vector<std::thread> lanes;
    for (int i = 0; i < emu::kLanes; ++i)
        lanes.emplace_back([&, i] {
            emu::t_lane = (uint32_t)i; emu::t_count = 0;
            ok_w[i] = map_on_state_wave(*req, st, dd, caps, nic_bits, mt, (uint32_t)i, mw[i]);
        });
    for (auto& t : lanes) t.join();
    for (int i = 1; i < emu::kLanes; ++i)
        if (ok_w[i] != ok_w[0] || std::memcmp(&mw[i], &mw[0], sizeof(nhdfit_mapping)) != 0) return -100;
    *scalar_out = ms; *wave_out = mw[0];
    *ok_out = (ok_s ? 1 : 0) | (ok_w[0] ? 2 : 0);
    if (ok_s != ok_w[0]) return 1;
    if (!ok_s) return 0;
    for (int g = 0; g < G; ++g)
        if (ms.gpu[g] != mw[0].gpu[g] || ms.nic_numa[g] != mw[0].nic_numa[g] || ms.nic_idx[g] != mw[0].nic_idx[g]) return 1;
    for (int g = 0; g <= G; ++g) if (ms.cpu[g] != mw[0].cpu[g]) return 1;
    return ms.valid == mw[0].valid ? 0 : 1;
}

// A big request's mapping on one winner: wide_core.h wide_map (one thread: the host twin's and the device's form for nodes of more than two
// NUMA nodes) against big_kernel.h wide_map_wave (lane = tuple, emulated lanes).  `wide` != NULL: that record, else node 0 of the planes.
// Returns 0 when return code and mapping agree, 1 otherwise, -100 if the lanes disagree on the code.
int we_big_map(const nhdfit_plane0* p0, const nhdfit_plane1* p1, const nhdfit_plane2* p2, const nhdfit_plane3* p3, const nhdfit_plane4* p4,
               const nhdfit_detail* det, const nhdfit_wide_node* wide, const nhdfit_big_req* r, const double* caps, const nhdfit_wide_share* share,
               nhdfit_big_mapping* scalar_out, nhdfit_big_mapping* wave_out, int* rc_out) {
    static BigWaveLds t;                                              // (the block's LDS record)
    if (wide) t.view = *wide; else wide_view(*p0, *p1, *p2, *p3, *p4, *det, 0, t.view);
    t.req = *r;
    const uint32_t U = t.view.numa_nodes ? t.view.numa_nodes : 1, G = r->n_groups <= NHDFIT_BIG_MAX_GROUPS ? r->n_groups : NHDFIT_BIG_MAX_GROUPS;
    if (U > 2) return -2;                                             // (the kernel takes the one-thread form there)
    const WideCaps wc(caps, wide ? share : nullptr);
    const int32_t sg = (int32_t)wide_table_slots(wide_ipow(2, G)), sc = (int32_t)wide_table_slots(wide_ipow(2, G + 1));   // sized as the device sizes them (two NUMA nodes)
    std::vector<int32_t> scratch(big_scratch_words(2, G)), tables(big_scratch_words(2, G));
    nhdfit_big_mapping ms;
    const int rs = wide_map(t.view, *r, wc, scratch.data(), ms, sg, sc);
    nhdfit_big_mapping mw[emu::kLanes];
    int rw[emu::kLanes];
    emu::acc[0] = emu::acc[1] = 0;
    std::vector<std::thread> lanes;
    for (int i = 0; i < emu::kLanes; ++i)
        lanes.emplace_back([&, i] {
            emu::t_lane = (uint32_t)i; emu::t_count = 0;
            rw[i] = wide_map_wave(t, wc, tables.data(), mw[i], sg, sc, (uint32_t)i);
        });
    for (auto& th : lanes) th.join();
    for (int i = 1; i < emu::kLanes; ++i) if (rw[i] != rw[0]) return -100;
    *scalar_out = ms; *wave_out = mw[0];
    rc_out[0] = rs; rc_out[1] = rw[0];
    if (rs != rw[0]) return 1;
    return std::memcmp(&ms, &mw[0], sizeof ms) == 0 ? 0 : 1;
}

}  // extern "C"
