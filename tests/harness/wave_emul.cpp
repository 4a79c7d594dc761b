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
#include <cstring>
#include <thread>
#include <vector>
#include "../../nhd_amd/csrc/seq_core.h"

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
}  // namespace emu

#define __device__
#define __forceinline__ inline
#define __ballot(p) emu::ballot(p)
#define __popcll(x) __builtin_popcountll(x)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_wave_barrier() emu::wave_barrier()

namespace {
#include "_wave_commit_block.inc"
}

extern "C" {

// nhdfit_commit's arithmetic in its wavefront form on one node record (same arguments as host_harness.cpp hh_commit + the
// dictionary's class count); returns the status every lane agreed on, -100 if the lanes disagreed
int we_commit(nhdfit_plane0* p0, nhdfit_plane1* p1, nhdfit_plane2* p2, nhdfit_plane3* p3, nhdfit_plane4* p4, nhdfit_detail* det,
              const nhdfit_req* req, const nhdfit_mapping* map, double busy_time,
              const uint32_t* sig_off, uint32_t nsig, const uint32_t* pool_off, const uint8_t* pool_glimit, const nhdfit_cc* cc,
              uint32_t ncls, nhdfit_placement* out) {
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
    nhdfit_detail dd = *det;
    nhdfit_placement pl;
    std::memset(&pl, 0xA5, sizeof pl);                           // (the wavefront form initialises the record itself)
    int status[emu::kLanes];
    emu::acc[0] = emu::acc[1] = 0;
    std::vector<std::thread> lanes;
    for (int i = 0; i < emu::kLanes; ++i)
        lanes.emplace_back([&, i] {
            emu::t_lane = (uint32_t)i; emu::t_count = 0;
            status[i] = commit_node_wave(st, dd, *req, *map, busy_time, sigs, ncls, pl, (uint32_t)i);
        });
    for (auto& t : lanes) t.join();
    for (int i = 1; i < emu::kLanes; ++i) if (status[i] != status[0]) return -100;
    *p0 = st.p0; *p1 = st.p1; *p2 = st.p2; *p3 = st.p3; *p4 = st.p4;
    *det = dd;
    *out = pl;
    return status[0];
}

}  // extern "C"
