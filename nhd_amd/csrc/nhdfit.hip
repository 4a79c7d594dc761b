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
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "seq_core.h"

using namespace nhdfit;

namespace {

// ------------------------------------------------------------------------------------------------
// device code
// ------------------------------------------------------------------------------------------------
struct DictView {
    const double* caps;
    uint32_t ncls;
    const uint64_t* group_sets;
    SigDict sig;
    // the same dictionary as ONE stream of 16-bit words the digest role stages in LDS (walking the three CSR levels in
    // global memory costs a dependent scalar load per level, pool and class - 2-3 us per signature):
    // [0, nsig]: word offset of each signature's record behind the table; record = { #pools, per pool: glimit << 8 | #cc,
    // then #cc x (cls << 8 | cnt) }
    const uint16_t* flat;
    uint32_t flat_words;             // 0: not available (the stream would not fit 16-bit offsets)
};

// v_writelane_b32 (SGPR -> one lane of a VGPR).  This clang has no __builtin_amdgcn_writelane; the
// asm label binds the declaration straight to the LLVM intrinsic, as the ROCm device libs do.
extern "C" __device__ int nhd_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

struct PaddedReq { nhdfit_req r; uint32_t pad; };             // LDS copies, 33-word stride: lane j -> bank j

// Coalesced copy of up to 64 consecutive request records (from pod0) into LDS, zero (= invalid) past P.
template <int THREADS>
__device__ __forceinline__ void stage_requests_lds(const nhdfit_req* __restrict__ reqs, uint32_t pod0, uint32_t P, PaddedReq* s_req) {
    constexpr uint32_t kParts = sizeof(nhdfit_req) / 16;
    const uint32_t live = pod0 < P ? (P - pod0 < (uint32_t)kTile ? P - pod0 : (uint32_t)kTile) : 0u;
    const uint4* src = reinterpret_cast<const uint4*>(reqs + pod0);
    for (uint32_t c = threadIdx.x; c < kTile * kParts; c += THREADS) {
        const uint32_t j = c / kParts;
        const uint4 v = j < live ? src[c] : make_uint4(0u, 0u, 0u, 0u);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&s_req[j]) + (c % kParts) * 4;
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
}

template <class T>
__device__ __forceinline__ T* carve(uint8_t*& p, size_t count) {      // 16-byte aligned slices of a block's LDS
    T* r = reinterpret_cast<T*>(p);
    p += (count * sizeof(T) + 15) & ~size_t(15);
    return r;
}
constexpr size_t lds_slice(size_t bytes) { return (bytes + 15) & ~size_t(15); }

struct DigestArgs {
    const nhdfit_req* reqs;          // class-sorted order (as staged)
    uint32_t P;
    DictView d;
    Layout L[kWClasses];             // image layout per row width W = 2 << class
    uint32_t pitch;                  // bytes between tile images
    uint8_t* tabs;                   // out: tile images
    PodHeader* hdr;                  // out: [tiles*64]
    unsigned long long* score;       // out: zeroed (the fit role accumulates with atomicMax)
    const uint64_t* xcls;            // interned (NUMA, free GPUs, signature) classes of the mirror: key of X row k
    const uint32_t* nx;              // number of classes
};
constexpr uint32_t kDictLdsWords = 6144;             // 12 KB for the staged signature stream (c5: 151 signatures = 1.5 K words)
constexpr size_t kDigestLds = lds_slice(kTile * sizeof(PaddedReq)) + lds_slice(kTile * sizeof(PodSums)) +
                              lds_slice(kTile * NHDFIT_MAX_CLASSES * (kMaxG + 1) * sizeof(uint16_t)) + lds_slice(kTile * sizeof(PodHeader)) +
                              lds_slice(kDictLdsWords * sizeof(uint16_t));
constexpr uint32_t kWcParts = 4;                     // blocks per tile that share its CPU rows (free-core count c = part mod 4)
constexpr uint32_t kDigestParts = 1 + kWcParts;      // part 0 = GPU / NIC rows (cold section + X), parts 1..4 = CPU rows, the last one also HP / GX

// Request digest, kDigestParts blocks per 64-pod tile: per-pod subset sums / NIC covers in LDS, then the table rows
// (lane = pod, one ballot per assignment).  The role is a chain of dependent phases, not a lot of work: it is cut
// into parts by table so that the chain of each block stays short.
template <int THREADS>
__device__ __forceinline__ void role_digest(const DigestArgs& a, uint32_t blk, uint8_t* lds) {
    PaddedReq* s_req = carve<PaddedReq>(lds, kTile);
    PodSums* s_sum = carve<PodSums>(lds, kTile);
    uint16_t (*s_cover)[NHDFIT_MAX_CLASSES][kMaxG + 1] =
        reinterpret_cast<uint16_t (*)[NHDFIT_MAX_CLASSES][kMaxG + 1]>(carve<uint16_t>(lds, kTile * NHDFIT_MAX_CLASSES * (kMaxG + 1)));
    PodHeader* s_hdr = carve<PodHeader>(lds, kTile);
    uint16_t* s_flat = carve<uint16_t>(lds, kDictLdsWords);

    const uint32_t tile = blk / kDigestParts, part = blk % kDigestParts;
    const uint32_t tid = threadIdx.x;
    uint8_t* img = a.tabs + (size_t)tile * a.pitch;

    stage_requests_lds<THREADS>(a.reqs, tile * kTile, a.P, s_req);
    __syncthreads();
    constexpr uint32_t NW = THREADS / 64;
    const uint32_t wave = tid >> 6, lane = tid & 63;
    if (tid < kTile) {
        const nhdfit_req& r = s_req[tid].r;
        const PodHeader h = pod_header(r);
        s_hdr[tid] = h;
        PodSums& ps = s_sum[tid];
        ps.G = r.n_groups; ps.W = 1u << (r.n_groups & 7u); ps.full = ps.W - 1;
        ps.misc_smt = r.misc_smt; ps.misc_nosmt = r.misc_nosmt;
        if (part == 0) {
            const uint32_t pod = tile * kTile + tid;
            a.hdr[pod] = h;
            if (pod < a.P) a.score[pod] = 0;
        }
    }
    {   // subset sums (pod_sums), one subset per (wavefront, lane = pod) instead of 16 in a row on one wavefront
        const nhdfit_req& r = s_req[lane].r;
        const bool ok = req_valid(r);
        for (uint32_t S = wave; S < (1u << kMaxG); S += NW) {
            if (!ok || S >= (1u << r.n_groups)) continue;
            uint32_t g = 0, x = 0, y = 0;
            for (uint32_t i = 0; i < r.n_groups; ++i)
                if (S >> i & 1) { g += r.gpus[i]; x += r.cpu_smt[i]; y += r.cpu_nosmt[i]; }
            s_sum[lane].gpu[S] = g; s_sum[lane].cpu_smt[S] = x; s_sum[lane].cpu_nosmt[S] = y;
        }
    }
    __syncthreads();

    const bool valid = (s_hdr[lane].flags & kPodValid) != 0;
    // the tile's row width: 2^(largest group count among its pods) - the same rule the host applies when it
    // builds the fit role's work items (tile_wclass)
    const uint32_t my_g = valid ? (s_hdr[lane].flags >> kPodGroupsShift) & 7u : 0u;
    const uint32_t wcls = __ballot(my_g >= 4) ? 3u : __ballot(my_g == 3) ? 2u : __ballot(my_g == 2) ? 1u : 0u;
    const Layout& L = a.L[wcls];
    const uint32_t W = L.W;
    uint8_t* hot = img + L.off_hot;
    // bit-sliced row: lane = pod holds its 16-bit entry (bit p = assignment p passes), one ballot per assignment
    // turns the 64 entries into the row's W words (bit j of word p = assignment p of pod j passes)
    auto emit_row = [&](uint8_t* row, uint32_t v) {
        unsigned long long mine = 0;
        for (uint32_t p = 0; p < W; ++p) {
            const unsigned long long word = __ballot(v >> p & 1);
            if (lane == p) mine = word;
        }
        if (lane < W) *reinterpret_cast<unsigned long long*>(row + lane * 8) = mine;
    };

    if (part != 0) {
        // CPU records WC[u][smt][c] = {m=0 row, m=1 row}: for a pod, socket, SMT mode and misc placement the entry is
        // { p : demand_p <= c } - the demands are read once per (socket, misc, smt) group and swept over c in
        // registers (one group per wavefront) instead of being re-read from LDS for each of the rows of the group
        for (uint32_t g = wave; g < 8; g += NW) {
            const uint32_t u = g >> 2, m = (g >> 1) & 1, smt = g & 1;
            uint32_t t[1 << kMaxG];
#pragma unroll
            for (uint32_t p = 0; p < (1u << kMaxG); ++p) {
                t[p] = 0xFFFFFFFFu;
                if (valid && p < s_sum[lane].W) {
                    const uint32_t* sum = smt ? s_sum[lane].cpu_smt : s_sum[lane].cpu_nosmt;
                    const uint32_t extra = m ? (smt ? s_sum[lane].misc_smt : s_sum[lane].misc_nosmt) : 0;
                    t[p] = sum[u ? p : (~p & s_sum[lane].full)] + extra;
                }
            }
            uint8_t* base = hot + (u ? L.hot_wc1 : L.hot_wc0) + smt * L.fc_dim * L.wc_stride + m * L.row;
            for (uint32_t c = part - 1; c < L.fc_dim; c += kWcParts) {
                uint32_t v = 0;
#pragma unroll
                for (uint32_t p = 0; p < (1u << kMaxG); ++p) v |= (t[p] <= c ? 1u : 0u) << p;
                emit_row(base + c * L.wc_stride, v);
            }
        }
        if (part != kWcParts) return;
        // 64-bit scalar-predicate rows: ballots over the 64 pods (lane = pod).  HP: one wavefront per row.  GX: there can
        // be hundreds of node-group sets (c5: every 1-3 name combination of 16 names) - a wavefront takes 64 sets at a time,
        // one coalesced load, and hands them round with v_readlane (a scalar load per row costs a memory round trip each);
        // lane i keeps the word of set i and stores its two rows (inactive, active: NHDScheduler.py:240-242, gx_bit).
        for (uint32_t k = wave; k < L.hp_rows; k += NW) {
            const uint64_t word = __ballot(hp_bit(s_hdr[lane], k));
            if (lane == 0) *reinterpret_cast<uint64_t*>(hot + L.hot_hp + 8 * k) = word;
        }
        const bool filtered = (s_hdr[lane].flags & kPodFilter) != 0;       // else: the caller filtered already - every row passes
        const uint64_t my_groups = s_hdr[lane].groups;
        const uint64_t unfiltered = __ballot(!filtered);
        if (tid == 0) *reinterpret_cast<uint64_t*>(hot + L.hot_gx) = 0;     // row 0: never
        for (uint32_t g0 = wave * 64; g0 < L.ngs; g0 += NW * 64) {
            const uint32_t cnt = L.ngs - g0 < 64u ? L.ngs - g0 : 64u;
            const uint64_t my_set = lane < cnt ? a.d.group_sets[g0 + lane] : 0ull;
            uint64_t mine = 0;
            for (uint32_t i = 0; i < cnt; ++i) {
                const uint64_t set = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_set, (int)i) |
                                     (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_set >> 32), (int)i) << 32;
                const uint64_t word = unfiltered | __ballot(filtered && (set & my_groups) != 0);
                if (lane == i) mine = word;
            }
            if (lane < cnt) {
                uint64_t* rows = reinterpret_cast<uint64_t*>(hot + L.hot_gx + 8 * (1 + 2 * (g0 + lane)));
                rows[0] = unfiltered;                                        // node not active
                rows[1] = mine;
            }
        }
        return;
    }

    // part 0: NIC covers per (pod, capacity class), then the cold rows A0/A1[f], R0/R1[sig].  The unions behind both are
    // instantiated per row width (uniform over the block): a two-group tile pays 4 terms per union, not 16.
    const bool staged = a.d.flat_words != 0 && a.d.flat_words <= kDictLdsWords;
    if (staged)
        for (uint32_t w = tid; w < a.d.flat_words / 2; w += THREADS)        // (the stream is padded to an even word count)
            reinterpret_cast<uint32_t*>(s_flat)[w] = reinterpret_cast<const uint32_t*>(a.d.flat)[w];
    auto covers_and_sig_rows = [&](auto width) {
        constexpr uint32_t WW = decltype(width)::value;
        for (uint32_t w = tid; w < kTile * a.d.ncls; w += THREADS) {
            const uint32_t j = w % kTile, c = w / kTile;
            if (s_hdr[j].flags & kPodValid) class_cover_w<WW>(s_req[j].r, a.d.caps[c], s_sum[j].W, s_sum[j].G, s_cover[j][c]);
        }
        __syncthreads();
        for (uint32_t sig = wave; sig < L.nsig; sig += NW) {    // one reach family per (signature, pod), both sockets' rows from it
            uint32_t reach = 0;
            if (staged) {
                // the record is read with the same address in every lane (LDS broadcast); readfirstlane hands the loop
                // bounds to the scalar unit so the walk stays wave-uniform
                auto word = [&](uint32_t i) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)s_flat[i]); };
                uint32_t at = a.d.sig.nsig + 1 + word(sig);
                const uint32_t npools = word(at++);
                reach = 1;
                for (uint32_t pl = 0; pl < npools; ++pl) {
                    const uint32_t head = word(at++), ncc = head & 0xFFu, glimit = head >> 8;
                    uint32_t pool = 1;
                    for (uint32_t k = 0; k < ncc; ++k) {
                        const uint32_t e = word(at++), cnt = e & 0xFFu, cls = e >> 8;
                        pool = dunion_n<WW>(pool, s_cover[lane][cls][cnt > (uint32_t)kMaxG ? kMaxG : cnt]);
                    }
                    if (glimit != NHDFIT_GLIMIT_NONE) pool &= size_le_mask(s_sum[lane].W, glimit);
                    reach = dunion_n<WW>(reach, pool);
                }
                if (!valid) reach = 0;
            } else {
                reach = valid ? sig_reach_w<WW>(a.d.sig, sig, &s_cover[lane][0][0], s_sum[lane].W) : 0u;
            }
            emit_row(img + L.off_r0 + sig * L.row, valid ? entry_r(reach, s_sum[lane].W, 0) : 0u);
            emit_row(img + L.off_r1 + sig * L.row, valid ? entry_r(reach, s_sum[lane].W, 1) : 0u);
        }
    };
    if (W == 2) covers_and_sig_rows(std::integral_constant<uint32_t, 2>{});
    else if (W == 4) covers_and_sig_rows(std::integral_constant<uint32_t, 4>{});
    else if (W == 8) covers_and_sig_rows(std::integral_constant<uint32_t, 8>{});
    else covers_and_sig_rows(std::integral_constant<uint32_t, 16>{});
    for (uint32_t k = wave; k < 2 * L.fg_dim; k += NW) {
        const uint32_t u = k >= L.fg_dim, f = u ? k - L.fg_dim : k;
        emit_row(img + (u ? L.off_a1 : L.off_a0) + f * L.row, valid ? entry_a(s_sum[lane], u, f) : 0u);
    }
    __syncthreads();                                            // the block reads back the cold rows it just wrote
    // hot rows X[class] = A_u[f] & (PCI-mode pods: R_u[sigPCI], NUMA-mode pods: R_u[sigNUMA]) - pure word
    // operations on the cold rows, one lane per (class, assignment)
    const uint64_t m_pci = __ballot((s_hdr[lane].flags & kPodPci) != 0);
    const uint32_t nx = a.nx[0] < L.x_cap ? a.nx[0] : L.x_cap;
    for (uint32_t i = tid; i < nx * W; i += THREADS) {
        const uint32_t k = i / W, p = i % W;
        const uint64_t key = a.xcls[k];
        const uint32_t u = xkey_u(key);
        const uint8_t* rbase = img + (u ? L.off_r1 : L.off_r0) + p * 8;
        const uint64_t av = ld64(img, (u ? L.off_a1 : L.off_a0) + xkey_f(key) * L.row + p * 8);
        const uint64_t rn = ld64(rbase, xkey_sig_numa(key) * L.row), rp = ld64(rbase, xkey_sig_pci(key) * L.row);
        *reinterpret_cast<uint64_t*>(hot + L.hot_x + k * L.x_stride + p * 8) = av & ((rp & m_pci) | (rn & ~m_pci));
    }
}

struct FitItem { uint32_t tile, wcls, c_begin, c_end; };       // one block of the fit role: chunks [c_begin, c_end) of a tile

struct FitArgs {
    const NodeRec* rec[kWClasses];   // node records per row width (k_xrecords), padded to a multiple of 64 nodes
    const nhdfit_plane4* p4;         // busy times (padded likewise)
    uint32_t n;                 // nodes in this shard
    uint32_t chunks;            // ceil(n / 64)
    uint64_t global_base;
    double busy_from;           // busy_threshold(now): a node is busy iff busy_time >= busy_from
    const uint8_t* tabs;        // tile images
    uint32_t pitch;
    uint32_t off_hot[kWClasses], hot_bytes[kWClasses], hot_hp[kWClasses];   // per row width: where the hot section starts, its size, its HP rows
    uint32_t hot_staged[kWClasses];   // == hot_bytes: the whole section is staged in LDS.  Smaller: only this prefix (it ends inside X) and the
                                      // HP rows behind it; X rows past the prefix are read from global memory
    uint32_t hp_bytes;
    uint32_t hp_last;           // last HP row of the staged batch (hp_rows - 1)
    const PodHeader* hdr;       // [tiles*64], zero flags beyond P
    uint32_t P;
    const uint64_t* cand;       // optional [chunks]: candidate nodes (bit = node) common to all pods of the call
    uint64_t* nm;               // optional node-major feasibility words [tiles][chunks*64]: bit j = pod 64*tile+j
    unsigned long long* score;  // [P], pre-zeroed
    const FitItem* items;
    uint32_t dbg_skip;          // tuning aid (NHDFIT_FIT_SKIP): 1 no table sweep, 2 no winner tracking, 4 constant record, 8 no predicate rows
};

// One step of the 64 x 64 bit-matrix transpose across a wavefront: exchange S x S sub-blocks between
// lanes l and l ^ S (S < 32, inside one 32-bit register).
// Value of x in lane (l ^ S), S in {1, 2, 4}: DPP moves inside a row of 16 lanes - no LDS crossbar
// (ds_bpermute), no address registers.
template <int S>
__device__ __forceinline__ uint32_t from_lane_xor(uint32_t x) {
    const int v = (int)x;
    if constexpr (S == 1) return (uint32_t)__builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
    else if constexpr (S == 2) return (uint32_t)__builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    else {
        const int y = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false);     // row_shl:4 -> banks 0,2 read lane+4
        return (uint32_t)__builtin_amdgcn_update_dpp(y, v, 0x114, 0xF, 0xA, false);  // row_shr:4 -> banks 1,3 read lane-4
    }
}

// One butterfly stage (S = 4, 2, 1) of the bit transpose on both words: lanes l and l^S exchange the off-diagonal
// S-bit blocks.  Branch-free: the partner's word rotated by +-S is merged under a per-lane mask (v_alignbit +
// v_bfi).  The rotate amount and the mask are rebuilt from a constant SGPR lane mask in 3 instructions per stage
// (volatile: kept out of the loop pre-header - as loop invariants they would pin 2 VGPRs per stage).
template <int S>
__device__ __forceinline__ void xpose_stage(uint32_t& lo, uint32_t& hi) {
    constexpr uint32_t M = S == 4 ? 0x0F0F0F0Fu : S == 2 ? 0x33333333u : 0x55555555u;   // bits b with (b & S) == 0
    constexpr uint64_t UP = S == 4 ? 0xF0F0F0F0F0F0F0F0ull : S == 2 ? 0xCCCCCCCCCCCCCCCCull : 0xAAAAAAAAAAAAAAAAull;   // lanes l with (l & S) != 0
    uint32_t amt, sgn;
    asm volatile("v_cndmask_b32_e64 %0, %2, %3, %4\n\tv_cndmask_b32_e64 %1, 0, -1, %4"
                 : "=&v"(amt), "=v"(sgn) : "n"(32 - S), "n"(S), "s"(UP));
    const uint32_t keep = M ^ sgn;                  // "up" lanes keep their high blocks, the others their low blocks
    const uint32_t ylo = from_lane_xor<S>(lo), yhi = from_lane_xor<S>(hi);
    const uint32_t rlo = __builtin_amdgcn_alignbit(ylo, ylo, amt), rhi = __builtin_amdgcn_alignbit(yhi, yhi, amt);
    lo = (lo & keep) | (rlo & ~keep);
    hi = (hi & keep) | (rhi & ~keep);
}

// in: lane l holds row l (bit j = column j) as (lo = columns 0..31, hi = columns 32..63);
// out: lane j holds column j (bit l = row l).  ~45 VALU instructions, no LDS traffic.
__device__ __forceinline__ void transpose64(uint32_t& lo, uint32_t& hi) {
    // 32 x 32 blocks: swap the hi word of lanes 0..31 with the lo word of lanes 32..63
    const auto s32 = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
    // 16 x 16 blocks of both words with one v_permlane16_swap: gather the low halves of (lo, hi) in one register and
    // the high halves in another, swap [high halves of lanes l] with [low halves of lanes l + 16]
    const uint32_t l16 = __builtin_amdgcn_perm(s32[1], s32[0], 0x05040100u), h16 = __builtin_amdgcn_perm(s32[1], s32[0], 0x07060302u);
    const auto s16 = __builtin_amdgcn_permlane16_swap(l16, h16, false, false);
    // 8 x 8 blocks: the same with bytes (the scatter of the previous stage folded into this gather); the swap is two
    // DPP moves whose bank masks pick the receiving lanes: lanes 0-7 of a row get the partner's even bytes as their odd
    // bytes, lanes 8-15 the partner's odd bytes as their even bytes
    const uint32_t l8 = __builtin_amdgcn_perm(s16[1], s16[0], 0x06020400u), h8 = __builtin_amdgcn_perm(s16[1], s16[0], 0x07030501u);
    const uint32_t h8x = (uint32_t)__builtin_amdgcn_update_dpp((int)h8, (int)l8, 0x128, 0xF, 0x3, false);   // row_ror:8
    const uint32_t l8x = (uint32_t)__builtin_amdgcn_update_dpp((int)l8, (int)h8, 0x128, 0xF, 0xC, false);
    lo = __builtin_amdgcn_perm(h8x, l8x, 0x05010400u);
    hi = __builtin_amdgcn_perm(h8x, l8x, 0x07030602u);
    xpose_stage<4>(lo, hi);
    xpose_stage<2>(lo, hi);
    xpose_stage<1>(lo, hi);
}

__device__ __forceinline__ uint4 lds16(const uint8_t* img, uint32_t off) {
    return *reinterpret_cast<const uint4*>(__builtin_assume_aligned(img + off, 16));
}
__device__ __forceinline__ uint2 lds8(const uint8_t* img, uint32_t off) {
    return *reinterpret_cast<const uint2*>(__builtin_assume_aligned(img + off, 8));
}

// Pods of the tile (bit j) for which some NUMA assignment passes CPU & GPU & NIC on this lane's node: per PAIR of
// assignments six 16-byte row fetches (ds_read_b128) and 16 three-input bit operations serve all 64 pods.
// a_* = byte addresses of the node's rows in the staged hot section; the m=1 row of a WC record follows its m=0 row.
template <int W>
__device__ __forceinline__ uint64_t sweep_assignments(const uint8_t* hot, uint32_t a_w0, uint32_t a_w1, uint32_t a_x0, uint32_t a_x1) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int q = 0; q < W / 2; ++q) {
        const uint32_t o = q * 16;
        const uint4 w0 = lds16(hot, a_w0 + o), w0m = lds16(hot, a_w0 + W * 8 + o);
        const uint4 w1 = lds16(hot, a_w1 + o), w1m = lds16(hot, a_w1 + W * 8 + o);
        const uint4 x0 = lds16(hot, a_x0 + o), x1 = lds16(hot, a_x1 + o);
        // words .x/.y = assignment 2q (pods 0-31 / 32-63), .z/.w = assignment 2q+1
        const uint32_t c0 = __builtin_amdgcn_bitop3_b32(w0.x, w1m.x, w0m.x & w1.x, 0xEA);     // (a & b) | c
        const uint32_t c1 = __builtin_amdgcn_bitop3_b32(w0.y, w1m.y, w0m.y & w1.y, 0xEA);
        const uint32_t c2 = __builtin_amdgcn_bitop3_b32(w0.z, w1m.z, w0m.z & w1.z, 0xEA);
        const uint32_t c3 = __builtin_amdgcn_bitop3_b32(w0.w, w1m.w, w0m.w & w1.w, 0xEA);
        lo |= __builtin_amdgcn_bitop3_b32(c0, x0.x, x1.x, 0x80) | __builtin_amdgcn_bitop3_b32(c2, x0.z, x1.z, 0x80);   // a & b & c
        hi |= __builtin_amdgcn_bitop3_b32(c1, x0.y, x1.y, 0x80) | __builtin_amdgcn_bitop3_b32(c3, x0.w, x1.w, 0x80);
    }
    return ((uint64_t)hi << 32) | lo;
}

// The same with the lanes whose X rows lie beyond the staged prefix of the hot section reading them from the image in
// global memory (the cluster holds more node classes than LDS has room for: slower, not wrong).
template <int W>
__device__ __forceinline__ uint64_t sweep_assignments_spill(const uint8_t* hot, const uint8_t* hot_global, uint32_t staged,
                                                            uint32_t a_w0, uint32_t a_w1, uint32_t a_x0, uint32_t a_x1) {
    const bool far0 = a_x0 + W * 8 > staged, far1 = a_x1 + W * 8 > staged;
    uint32_t lo = 0, hi = 0;
#pragma unroll 1
    for (int q = 0; q < W / 2; ++q) {
        const uint32_t o = q * 16;
        const uint4 w0 = lds16(hot, a_w0 + o), w0m = lds16(hot, a_w0 + W * 8 + o);
        const uint4 w1 = lds16(hot, a_w1 + o), w1m = lds16(hot, a_w1 + W * 8 + o);
        const uint4 x0 = far0 ? *reinterpret_cast<const uint4*>(hot_global + a_x0 + o) : lds16(hot, a_x0 + o);
        const uint4 x1 = far1 ? *reinterpret_cast<const uint4*>(hot_global + a_x1 + o) : lds16(hot, a_x1 + o);
        const uint32_t c0 = __builtin_amdgcn_bitop3_b32(w0.x, w1m.x, w0m.x & w1.x, 0xEA);
        const uint32_t c1 = __builtin_amdgcn_bitop3_b32(w0.y, w1m.y, w0m.y & w1.y, 0xEA);
        const uint32_t c2 = __builtin_amdgcn_bitop3_b32(w0.z, w1m.z, w0m.z & w1.z, 0xEA);
        const uint32_t c3 = __builtin_amdgcn_bitop3_b32(w0.w, w1m.w, w0m.w & w1.w, 0xEA);
        lo |= __builtin_amdgcn_bitop3_b32(c0, x0.x, x1.x, 0x80) | __builtin_amdgcn_bitop3_b32(c2, x0.z, x1.z, 0x80);
        hi |= __builtin_amdgcn_bitop3_b32(c1, x0.y, x1.y, 0x80) | __builtin_amdgcn_bitop3_b32(c3, x0.w, x1.w, 0x80);
    }
    return ((uint64_t)hi << 32) | lo;
}

// The P x N pass.  Block = chunks [c_begin, c_end) of one pod tile: the hot section of the tile's table image is
// staged in LDS, every wavefront sweeps a contiguous run of 64-node chunks (lane = node: one 16-byte record and
// the busy time per node), writes the node-major verdict word and tracks the tile's first-fit winners.
//
// Winner tracking without transposing every chunk: a wavefront walks its chunks in ascending node order, so a
// pod's first hit is its best node of that run.  The pods still without a hit (and the GPU-less pods still without
// a GPU-less node, SelectNode's preference) are two wave-uniform 64-bit masks; a chunk whose verdict words do
// not touch them - all but the first one or two of a run - costs four instructions.  Only a chunk with news is
// transposed (lane = pod) and scored.
template <int BLOCK, int W, bool SPILL>
__device__ __forceinline__ void role_fit_w(const FitArgs& a, const FitItem it, uint8_t* lds) {
    constexpr int NW = BLOCK / 64;
    constexpr int WC = W == 2 ? 0 : W == 4 ? 1 : W == 8 ? 2 : 3;
    const uint32_t hot_bytes = a.hot_bytes[WC];
    uint8_t* hot = lds;
    const uint32_t staged = a.hot_staged[WC];
    const bool spill = SPILL && staged < hot_bytes;                                 // block-uniform; SPILL: the launch was told to expect it
    unsigned long long (*s_best)[64] = reinterpret_cast<unsigned long long (*)[64]>(lds + lds_slice(spill ? staged + a.hp_bytes : hot_bytes));
    const uint32_t tile = it.tile;
    const uint32_t lane = threadIdx.x & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t pod0 = tile * kTile;
    const NodeRec* __restrict__ recs = a.rec[WC];
    const uint32_t len = it.c_end - it.c_begin, per = (len + NW - 1) / NW;
    const uint32_t c_first = it.c_begin + wave * per;
    const uint32_t c_last = (a.dbg_skip & 64) ? c_first : c_first + per < it.c_end ? c_first + per : it.c_end;

    // Everything the block needs first is requested before anything is waited for: the tile's request headers, the
    // wavefront's first node records and the hot section of the table image are independent L2 round trips - issued one
    // after the other behind a barrier they would add up.
    const PodHeader my_h = a.hdr[pod0 + lane];
    uint4 rv = make_uint4(0u, 0u, 0u, 0u);
    double bt = 0.0;
    if (c_first < c_last) {
        rv = *reinterpret_cast<const uint4*>(recs + c_first * 64 + lane);           // {w0,w1}, {x0,x1}, {gx,hp}, {flags,pad}
        bt = a.p4[c_first * 64 + lane].busy_time;
    }
    const uint8_t* hot_global = a.tabs + (size_t)tile * a.pitch + a.off_hot[WC];
    {   // stage the hot section of the tile's table image in LDS (16 B per lane, fully coalesced)
        const uint4* src = reinterpret_cast<const uint4*>(hot_global);
        uint4* dst = reinterpret_cast<uint4*>(hot);
        if (!(a.dbg_skip & 16)) for (uint32_t i = threadIdx.x; i < staged / 16; i += BLOCK) dst[i] = src[i];
        if (spill) {                                                                 // the HP rows go right behind the prefix
            const uint4* hsrc = reinterpret_cast<const uint4*>(hot_global + a.hot_hp[WC]);
            for (uint32_t i = threadIdx.x; i < a.hp_bytes / 16; i += BLOCK) dst[staged / 16 + i] = hsrc[i];
        }
    }
    __syncthreads();

    // lane-as-pod view of the tile's 64 request headers -> class masks of the tile (scalar registers)
    const bool my_pod_live = pod0 + lane < a.P;
    const bool my_pod_needs_gpu = (my_h.flags & kPodNeedGpu) != 0;
    const uint64_t m_need = __ballot(my_pod_needs_gpu);
    uint64_t need_any = __ballot(my_pod_live);                               // pods without a feasible node so far
    uint64_t need_pref = __ballot(my_pod_live && !my_pod_needs_gpu);         // GPU-less pods without a GPU-less node so far
    uint32_t best_any = ~0u, best_pref = ~0u;                                // lane = pod: local node index

    const uint32_t hp_last = a.hp_last, hot_hp = spill ? staged : a.hot_hp[WC];
    const size_t npad = (size_t)a.chunks * 64;
    // the next chunk's record and busy time are requested before this chunk is worked on: a wavefront's chunks are
    // one dependent chain of L2 round trips otherwise
    for (uint32_t c = c_first; c < c_last; ++c) {
        const uint32_t i = c * 64 + lane;
        uint4 rv_next = rv;
        double bt_next = bt;
        if (c + 1 < c_last && !(a.dbg_skip & 4)) {
            rv_next = *reinterpret_cast<const uint4*>(recs + i + 64);
            bt_next = a.p4[i + 64].busy_time;
        }
        const uint32_t a_w0 = (rv.x & 0xFFFFu) << 3, a_w1 = (rv.x >> 16) << 3;
        const uint32_t a_x0 = (rv.y & 0xFFFFu) << 3, a_x1 = (rv.y >> 16) << 3;
        const uint32_t a_gx = (rv.z & 0xFFFFu) << 3;
        const uint32_t hp = rv.z >> 16;
        const uint32_t a_hp = hot_hp + (hp < hp_last ? hp : hp_last) * 8;
        const bool nogpu = (rv.w & kRecNoGpu) != 0;

        // (1) NUMA-assignment feasibility against all 64 pods (bit-sliced tables), (2) scalar predicates
        uint64_t okm;
        if (SPILL && spill && __ballot(a_x0 + W * 8 > staged || a_x1 + W * 8 > staged))
            okm = sweep_assignments_spill<W>(hot, hot_global, staged, a_w0, a_w1, a_x0, a_x1);
        else
            okm = (a.dbg_skip & 1) ? ((uint64_t)rv.y << 32 | rv.x) : sweep_assignments<W>(hot, a_w0, a_w1, a_x0, a_x1);
        const uint2 gx = (a.dbg_skip & 8) ? make_uint2(rv.z, rv.w) : lds8(hot, a_gx), hpw = (a.dbg_skip & 8) ? make_uint2(~0u, ~0u) : lds8(hot, a_hp);
        const bool busy = bt >= a.busy_from;                                  // Node.IsBusy, nhd/Node.py:847-850
        uint32_t wlo = (uint32_t)okm & gx.x & hpw.x, whi = (uint32_t)(okm >> 32) & gx.y & hpw.y;
        if (busy) { wlo &= ~(uint32_t)m_need; whi &= ~(uint32_t)(m_need >> 32); }      // Matcher.py:107-111
        if (a.cand) {                                                         // candidate dict of the call (FindNode's nl)
            const uint64_t cw = a.cand[c];
            if (!(cw >> lane & 1)) wlo = whi = 0;
        }
        if (a.nm) a.nm[(size_t)tile * npad + i] = ((uint64_t)whi << 32) | wlo;

        // (3) does this chunk change any pod's winner?
        const uint32_t nlo = (uint32_t)need_any | (nogpu ? (uint32_t)need_pref : 0u);
        const uint32_t nhi = (uint32_t)(need_any >> 32) | (nogpu ? (uint32_t)(need_pref >> 32) : 0u);
        if (!(a.dbg_skip & 2) && __ballot(((wlo & nlo) | (whi & nhi)) != 0)) {
            const uint64_t nogpu_mask = __ballot(nogpu);
            transpose64(wlo, whi);                                            // lane j: pod j's verdict over the chunk's 64 nodes
            const uint64_t word = ((uint64_t)whi << 32) | wlo;
            const uint64_t pref = my_pod_needs_gpu ? 0ull : word & nogpu_mask;
            if (word && best_any == ~0u) best_any = c * 64 + (uint32_t)__builtin_ctzll(word);
            if (pref && best_pref == ~0u) best_pref = c * 64 + (uint32_t)__builtin_ctzll(pref);
            need_any &= ~__ballot(word != 0);
            need_pref &= ~__ballot(pref != 0);
        }
        rv = rv_next;
        bt = bt_next;
    }
    unsigned long long best = 0;
    if (best_pref != ~0u) best = score_of(true, a.global_base + best_pref);
    else if (best_any != ~0u) best = score_of(false, a.global_base + best_any);
    if (a.dbg_skip & 32) return;
    s_best[wave][lane] = best;
    __syncthreads();
    if (wave == 0 && my_pod_live) {
        unsigned long long m = s_best[0][lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) m = s_best[w][lane] > m ? s_best[w][lane] : m;
        if (m) atomicMax(&a.score[pod0 + lane], m);
    }
}

template <int BLOCK, bool SPILL = false>
__device__ __forceinline__ void role_fit(const FitArgs& a, uint32_t blk, uint8_t* lds) {
    FitItem it = a.items[blk];                      // block-uniform: keep it in scalar registers
    it.tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.tile);
    it.wcls = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.wcls);
    it.c_begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.c_begin);
    it.c_end = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.c_end);
    switch (it.wcls) {
        case 0: role_fit_w<BLOCK, 2, SPILL>(a, it, lds); break;
        case 1: role_fit_w<BLOCK, 4, SPILL>(a, it, lds); break;
        case 2: role_fit_w<BLOCK, 8, SPILL>(a, it, lds); break;
        default: role_fit_w<BLOCK, 16, SPILL>(a, it, lds); break;
    }
}

struct MapArgs {
    const nhdfit_plane0* p0;
    const nhdfit_plane1* p1;
    const nhdfit_plane2* p2;
    const nhdfit_plane3* p3;
    const nhdfit_detail* det;
    const uint8_t* tabs;             // tile images (their cold R rows: NIC-feasible assignments of a winner)
    uint32_t pitch;
    const uint8_t* tile_wcls;        // row width class of every tile
    ColdView L[kWClasses];
    uint32_t n;
    uint64_t global_base;
    const nhdfit_req* reqs;
    uint32_t P;
    const unsigned long long* score;
    const double* caps;
    nhdfit_mapping* out;
};

// One pod per wavefront: the mapping is a long, branchy, strictly sequential computation (the
// CPython set model), so lanes working on different pods would serialise each other's control flow.
// Lane 0 of each wave does the work (no divergence); 4 096 pods = 4 096 short waves spread over the chip.
// GENERIC = false: pods with G <= 3 (register-resident set model, no scratch traffic);
// GENERIC = true : pods with G == 4 (launched only when the batch contains such pods).
constexpr int kMapWaves = 4;
template <bool GENERIC>
__global__ __launch_bounds__(64 * kMapWaves) void k_map(MapArgs a) {
    const uint32_t p = __builtin_amdgcn_readfirstlane(blockIdx.x * kMapWaves + (threadIdx.x >> 6));
    if (p >= a.P || (threadIdx.x & 63) != 0) return;     // one working lane per wave: scratch traffic of one thread
    if ((a.reqs[p].n_groups > 3) != GENERIC) return;
    // everything indexed dynamically (request, winner detail, result) stays in global memory: no scratch
    nhdfit_mapping& m = a.out[p];
    memset(&m, 0, sizeof(m));
    const unsigned long long s = a.score[p];
    if (s) {
        const uint64_t gi = NHDFIT_SCORE_INDEX(s);
        if (gi >= a.global_base && gi < a.global_base + a.n) {
            const uint32_t i = (uint32_t)(gi - a.global_base);
            WinnerState w;
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
            const nhdfit_req& rq = a.reqs[p];
            const uint32_t bits = nic_assignment_bits(a.tabs + (size_t)(p / kTile) * a.pitch, a.L[a.tile_wcls[p / kTile]], p % kTile,
                                                      rq.map_type == NHDFIT_MAP_PCI, a.p3[i]);
            const uint32_t codes = nic_codes_from_table_bits(bits, (int)rq.n_groups, w.U);
            if (GENERIC) map_winner_t<GenericOps>(rq, w, codes, m);
            else map_winner_t<SmallOps>(rq, w, codes, m);
        }
    }
}

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

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_fit_only(FitArgs a) {
    extern __shared__ __align__(16) uint8_t lds[];
    role_fit<BLOCK>(a, blockIdx.x, lds);
}

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
        uint32_t s = xhash(key);
        for (uint32_t probes = 0; probes < kXSlots; ++probes, s = (s + 1) & (kXSlots - 1)) {
            const unsigned long long prev = atomicCAS(&a.x.key[s], 0ull, key);
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
__global__ __launch_bounds__(256) void k_xrecords(RecArgs a) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= a.count) return;
    const uint32_t i = a.first + t;
    if (i >= a.npad) return;
    if (i >= a.n) {                                                // padding of the last chunk
        for (int w = 0; w < kWClasses; ++w) a.rec[w][i] = dead_record(a.L[w]);
        return;
    }
    const NodeIdx n = node_index(a.p0[i], a.p1[i], a.p2[i], a.p4[i], a.fc_dim, a.fg_dim, a.ngs);
    const nhdfit_plane3 q3 = a.p3[i];
    const uint32_t x0 = a.x.id[xslot_find(a.x, xkey(0, n.f0, q3.sig_numa[0], q3.sig_pci[0]))];
    const uint32_t x1 = a.x.id[xslot_find(a.x, xkey(1, n.f1, q3.sig_numa[1], q3.sig_pci[1]))];
    for (int w = 0; w < kWClasses; ++w) a.rec[w][i] = make_record(n, x0, x1, a.L[w]);
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

template <int BLOCK, bool SPILL = false>      // SPILL: some tiles stage only a prefix of their hot section (refresh_layouts)
__global__ __launch_bounds__(BLOCK, SPILL ? (BLOCK == 512 ? 4 : 5) : (BLOCK == 512 ? 6 : 7)) void k_step(StepArgs a) {   // SPILL: two blocks per CU (LDS)   // 512 threads = 2 waves per SIMD: 3 blocks per CU either way
    extern __shared__ __align__(16) uint8_t lds[];
    uint32_t blk = blockIdx.x;
    const unsigned long long t0 = a.role_clock ? (unsigned long long)wall_clock64() : 0ull;
    // Grid order: the fit role first.  Its blocks are sized to fill two of the three block slots of every CU, so all of
    // them start at once; the side roles (short latency chains on few wavefronts) take the third slot, with issue
    // priority so that they finish - and hand the slot on - sooner.
    if (blk < a.nb_fit) {
        role_fit<BLOCK, SPILL>(a.fit, blk, lds);
        stamp(a.role_clock, 4, t0);
        return;
    }
    blk -= a.nb_fit;
    if (a.side_prio) __builtin_amdgcn_s_setprio(3);
    if (blk < a.nb_choose) {
        if ((threadIdx.x & 63) == 0)
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

// Profiling aid (NHDFIT_ROLE_KERNELS=1): one role per launch, so that rocprofv3 --stats names each role's stand-alone time.
template <int BLOCK, int ROLE>
__global__ __launch_bounds__(BLOCK, 6) void k_role(StepArgs a) {
    extern __shared__ __align__(16) uint8_t lds[];
    const uint32_t blk = blockIdx.x;
    if constexpr (ROLE == 0) {
        if ((threadIdx.x & 63) == 0)
            role_choose(a.choose, (uint32_t)__builtin_amdgcn_readfirstlane((int)(blk * (BLOCK / 64) + (threadIdx.x >> 6))),
                        a.nb_choose * (BLOCK / 64), (a.shapes_P + kTile - 1) / kTile);
    } else if constexpr (ROLE == 1) role_shapes<BLOCK>(a.shapes_m, a.shapes_h, blk, lds);
    else if constexpr (ROLE == 2) role_finish<BLOCK>(a.finish_m, a.finish_h, blk, lds);
    else if constexpr (ROLE == 3) role_digest<BLOCK>(a.digest, blk, lds);
    else role_fit<BLOCK>(a.fit, blk, lds);
}

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

struct UndoRec { uint32_t node, pad[3]; NodeState st; nhdfit_detail d; };

struct SeqArgs {
    nhdfit_plane0* p0; nhdfit_plane1* p1; nhdfit_plane2* p2; nhdfit_plane3* p3; nhdfit_plane4* p4; nhdfit_detail* det;   // the mirror (modified)
    uint32_t n, chunks; uint64_t global_base; double now;
    const nhdfit_req* reqs; const unsigned long long* score; uint32_t P;
    const uint32_t* order;           // caller's pod i -> staged (class-sorted) position
    const uint8_t* tabs; uint32_t pitch; const uint8_t* tile_wcls; Layout L[kWClasses];
    uint64_t* rows;                  // [chunks][P] verdict rows of the snapshot; kept current for the pods without GPUs
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
    bool okg = false, okc = false;
    if (lane < nG) {
        uint32_t t0 = 0, t1 = 0;
        for (int g = 0; g < G; ++g) { if (tup_digit(lane, G, U, g)) t1 += r.gpus[g]; else t0 += r.gpus[g]; }
        okg = t0 <= (uint32_t)w.free_g[0] && t1 <= (uint32_t)w.free_g[1];
    }
    if (lane >= 32 && lane - 32 < nC) {
        const uint32_t code = lane - 32;
        uint32_t t0 = 0, t1 = 0;
        for (int g = 0; g <= G; ++g) {
            const uint32_t d = g < G ? (w.smt ? r.cpu_smt[g] : r.cpu_nosmt[g]) : (w.smt ? r.misc_smt : r.misc_nosmt);
            if (tup_digit(code, G + 1, U, g)) t1 += d; else t0 += d;
        }
        okc = t0 <= (uint32_t)w.free_c[0] && t1 <= (uint32_t)w.free_c[1];
    }
    sg_mask = (uint32_t)__ballot(okg);
    sc_mask = (uint32_t)(__ballot(okc) >> 32);
}
// first_nic_choice: lane = position in the reference's enumeration order (an odometer whose most significant digits are
// the NUMA-0 groups in ascending order, then the NUMA-1 groups; last digit fastest), 64 positions per pass
__device__ __forceinline__ bool first_nic_choice_wave(const nhdfit_req& r, const WinnerState& w, uint32_t gcode, bool pci, uint32_t lane, int8_t nic_idx[kMaxG]) {
    const int G = (int)r.n_groups;
    uint32_t order = 0, numa = 0;
    int n = 0;
    for (int u = 0; u < w.U; ++u)
        for (int g = 0; g < G; ++g)
            if (tup_digit(gcode, G, w.U, g) == u) { order = nib_set(order, n, (uint32_t)g); numa |= (uint32_t)u << g; ++n; }
    uint32_t total = 1;
    for (int g = 0; g < G; ++g) {
        const uint32_t k = w.d->nic_cnt[(numa >> g) & 1];
        if (k == 0) return false;
        total *= k;
    }
    for (uint32_t base = 0; base < total; base += 64) {
        uint32_t rem = base + lane, pick = 0;
        const bool live = rem < total;
        for (int pos = G - 1; pos >= 0; --pos) {
            const int g = (int)nib_get(order, pos);
            const uint32_t k = w.d->nic_cnt[(numa >> g) & 1];
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
                if (((numa >> h) & 1) == u && nib_get(pick, h) == k) { rx = rx - r.rx[h]; tx = tx - r.tx[h]; }
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
            const uint32_t best = (uint32_t)__builtin_amdgcn_readlane((int)pick, __builtin_ctzll(any));
            for (int g = 0; g < G; ++g) nic_idx[g] = (int8_t)nib_get(best, g);
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
__device__ __forceinline__ bool map_on_state_wave(const nhdfit_req& r, const NodeState& s, const nhdfit_detail& d, const double* caps, uint32_t nic_bits,
                                                  const MapTables& t, uint32_t lane, nhdfit_mapping& m) {
    const WinnerState w = state_view(s, d, caps);
    const int G = (int)r.n_groups, U = w.U;
    m = nhdfit_mapping{};
    const uint32_t codes = nic_codes_from_table_bits(nic_bits, G, U);
    if (G > 3) return map_generic_cold(&r, &w, codes, &m);
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
    for (int g = 0; g < kMaxG; ++g) { m.gpu[g] = m.nic_numa[g] = m.nic_idx[g] = -1; }
    for (int g = 0; g <= kMaxG; ++g) m.cpu[g] = -1;
    if (!first_nic_choice_wave(r, w, gcode, r.map_type == NHDFIT_MAP_PCI, lane, m.nic_idx)) return false;
    for (int g = 0; g < G; ++g) { m.gpu[g] = (int8_t)tup_digit(gcode, G, U, g); m.nic_numa[g] = m.gpu[g]; }
    for (int g = 0; g <= G; ++g) m.cpu[g] = (int8_t)tup_digit((uint32_t)ccode, G + 1, U, g);
    m.valid = 1;
    return true;
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
    unsigned long long t_find = 0, t_pick = 0, t_map = 0, n_rounds = 0, tick = a.prof ? wall_clock64() : 0;
    unsigned long long t_sub[5] = {0, 0, 0, 0, 0}, sub = 0, t_c[5] = {0, 0, 0, 0, 0};
    auto sublap = [&](int k) { if (a.prof && wave == 0) { const unsigned long long t = wall_clock64(); t_sub[k] += t - sub; sub = t; } };
    auto lap = [&](unsigned long long& acc) { if (a.prof) { const unsigned long long t = wall_clock64(); acc += t - tick; tick = t; } };
    while (i < a.P) {
        if (s_stop) break;
        // (1) wavefront w: pod i + w
        const uint32_t mine = i + wave;
        int32_t have = -2;
        if (mine < a.P) {
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
                            w = __hip_atomic_load(&a.rows[(size_t)c * a.P + pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
                            atomicAnd(reinterpret_cast<unsigned long long*>(&a.rows[(size_t)(vv[u] >> 6) * a.P + (size_t)tt[u] * 64 + j]), ~(1ull << (vv[u] & 63)));
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
struct CommitArgs {
    nhdfit_plane0* p0; nhdfit_plane1* p1; nhdfit_plane2* p2; nhdfit_plane3* p3; nhdfit_plane4* p4; nhdfit_detail* det;
    uint32_t node; nhdfit_req req; nhdfit_mapping map; double busy_time; SigTable sigs; nhdfit_placement* out;
};
__global__ __launch_bounds__(64) void k_commit(CommitArgs a) {
    if (threadIdx.x != 0) return;
    NodeState s;
    s.p0 = a.p0[a.node]; s.p1 = a.p1[a.node]; s.p2 = a.p2[a.node]; s.p3 = a.p3[a.node]; s.p4 = a.p4[a.node];
    nhdfit_detail d = a.det[a.node];
    nhdfit_placement pl;
    memset(&pl, 0, sizeof pl);
    commit_node(s, d, a.req, a.map, a.busy_time, a.sigs, pl);
    a.p0[a.node] = s.p0; a.p1[a.node] = s.p1; a.p2[a.node] = s.p2; a.p3[a.node] = s.p3; a.p4[a.node] = s.p4;
    a.det[a.node] = d;
    *a.out = pl;
}

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
        GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !AllReduce || !CommDestroy || !GetErrorString || !CommInitAll || !GroupStart || !GroupEnd) {
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

}  // namespace

struct nhdfit_ctx {
    int dev = -1;
    hipStream_t stream = nullptr;        // the step launches, in order
    hipStream_t s_red = nullptr;         // the all-reduce of sharded runs, overlapping the next step launch
    hipEvent_t ev_fit[kBufs] = {}, ev_red[kBufs] = {};   // stream <-> s_red hand-over (sharded runs only)
    // software pipeline: number of steps (since the last stage_requests) whose phase has been launched
    uint64_t n_dig = 0, n_fit = 0, n_shaped = 0, n_chosen = 0, n_finished = 0;
    bool geom_big = true;                // 512-thread step blocks (256 for small problems)
    uint32_t digest_parts = getenv("NHDFIT_DIGEST_PARTS") ? (uint32_t)atoi(getenv("NHDFIT_DIGEST_PARTS")) : 2;   // tuning aid
    uint32_t side_prio = getenv("NHDFIT_SIDE_PRIO") ? (uint32_t)atoi(getenv("NHDFIT_SIDE_PRIO")) : 1;   // tuning aid
    int seq_pods = getenv("NHDFIT_SEQ_PODS") && atoi(getenv("NHDFIT_SEQ_PODS")) == 8 ? 8 : 16;   // tuning aid: pods per round of the sequential kernel
    uint32_t choose_split = getenv("NHDFIT_CHOOSE_SPLIT") ? (uint32_t)atoi(getenv("NHDFIT_CHOOSE_SPLIT")) : 16;   // tuning aid: wavefronts per tile
    bool split = getenv("NHDFIT_SPLIT") != nullptr;
    bool role_kernels = getenv("NHDFIT_ROLE_KERNELS") != nullptr;   // profiling aid: every role as a kernel of its own (512-thread geometry only)
    DevBuf<unsigned long long> role_clock;            // profiling aid: NHDFIT_ROLE_TIMES=<step> prints the role windows of that step
    int64_t role_step = getenv("NHDFIT_ROLE_TIMES") ? atoll(getenv("NHDFIT_ROLE_TIMES")) : -1;   // profiling aid: launch the side roles apart from the fit role
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

    // requests / results
    DevBuf<nhdfit_req> reqs; uint32_t P = 0;
    std::vector<uint32_t> perm;          // device (class-sorted) position -> caller's pod index
    PinBuf<nhdfit_req> pin_reqs; PinBuf<uint8_t> pin_wcls; PinBuf<uint64_t> pin_score; PinBuf<nhdfit_mapping> pin_maps;   // host staging of one call
    PinBuf<uint8_t> pin_items;           // the fit role's work items on their way to the device
    DevBuf<PodHeader> hdr[kBufs]; DevBuf<uint8_t> tabs[kBufs];
    DevBuf<unsigned long long> score[kBufs]; DevBuf<nhdfit_mapping> maps[kBufs];
    DevBuf<uint64_t> nm;                 // node-major verdict words [tiles][chunks*64] (one buffer: the fit roles of consecutive steps run in stream order)
    DevBuf<uint64_t> bitmap;             // pod-major rows [chunks][P], converted from `nm` on demand (fetch, mode B)
    DevBuf<uint64_t> cand;               // [chunks] candidate nodes of the call
    DevBuf<uint8_t> tile_wcls;           // row width class per staged tile
    DevBuf<FitItem> items; uint32_t n_items = 0;   // work items of the fit role (blocks), heaviest tiles first
    std::vector<uint8_t> h_tile_wcls;
    DevBuf<unsigned long long> shape_keys[kBufs]; DevBuf<uint32_t> shape_res[kBufs]; DevBuf<int32_t> shape_slot[kBufs];   // mapping dedup tables
    DevBuf<uint32_t> shape_list[kBufs];  // distinct shapes per tile
    DevBuf<AscEntry> asc;                // layouts of ascending-filled CPython sets (static table, built at creation)
    DevBuf<uint8_t> choose_tab;          // choose_tuples tabulated for U = 2, G <= 2 (static table, built at creation)
    // node records (fit_core.h NodeRec) + the class table behind their X rows; [rec_lo, rec_hi) = nodes whose records
    // are stale (uploads, commits), rec_all = every record (dictionary / capacity / node count changed)
    DevBuf<NodeRec> rec[kWClasses];
    DevBuf<unsigned long long> xkeys; DevBuf<uint32_t> xids; DevBuf<uint64_t> xcls; DevBuf<uint32_t> xnx;
    uint32_t rec_lo = 0, rec_hi = 0; bool rec_all = true;
    uint32_t nx = 0, x_cap = kMinXCap;   // interned classes (as of the last record update) / provisioned X rows
    uint32_t fit_blocks = getenv("NHDFIT_FIT_BLOCKS") ? (uint32_t)atoi(getenv("NHDFIT_FIT_BLOCKS")) : 0;   // tuning aid: blocks of the fit role
    bool use_choose_tab = getenv("NHDFIT_NO_CHOOSE_TABLE") == nullptr;   // tuning aid: run the set model for every shape
    // set-layout state machine for three-group pods (set_states.h): verified against the model on the host
    // (tests/test_pyset_emulation.py), parity-green and 10 % faster per step on the GPU (profiles/r02):
    // on by default, NHDFIT_NO_SET_STATES=1 runs the insertion-by-insertion model instead
    DevBuf<uint64_t> st_info; DevBuf<uint32_t> st_next, st_asc; uint32_t st_n = 0;
    bool use_set_states = getenv("NHDFIT_NO_SET_STATES") == nullptr;
    // mode B
    DevBuf<uint64_t> nogpu, taken, tile_masks; DevBuf<int32_t> touched; DevBuf<uint16_t> gl_tiles; std::vector<uint32_t> order_host; std::vector<SeqResult> seq_host; DevBuf<UndoRec> undo; DevBuf<SeqResult> seq_out; DevBuf<nhdfit_placement> seq_place;
    DevBuf<uint32_t> order, seq_counters;
    DevBuf<uint64_t> sig_keys; DevBuf<uint32_t> sig_ids; uint32_t sig_mask = 0;   // canonical NIC-state key -> signature id (commit_core.h)
    bool use_cand = false, want_bitmap = true, want_map = true;

    // timing
    hipEvent_t ev[kEventRing][2];        // start / end of sampled step launches
    uint8_t ev_kind[kEventRing] = {};    // 0 = step launch with a fit role, 1 = digest-only launch
    int ev_pending = 0;
    nhdfit_stats stats;

    // collective
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
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
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail((c), NHDFIT_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

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
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->s_red));
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
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->s_red, hipStreamNonBlocking);
    for (int b = 0; b < kBufs && e == hipSuccess; ++b) {
        e = hipEventCreateWithFlags(&c->ev_fit[b], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_red[b], hipEventDisableTiming);
    }
    for (auto& q : c->ev)
        for (auto& x : q)
            if (e == hipSuccess) e = hipEventCreate(&x);
    if (e == hipSuccess) e = c->xkeys.reserve(kXSlots);
    if (e == hipSuccess) e = c->xids.reserve(kXSlots);
    if (e == hipSuccess) e = c->xcls.reserve(kXSlots / 2);
    if (e == hipSuccess) e = c->xnx.reserve(2);
    if (e == hipSuccess) e = hipMemsetAsync(c->xkeys.p, 0, kXSlots * sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->xids.p, 0xFF, kXSlots * sizeof(uint32_t), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->xnx.p, 0, 2 * sizeof(uint32_t), c->stream);
    if (e == hipSuccess) e = c->asc.reserve(kAscEntries);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_build_asc, dim3((kAscEntries + 255) / 256), dim3(256), 0, c->stream, c->asc.p);
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
            hipLaunchKernelGGL(k_build_choose, dim3((kChooseEntries + 255) / 256), dim3(256), 0, c->stream, c->asc.p, c->choose_tab.p);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
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
    c->pin_reqs.release(); c->pin_wcls.release(); c->pin_score.release(); c->pin_maps.release(); c->pin_items.release();
    c->caps.release(); c->sig_off.release(); c->pool_off.release(); c->pool_glimit.release(); c->cc.release(); c->sig_flat.release();
    c->reqs.release(); c->bitmap.release(); c->nm.release(); c->cand.release(); c->tile_wcls.release(); c->items.release(); c->xkeys.release(); c->xids.release(); c->xcls.release(); c->xnx.release(); for (auto& r : c->rec) r.release(); c->role_clock.release(); c->asc.release(); c->choose_tab.release(); c->st_info.release(); c->st_next.release(); c->st_asc.release(); c->group_sets.release();
    for (int b = 0; b < kBufs; ++b) { c->shape_keys[b].release(); c->shape_res[b].release(); c->shape_slot[b].release(); c->shape_list[b].release(); }
    c->nogpu.release(); c->taken.release(); c->tile_masks.release(); c->touched.release(); c->gl_tiles.release(); c->seq_counters.release(); c->undo.release(); c->seq_out.release(); c->seq_place.release(); c->order.release(); c->sig_keys.release(); c->sig_ids.release();
    for (int b = 0; b < kBufs; ++b) {
        c->hdr[b].release(); c->tabs[b].release(); c->score[b].release(); c->maps[b].release();
        if (c->ev_fit[b]) (void)hipEventDestroy(c->ev_fit[b]);
        if (c->ev_red[b]) (void)hipEventDestroy(c->ev_red[b]);
    }
    for (auto& q : c->ev)
        for (auto& x : q)
            if (x) (void)hipEventDestroy(x);
    if (c->stream) (void)hipStreamDestroy(c->stream);
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
    c->ncls = ncls;
    c->nsig = nsig;
    c->ngs = n_group_sets ? n_group_sets : 1;
    c->max_cores = max_cores_per_numa;
    c->max_gpus = max_gpus_per_numa;
    c->P = 0;                                       // staged tables (if any) were built for the old dictionary
    HIPCHK(c, hipFuncSetAttribute((const void*)k_fit_only<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_fit_only<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_role<512, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_role<512, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_step<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
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
        c->rec_all = true;
        const size_t padded = ((size_t)capacity + 63) & ~size_t(63);       // the fit role reads whole 64-node chunks
        HIPCHK(c, c->p0.reserve(padded)); HIPCHK(c, c->p1.reserve(padded)); HIPCHK(c, c->p2.reserve(padded));
        HIPCHK(c, c->p3.reserve(padded)); HIPCHK(c, c->p4.reserve(padded)); HIPCHK(c, c->det.reserve(capacity));
        HIPCHK(c, c->origin.reserve(capacity)); c->origin_hi = 0;
        for (auto& r : c->rec) HIPCHK(c, r.reserve(padded));
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
    c->pitch = pitch;
    c->lds_bytes = lds;
    c->x_spill = spill;
    if (c->P) {
        const uint32_t tiles = (c->P + kTile - 1) / kTile;
        for (int b = 0; b < kBufs; ++b) HIPCHK(c, c->tabs[b].reserve((size_t)tiles * pitch));
    }
    return NHDFIT_OK;
}
}  // namespace

int nhdfit_stage_requests(nhdfit_ctx* c, const nhdfit_req* reqs, uint32_t P) {
    if (!c) return NHDFIT_E_INVAL;
    if (!reqs || !P) return fail(c, NHDFIT_E_INVAL, "no requests");
    if (!c->nsig) return fail(c, NHDFIT_E_STATE, "set the dictionary first");
    HIPCHK(c, hipSetDevice(c->dev));
    { int rc_ = sync_all(c); if (rc_) return rc_; }
    { int rc_ = drain_events(c); if (rc_) return rc_; }
    c->n_dig = c->n_fit = c->n_shaped = c->n_chosen = c->n_finished = 0;
    const uint32_t tiles = (P + kTile - 1) / kTile;
    int32_t hp_max = 0;
    for (uint32_t p = 0; p < P; ++p) {
        if (reqs[p].hugepages_gb < 0) return fail(c, NHDFIT_E_INVAL, "pod %u asks for a negative number of hugepages", p);
        hp_max = reqs[p].hugepages_gb > hp_max ? reqs[p].hugepages_gb : hp_max;
    }
    if (hp_max > kMaxHpRows - 2)
        return fail(c, NHDFIT_E_LIMIT, "a pod asks for %d GiB of hugepages (limit %d)", hp_max, kMaxHpRows - 2);
    HIPCHK(c, c->reqs.reserve(P));
    for (int b = 0; b < kBufs; ++b) {
        HIPCHK(c, c->hdr[b].reserve((size_t)tiles * kTile));
        HIPCHK(c, c->score[b].reserve(P));
        HIPCHK(c, c->maps[b].reserve(P));
    }
    // Pods are staged sorted by request class so that 64-pod tiles are homogeneous (narrow table rows, fast sweep of
    // the fit role) and the lanes of the mapping roles have similar group counts; results are un-permuted in fetch.
    c->perm.resize(P);
    std::vector<uint16_t> key(P);
    c->n_big_pods = 0;
    uint32_t start[512 + 1] = {0};
    for (uint32_t p = 0; p < P; ++p) {
        const PodHeader h = pod_header(reqs[p]);
        // group count is the major key, descending: the tiles with the most assignments to sweep are the
        // first blocks of the fit grid (longest-first keeps the tail of the launch short)
        key[p] = (uint16_t)(((h.flags & kPodValid) ? 0u : 1u << 8) | ((reqs[p].n_groups > 3 ? 1u : 0u) << 7) |
                            ((15u - (reqs[p].n_groups & 15u)) << 3) | ((h.flags & (kPodNeedGpu | kPodPci | kPodFilter)) >> 1));
        c->n_big_pods += reqs[p].n_groups > 3;
        start[key[p] + 1]++;
    }
    for (uint32_t k = 0; k < 512; ++k) start[k + 1] += start[k];          // stable counting sort: 9-bit keys
    for (uint32_t p = 0; p < P; ++p) c->perm[start[key[p]]++] = p;
    HIPCHK(c, c->pin_reqs.reserve(P));
    nhdfit_req* sorted = c->pin_reqs.p;                                   // (free again: sync_all above waited for the last copy out of it)
    for (uint32_t i = 0; i < P; ++i) sorted[i] = reqs[c->perm[i]];
    HIPCHK(c, hipMemcpyAsync(c->reqs.p, sorted, (size_t)P * sizeof *reqs, hipMemcpyHostToDevice, c->stream));
    // row width class of every tile: 2^(largest group count among its valid pods) assignments - the digest role
    // derives the same class from the same records
    c->h_tile_wcls.assign(tiles, 0);
    c->max_wcls = 0;
    for (uint32_t i = 0; i < P; ++i)
        if (req_valid(sorted[i])) {
            const uint8_t w = (uint8_t)wclass_of(sorted[i].n_groups);
            if (w > c->h_tile_wcls[i / kTile]) c->h_tile_wcls[i / kTile] = w;
            if (w > c->max_wcls) c->max_wcls = w;
        }
    HIPCHK(c, c->tile_wcls.reserve(tiles));
    HIPCHK(c, c->pin_wcls.reserve(tiles));
    memcpy(c->pin_wcls.p, c->h_tile_wcls.data(), tiles);
    HIPCHK(c, hipMemcpyAsync(c->tile_wcls.p, c->pin_wcls.p, tiles, hipMemcpyHostToDevice, c->stream));
    c->P = P;
    c->hp_rows = (uint32_t)hp_max + 2;
    c->n_items = 0;                                 // the fit role's work items are rebuilt at the next step
    c->use_cand = false;
    return refresh_layouts(c);
}

static int stage_cand(nhdfit_ctx* c, const uint64_t* cand) {
    const size_t chunks = (c->n + 63) / 64;
    HIPCHK(c, c->cand.reserve(chunks ? chunks : 1));
    HIPCHK(c, hipMemcpy(c->cand.p, cand, chunks * sizeof(uint64_t), hipMemcpyHostToDevice));
    c->use_cand = true;
    return NHDFIT_OK;
}

namespace {

// Bring the node records (and the class table behind their X rows) up to date with the mirror.  Cheap no-op when
// nothing changed; otherwise three small kernels over the touched nodes and one 8-byte read-back (the host sizes the
// tile images by the class count).  A grown class count or a changed dictionary re-does every record.
int ensure_records(nhdfit_ctx* c) {
    if (!c->rec_all && c->rec_lo == c->rec_hi) return NHDFIT_OK;
    if (!c->n) { c->rec_all = false; c->rec_lo = c->rec_hi = 0; return NHDFIT_OK; }
    const uint32_t npad = (c->n + 63) & ~63u;
    for (int pass = 0; pass < 2; ++pass) {
        const uint32_t first = c->rec_all ? 0 : c->rec_lo, count = c->rec_all ? npad : c->rec_hi - c->rec_lo;
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
        }
        const dim3 grid((count + 255) / 256), block(256);
        if (pass == 0) {
            hipLaunchKernelGGL(k_xkeys, grid, block, 0, c->stream, r);
            hipLaunchKernelGGL(k_xassign, dim3(1), dim3(1024), 0, c->stream, r.x);
        }
        hipLaunchKernelGGL(k_xrecords, grid, block, 0, c->stream, r);
        HIPCHK(c, hipGetLastError());
        if (pass == 1) break;
        uint32_t nx[2] = {0, 0};
        HIPCHK(c, hipMemcpyAsync(nx, c->xnx.p, sizeof nx, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (nx[1] || nx[0] > kXSlots / 2) return fail(c, NHDFIT_E_LIMIT, "more than %u distinct (free GPUs, NIC signature) node classes", kXSlots / 2);
        c->nx = nx[0];
        if (nx[0] <= c->x_cap) break;
        // more classes than provisioned rows: every hot-section offset moves -> staged tables and all records are redone
        { int rc_ = sync_all(c); if (rc_) return rc_; }
        c->x_cap = x_capacity(nx[0]);
        c->rec_all = true;
        c->n_dig = c->n_fit;
        int rc = refresh_layouts(c);
        if (rc) return rc;
    }
    c->rec_all = false;
    c->rec_lo = c->rec_hi = 0;
    return NHDFIT_OK;
}

// Work items of the fit role (one block each): the tiles' chunk ranges cut so that every block carries about the
// same cost - a chunk of a tile with W assignments costs ~(6 + W) - and the block count is a fixed multiple of what
// the chip holds at once (NHDFIT_FIT_BLOCKS overrides the target).  Wide tiles first: longest-first keeps the tail
// of the launch short.  Small problems simply get one wavefront-run per chunk.
int build_items(nhdfit_ctx* c, uint32_t nw) {
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
    static const bool xcd_items = !(getenv("NHDFIT_XCD_ITEMS") && atoi(getenv("NHDFIT_XCD_ITEMS")) == 0);
    const bool by_xcd = xcd_items && !c->fit_blocks && chunks >= 8u * 4u * nw && nw == 8;
    for (uint32_t t = 0; t < tiles; ++t) {           // staged order = widest tiles first
        const uint32_t w = c->h_tile_wcls[t];
        const uint64_t cost = (uint64_t)chunks * (6u + (2u << w));
        uint32_t nb = (uint32_t)((cost * target + total / 2) / total);
        nb = std::max(1u, std::min(nb, (chunks + nw - 1) / nw));          // at least one chunk per wavefront
        if (by_xcd) {
            static const uint32_t force_k = getenv("NHDFIT_XCD_K") ? (uint32_t)atoi(getenv("NHDFIT_XCD_K")) : 0u;   // tuning aid
            const uint32_t k = force_k ? force_k : nb <= 11 ? 1u : nb <= 23 ? 2u : 4u;         // 8, 16 or 32 blocks
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
    HIPCHK(c, c->pin_items.reserve(items.size() * sizeof(FitItem) + 1));
    memcpy(c->pin_items.p, items.data(), items.size() * sizeof(FitItem));
    HIPCHK(c, hipMemcpyAsync(c->items.p, c->pin_items.p, items.size() * sizeof(FitItem), hipMemcpyHostToDevice, c->stream));
    c->n_items = (uint32_t)items.size();
    return NHDFIT_OK;
}

// One launch of the step kernel with every role that has work (see k_step).  `with_fit`: the fit role for step
// n_fit plus the digest of step n_fit + 1; `flushing`: nothing new will follow, drain the mapping phases.
int launch_step(nhdfit_ctx* c, bool with_fit, bool with_digest, double now, bool flushing) {
    const uint32_t P = c->P, tiles = (P + kTile - 1) / kTile;
    const uint32_t chunks = (c->n + 63) / 64;
    const bool big = c->geom_big;
    const uint32_t block = big ? 512 : 256, nw = block / 64;
    const bool small_map = c->want_map && c->n_big_pods < P;
    if (with_fit || with_digest) { int rc_ = ensure_records(c); if (rc_) return rc_; }

    StepArgs a;
    memset(&a, 0, sizeof a);
    a.shapes_P = P;
    a.side_prio = c->side_prio;
    auto map_args = [&](int b) {
        MapArgs m;
        memset(&m, 0, sizeof m);
        m.p0 = c->p0.p; m.p1 = c->p1.p; m.p2 = c->p2.p; m.p3 = c->p3.p; m.det = c->det.p;
        m.tabs = c->tabs[b].p; m.pitch = c->pitch; m.tile_wcls = c->tile_wcls.p;
        for (int w = 0; w < kWClasses; ++w) m.L[w] = cold_view(c->L[w]);
        m.n = c->n; m.global_base = c->global_base; m.reqs = c->reqs.p; m.P = P;
        m.score = c->score[b].p; m.caps = c->caps.p; m.out = c->maps[b].p;
        return m;
    };
    auto shape_args = [&](int b) {
        return ShapeArgs{c->shape_keys[b].p, c->shape_res[b].p, c->shape_slot[b].p, c->shape_list[b].p, c->asc.p,
                         c->use_choose_tab ? c->choose_tab.p : nullptr,
                         c->use_set_states ? SetStates{c->st_info.p, c->st_next.p, c->st_asc.p, c->st_n} : SetStates{nullptr, nullptr, nullptr, 0}};
    };
    // mapping phases of earlier steps: each advances by at most one step per launch
    bool did_shapes = false, did_choose = false, did_finish = false;
    if (small_map) {
        if (c->n_finished < c->n_chosen) {
            const int b = (int)(c->n_finished % kBufs);
            a.finish_m = map_args(b); a.finish_h = shape_args(b); a.nb_finish = (P + block / 4 - 1) / (block / 4); did_finish = true;
        }
        if (c->n_chosen < c->n_shaped) {
            a.choose = shape_args((int)(c->n_chosen % kBufs));
            a.nb_choose = (tiles * c->choose_split + nw - 1) / nw; did_choose = true;
        }
        // scores of step s are final once its fit launch (and, sharded, its all-reduce) is done; sharded runs give
        // the all-reduce one launch of slack so that it overlaps the next fit instead of stalling the stream
        const uint64_t ready = c->comm && !flushing && c->n_fit ? c->n_fit - 1 : c->n_fit;
        if (c->n_shaped < ready) {
            const int b = (int)(c->n_shaped % kBufs);
            if (c->comm) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_red[b], 0));
            HIPCHK(c, c->shape_keys[b].reserve((size_t)tiles * kTile));
            HIPCHK(c, c->shape_res[b].reserve((size_t)tiles * kTile));
            HIPCHK(c, c->shape_slot[b].reserve(P));
            HIPCHK(c, c->shape_list[b].reserve(tiles));
            a.shapes_m = map_args(b); a.shapes_h = shape_args(b); a.nb_shapes = (P + block / 4 - 1) / (block / 4); did_shapes = true;
        }
    }
    with_digest = with_digest && c->n_dig <= c->n_fit + (with_fit ? 1 : 0) + (c->split ? 1 : 0);   // at most one step ahead of the fit
    if (with_digest) {
        const int b = (int)(c->n_dig % kBufs);                              // the next undigested step
        DigestArgs& d = a.digest;
        d.reqs = c->reqs.p; d.P = P;
        d.d = DictView{c->caps.p, c->ncls, c->group_sets.p, SigDict{c->sig_off.p, c->pool_off.p, c->pool_glimit.p, c->cc.p, c->nsig}, c->sig_flat.p, c->flat_words};
        for (int w = 0; w < kWClasses; ++w) d.L[w] = c->L[w];
        d.pitch = c->pitch; d.tabs = c->tabs[b].p; d.hdr = c->hdr[b].p; d.score = c->score[b].p;
        d.xcls = c->xcls.p; d.nx = c->xnx.p;
        a.nb_digest = tiles * kDigestParts;
    }
    uint32_t nb_fit = 0;
    int bf = -1;
    if (with_fit) {
        bf = (int)(c->n_fit % kBufs);
        if (c->want_bitmap) HIPCHK(c, c->nm.reserve((size_t)tiles * chunks * 64));
        if (!c->n_items) { int rc_ = build_items(c, nw); if (rc_) return rc_; }
        FitArgs& f = a.fit;
        for (int w = 0; w < kWClasses; ++w) {
            f.rec[w] = c->rec[w].p; f.off_hot[w] = c->L[w].off_hot; f.hot_bytes[w] = c->L[w].hot_bytes; f.hot_hp[w] = c->L[w].hot_hp; f.hot_staged[w] = c->hot_staged[w];
        }
        f.hp_last = c->hp_rows - 1; f.hp_bytes = align16(c->hp_rows * 8);
        f.p4 = c->p4.p;
        f.n = c->n; f.chunks = chunks; f.global_base = c->global_base; f.busy_from = busy_threshold(now);
        f.tabs = c->tabs[bf].p; f.pitch = c->pitch; f.hdr = c->hdr[bf].p; f.P = P;
        f.cand = c->use_cand ? c->cand.p : nullptr;
        f.nm = c->want_bitmap ? c->nm.p : nullptr;
        f.score = c->score[bf].p;
        f.items = c->items.p;
        f.dbg_skip = getenv("NHDFIT_FIT_SKIP") ? (uint32_t)atoi(getenv("NHDFIT_FIT_SKIP")) : 0;
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
    if (with_fit && (int64_t)c->n_fit == c->role_step) {
        HIPCHK(c, c->role_clock.reserve(10));
        unsigned long long init[10];
        for (int k = 0; k < 5; ++k) { init[2 * k] = ~0ull; init[2 * k + 1] = 0; }
        HIPCHK(c, hipMemcpyAsync(c->role_clock.p, init, sizeof init, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        a.role_clock = c->role_clock.p;
    }
    const bool timed = (with_fit && (c->n_fit < 2 || (c->n_fit & 7) == 0)) || (!with_fit && with_digest);
    if (timed && c->ev_pending == kEventRing) { int rc = drain_events(c); if (rc) return rc; }
    if (timed) HIPCHK(c, hipEventRecord(c->ev[c->ev_pending][0], c->stream));
    if (c->x_spill) {               // more node classes than LDS rows: the variant whose fit role reads the rest from global memory
        if (big) hipLaunchKernelGGL((k_step<512, true>), dim3(grid), dim3(512), lds, c->stream, a);
        else     hipLaunchKernelGGL((k_step<256, true>), dim3(grid), dim3(256), lds, c->stream, a);
    } else
    if (c->role_kernels) {
        const uint32_t nb[5] = {a.nb_choose, a.nb_shapes, a.nb_finish, a.nb_digest, nb_fit};
        if (nb[0]) hipLaunchKernelGGL((k_role<512, 0>), dim3(nb[0]), dim3(512), 0, c->stream, a);
        if (nb[1]) hipLaunchKernelGGL((k_role<512, 1>), dim3(nb[1]), dim3(512), map_lds_bytes<512>(), c->stream, a);
        if (nb[2]) hipLaunchKernelGGL((k_role<512, 2>), dim3(nb[2]), dim3(512), map_lds_bytes<512>(), c->stream, a);
        if (nb[3]) hipLaunchKernelGGL((k_role<512, 3>), dim3(nb[3]), dim3(512), kDigestLds, c->stream, a);
        if (nb[4]) hipLaunchKernelGGL((k_role<512, 4>), dim3(nb[4]), dim3(512), lds, c->stream, a);
    } else
    if (grid == nb_fit && c->split) {
        if (big) hipLaunchKernelGGL((k_fit_only<512>), dim3(grid), dim3(512), lds, c->stream, a.fit);
        else     hipLaunchKernelGGL((k_fit_only<256>), dim3(grid), dim3(256), lds, c->stream, a.fit);
    } else
    if (big) hipLaunchKernelGGL((k_step<512>), dim3(grid), dim3(512), lds, c->stream, a);
    else     hipLaunchKernelGGL((k_step<256>), dim3(grid), dim3(256), lds, c->stream, a);
    HIPCHK(c, hipGetLastError());
    if (timed) {
        HIPCHK(c, hipEventRecord(c->ev[c->ev_pending][1], c->stream));
        c->ev_kind[c->ev_pending++] = with_fit ? 0 : 1;
    }
    if (a.role_clock) {
        unsigned long long t[10];
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemcpy(t, c->role_clock.p, sizeof t, hipMemcpyDeviceToHost));
        unsigned long long first = ~0ull;
        for (int k = 0; k < 5; ++k) first = t[2 * k] < first ? t[2 * k] : first;
        static const char* names[5] = {"choose", "shapes", "finish", "digest", "fit"};
        for (int k = 0; k < 5; ++k)
            if (t[2 * k + 1]) fprintf(stderr, "[nhdfit] step %lld role %-6s: first block starts +%.2f us, last block ends +%.2f us\n",
                                      (long long)c->role_step, names[k], (t[2 * k] - first) * 0.01, (t[2 * k + 1] - first) * 0.01);
    }
    c->n_finished += did_finish; c->n_chosen += did_choose; c->n_shaped += did_shapes;
    if (with_digest) c->n_dig++;
    if (with_fit) {
        hipStream_t after = c->stream;           // where the scores of this step become final
        if (c->comm) {      // one communicator -> its collectives stay on one stream (s_red), in step order
            HIPCHK(c, hipEventRecord(c->ev_fit[bf], c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->s_red, c->ev_fit[bf], 0));
            ncclResult_t r = g_rccl.AllReduce(c->score[bf].p, c->score[bf].p, P, ncclUint64, ncclMax, c->comm, c->s_red);
            if (r != ncclSuccess) return fail(c, NHDFIT_E_RCCL, "ncclAllReduce: %s", g_rccl.GetErrorString(r));
            after = c->s_red;
        }
        if (c->want_map && c->n_big_pods) {      // pods with 4 proc groups: generic set model (scratch-heavy, kept out of k_step)
            const dim3 mg((P + kMapWaves - 1) / kMapWaves), mb(64 * kMapWaves);
            hipLaunchKernelGGL(k_map<true>, mg, mb, 0, after, map_args(bf));
            HIPCHK(c, hipGetLastError());
        }
        if (c->comm) HIPCHK(c, hipEventRecord(c->ev_red[bf], c->s_red));
        c->n_fit++;
        // no mapping roles for this step (output switched off, or only 4-group pods): nothing to catch up on later
        if (!small_map) c->n_shaped = c->n_chosen = c->n_finished = c->n_fit;
        c->stats.evals_last = (uint64_t)P * c->n;
        // algorithmic bytes of the step, SURVEY.md section 8(d): ceil(P/T) * N * B_node + P * B_req + P * N / 8 + 8 * P
        // with T = 64 pods per tile, B_node = 24 (the 16-byte node record + the 8-byte busy time every tile streams),
        // B_req = 128; the P * N / 8 term is the node-major verdict matrix (dropped when that output is switched off)
        c->stats.bytes_last = (uint64_t)tiles * c->n * 24ull + (uint64_t)P * sizeof(nhdfit_req) +
                              (c->want_bitmap ? (uint64_t)tiles * chunks * 64ull * 8ull : 0ull) + (uint64_t)P * 8ull;
        c->stats.nodes = c->n; c->stats.nsig = c->nsig; c->stats.ncls = c->ncls; c->stats.lds_bytes = c->lds_bytes;
    }
    return NHDFIT_OK;
}

// pod-major rows [chunks][P] of the last step's verdict matrix (stream-ordered after the step that produced it)
int convert_rows(nhdfit_ctx* c) {
    const uint32_t chunks = (c->n + 63) / 64, tiles = (c->P + kTile - 1) / kTile;
    HIPCHK(c, c->bitmap.reserve((size_t)chunks * c->P));
    hipLaunchKernelGGL(k_rows, dim3((tiles * chunks + 3) / 4), dim3(256), 0, c->stream, c->nm.p, c->bitmap.p, chunks, c->P);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return NHDFIT_OK;
}

int flush_pipeline(nhdfit_ctx* c) {
    if (!c->P || !c->want_map || c->n_big_pods >= c->P) return NHDFIT_OK;
    while (c->n_finished < c->n_fit) {
        int rc = launch_step(c, false, false, 0.0, true);
        if (rc) return rc;
    }
    return NHDFIT_OK;
}

}  // namespace

int nhdfit_enqueue_step(nhdfit_ctx* c, double now) {
    if (!c) return NHDFIT_E_INVAL;
    if (!c->P) return fail(c, NHDFIT_E_STATE, "stage requests first");
    if (!c->n) return fail(c, NHDFIT_E_STATE, "no nodes uploaded");
    HIPCHK(c, hipSetDevice(c->dev));
    if (c->n_fit == 0) {
        // 512-thread blocks (8 waves; 3 co-resident blocks per CU at 70 VGPRs: one block's LDS fill overlaps the
        // others' sweep), 256-thread blocks for small problems so that the grid still covers the chip
        const uint32_t tiles = (c->P + kTile - 1) / kTile, chunks = (c->n + 63) / 64;
        c->geom_big = (uint64_t)tiles * ((chunks + 31) / 32) >= (uint32_t)c->prop.multiProcessorCount;
        if (const char* b = getenv("NHDFIT_BLOCK")) c->geom_big = atoi(b) >= 512;      // tuning aid
        if (c->role_kernels) c->geom_big = true;
    }
    if (c->n_dig <= c->n_fit) {                      // first step after staging: its digest has not run yet
        int rc = launch_step(c, false, true, now, false);
        if (rc) return rc;
    }
    if (c->split) {                                  // profiling aid: side roles and fit role as two launches
        int rc = launch_step(c, false, true, now, false);
        if (rc) return rc;
        return launch_step(c, true, false, now, false);
    }
    return launch_step(c, true, true, now, false);
}

int nhdfit_sync(nhdfit_ctx* c) {
    if (!c) return NHDFIT_E_INVAL;
    HIPCHK(c, hipSetDevice(c->dev));
    { int rc_ = sync_all(c); if (rc_) return rc_; }
    return drain_events(c);
}

int nhdfit_fetch(nhdfit_ctx* c, uint64_t* score_out, uint64_t* bitmap_out, nhdfit_mapping* map_out) {
    if (!c) return NHDFIT_E_INVAL;
    if (!c->P || !c->n_fit) return fail(c, NHDFIT_E_STATE, "nothing staged / no step enqueued");
    HIPCHK(c, hipSetDevice(c->dev));
    if (map_out && !c->want_map) return fail(c, NHDFIT_E_STATE, "mapping output is disabled");
    const uint32_t P = c->P;
    const int b = (int)((c->n_fit - 1) % kBufs);               // results of the most recent step
    // the launches that finish the mappings still in flight, the copies behind them on the same stream, ONE wait
    { int rc_ = flush_pipeline(c); if (rc_) return rc_; }
    if (c->comm) HIPCHK(c, hipStreamSynchronize(c->s_red));
    if (score_out) {
        HIPCHK(c, c->pin_score.reserve(P));
        HIPCHK(c, hipMemcpyAsync(c->pin_score.p, c->score[b].p, (size_t)P * 8, hipMemcpyDeviceToHost, c->stream));
    }
    if (map_out) {
        HIPCHK(c, c->pin_maps.reserve(P));
        HIPCHK(c, hipMemcpyAsync(c->pin_maps.p, c->maps[b].p, (size_t)P * sizeof(nhdfit_mapping), hipMemcpyDeviceToHost, c->stream));
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
        { int rc_ = convert_rows(c); if (rc_) return rc_; }
        std::vector<uint64_t> tmp(chunks * P);
        HIPCHK(c, hipMemcpy(tmp.data(), c->bitmap.p, chunks * P * 8, hipMemcpyDeviceToHost));
        for (size_t ch = 0; ch < chunks; ++ch)
            for (uint32_t i = 0; i < P; ++i) bitmap_out[ch * P + c->perm[i]] = tmp[ch * P + i];
    }
    return NHDFIT_OK;
}

int nhdfit_find(nhdfit_ctx* c, const nhdfit_req* reqs, uint32_t P, double now, const uint64_t* cand,
                uint64_t* score_out, uint64_t* bitmap_out, nhdfit_mapping* map_out) {
    static const bool prof = getenv("NHDFIT_FIND_PROF") != nullptr;      // tuning aid: host-side phase times of the call
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!prof) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[nhdfit] find P=%u %s %.1f us\n", P, what, std::chrono::duration<double, std::micro>(t1 - t0).count());
        t0 = t1;
    };
    int rc = nhdfit_stage_requests(c, reqs, P);
    if (rc) return rc;
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

int nhdfit_schedule_batch(nhdfit_ctx* c, const nhdfit_req* reqs, uint32_t P, double now, const uint64_t* cand, int apply,
                          int64_t* node_out, nhdfit_mapping* map_out, nhdfit_placement* place_out, int32_t* status_out,
                          uint32_t* n_done) {
    if (!c) return NHDFIT_E_INVAL;
    if (c->comm) return fail(c, NHDFIT_E_STATE, "sequential (mode B) batches are single-shard: detach the communicator");
    if (!node_out || !n_done) return fail(c, NHDFIT_E_INVAL, "node_out / n_done is NULL");
    if (!std::isfinite(now)) return fail(c, NHDFIT_E_INVAL, "now must be finite (a placed node is busy at `now`)");
    HIPCHK(c, hipSetDevice(c->dev));
    hipStream_t sm = c->stream;
    const uint32_t chunks = (c->n + 63) / 64;
    int rc;
    {
        // snapshot pass: digest + fit (verdict matrix and first-fit scores; the mapping roles are not needed - every
        // placement of the batch is mapped against the node's state at ITS turn)
        const bool wb = c->want_bitmap, wm = c->want_map;
        c->want_bitmap = true; c->want_map = false;
        rc = nhdfit_stage_requests(c, reqs, P);
        if (!rc && cand) rc = stage_cand(c, cand);
        if (!rc) rc = nhdfit_enqueue_step(c, now);
        c->want_bitmap = wb; c->want_map = wm;
        if (rc) return rc;
        if ((rc = convert_rows(c))) return rc;
        const uint32_t tiles = (P + kTile - 1) / kTile;
        HIPCHK(c, c->nogpu.reserve(chunks ? chunks : 1));
        HIPCHK(c, c->taken.reserve(chunks ? chunks : 1));
        HIPCHK(c, c->tile_masks.reserve((size_t)tiles * 2));
        HIPCHK(c, c->touched.reserve(c->n ? c->n : 1));
        HIPCHK(c, c->gl_tiles.reserve(tiles));
        HIPCHK(c, c->seq_counters.reserve(4));
        HIPCHK(c, c->undo.reserve(apply ? 1 : P));
        HIPCHK(c, c->seq_out.reserve(P));
        HIPCHK(c, c->seq_place.reserve(P));
        HIPCHK(c, c->order.reserve(P));
        c->order_host.resize(P);                              // caller's pod -> staged (class-sorted) position
        for (uint32_t i = 0; i < P; ++i) c->order_host[c->perm[i]] = i;
        HIPCHK(c, hipMemcpyAsync(c->order.p, c->order_host.data(), P * sizeof(uint32_t), hipMemcpyHostToDevice, sm));
        HIPCHK(c, hipMemsetAsync(c->taken.p, 0, (size_t)(chunks ? chunks : 1) * sizeof(uint64_t), sm));
        HIPCHK(c, hipMemsetAsync(c->touched.p, 0xFF, (size_t)c->n * sizeof(int32_t), sm));
        HIPCHK(c, hipMemsetAsync(c->seq_counters.p, 0, 4 * sizeof(uint32_t), sm));
        hipLaunchKernelGGL(k_nogpu, dim3(chunks), dim3(64), 0, sm, c->p2.p, c->n, c->nogpu.p);
        const int b0 = (int)((c->n_fit - 1) % kBufs);
        hipLaunchKernelGGL(k_tile_masks, dim3(tiles), dim3(64), 0, sm, c->hdr[b0].p, tiles, c->tile_masks.p);
        HIPCHK(c, hipGetLastError());
    }
    const int b = (int)((c->n_fit - 1) % kBufs);
    SeqArgs sa;
    memset(&sa, 0, sizeof sa);
    sa.p0 = c->p0.p; sa.p1 = c->p1.p; sa.p2 = c->p2.p; sa.p3 = c->p3.p; sa.p4 = c->p4.p; sa.det = c->det.p;
    sa.n = c->n; sa.chunks = chunks; sa.global_base = c->global_base; sa.now = now;
    sa.reqs = c->reqs.p; sa.score = c->score[b].p; sa.P = P; sa.order = c->order.p;
    sa.tabs = c->tabs[b].p; sa.pitch = c->pitch; sa.tile_wcls = c->tile_wcls.p;
    for (int w = 0; w < kWClasses; ++w) sa.L[w] = c->L[w];
    sa.rows = c->bitmap.p; sa.taken = c->taken.p; sa.nogpu = c->nogpu.p; sa.tile_masks = c->tile_masks.p;
    sa.caps = c->caps.p; sa.sigs = sig_table(c); sa.fc_dim = c->max_cores + 1; sa.fg_dim = c->max_gpus + 1; sa.ngs = c->ngs;
    sa.mt = map_tables(c);
    sa.undo = c->undo.p; sa.touched = c->touched.p; sa.counters = c->seq_counters.p; sa.keep_undo = apply ? 0 : 1;
    sa.out = c->seq_out.p; sa.place = c->seq_place.p; sa.n_done = c->seq_counters.p + 1; sa.gl_tiles = c->gl_tiles.p;
    const uint32_t tiles_b = (P + kTile - 1) / kTile;
    size_t seq_lds = lds_slice((size_t)tiles_b * 16) + lds_slice(((size_t)c->sig_mask + 1) * 8) + lds_slice(((size_t)c->sig_mask + 1) * 4) + lds_slice((size_t)P * 4) + lds_slice(tiles_b);
    sa.lds_tables = seq_lds <= 96 * 1024;
    if (!sa.lds_tables) seq_lds = 0;
    const int seq_pods = c->seq_pods;
    HIPCHK(c, hipFuncSetAttribute(seq_pods == 16 ? (const void*)k_seq<16> : (const void*)k_seq<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    const bool seq_prof = getenv("NHDFIT_SEQ_PROF") != nullptr;
    if (seq_prof) { HIPCHK(c, c->role_clock.reserve(16)); sa.prof = c->role_clock.p; }
    if (seq_pods == 16) hipLaunchKernelGGL(k_seq<16>, dim3(1), dim3(1024), seq_lds, sm, sa);
    else hipLaunchKernelGGL(k_seq<8>, dim3(1), dim3(512), seq_lds, sm, sa);
    if (seq_prof) {
        unsigned long long t[16];
        HIPCHK(c, hipStreamSynchronize(sm));
        HIPCHK(c, hipMemcpy(t, c->role_clock.p, sizeof t, hipMemcpyDeviceToHost));
        const double per = 0.01 / (double)(t[3] ? t[3] : 1);
        fprintf(stderr, "[nhdfit] k_seq wave 0 per round: node load %.1f, NIC bits %.1f, mapping %.1f, commit %.1f, write-back %.1f us\n",
                t[5] * per, t[6] * per, t[7] * per, t[8] * per, t[9] * per);
        fprintf(stderr, "[nhdfit] k_seq: %llu pods in %llu rounds; scan %.1f us, pick %.1f us, map+commit %.1f us, columns %.1f us per round\n", t[4], t[3],
                t[0] * per, t[2] * per, t[1] * per, t[10] * per);
        fprintf(stderr, "[nhdfit] k_seq columns: evaluate %.1f, patch %.1f, fence %.1f, barrier %.1f us per round; %.1f (node, tile) pairs per round\n",
                t[11] * per, t[12] * per, t[13] * per, t[14] * per, (double)t[15] / (double)(t[3] ? t[3] : 1));
    }
    if (!apply) hipLaunchKernelGGL(k_undo, dim3(P), dim3(64), 0, sm, sa);
    HIPCHK(c, hipGetLastError());
    c->seq_host.resize(P);
    uint32_t counters[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(c->seq_host.data(), c->seq_out.p, P * sizeof(SeqResult), hipMemcpyDeviceToHost, sm));
    HIPCHK(c, hipMemcpyAsync(counters, c->seq_counters.p, sizeof counters, hipMemcpyDeviceToHost, sm));
    if (place_out) HIPCHK(c, hipMemcpyAsync(place_out, c->seq_place.p, (size_t)P * sizeof(nhdfit_placement), hipMemcpyDeviceToHost, sm));
    rc = nhdfit_sync(c);
    if (rc) return rc;
    *n_done = counters[1];
    int64_t lo = -1, hi = -1;
    for (uint32_t i = 0; i < counters[1]; ++i) {
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
    HIPCHK(c, hipSetDevice(c->dev));
    HIPCHK(c, c->seq_place.reserve(1));
    CommitArgs ca;
    memset(&ca, 0, sizeof ca);
    ca.p0 = c->p0.p; ca.p1 = c->p1.p; ca.p2 = c->p2.p; ca.p3 = c->p3.p; ca.p4 = c->p4.p; ca.det = c->det.p;
    ca.node = node; ca.req = *req; ca.map = *map; ca.busy_time = busy_time; ca.sigs = sig_table(c); ca.out = c->seq_place.p;
    hipLaunchKernelGGL(k_commit, dim3(1), dim3(64), 0, c->stream, ca);       // stream order: after every step in flight
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(place_out, c->seq_place.p, sizeof(nhdfit_placement), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
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
    HIPCHK(c, hipStreamSynchronize(c->stream));                  // a delta kernel in flight may still write its records
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
    { int rc_ = flush_pipeline(c); if (rc_) return rc_; }       // mapping phases of steps in flight read the nodes as they were matched
    HIPCHK(c, c->deltas.reserve(n)); HIPCHK(c, c->delta_run.reserve(run.size())); HIPCHK(c, c->delta_status.reserve(n));
    HIPCHK(c, hipMemcpyAsync(c->deltas.p, sorted.data(), n * sizeof(nhdfit_delta), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->delta_run.p, run.data(), run.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    DeltaArgs da;
    memset(&da, 0, sizeof da);
    da.p0 = c->p0.p; da.p1 = c->p1.p; da.p2 = c->p2.p; da.p3 = c->p3.p; da.p4 = c->p4.p; da.det = c->det.p; da.origin = c->origin.p;
    da.deltas = c->deltas.p; da.run = c->delta_run.p; da.n_runs = n_runs; da.sigs = sig_table(c); da.status = c->delta_status.p;
    hipLaunchKernelGGL(k_delta, dim3((n_runs + 63) / 64), dim3(64), 0, c->stream, da);
    HIPCHK(c, hipGetLastError());
    std::vector<uint8_t> st(n);
    HIPCHK(c, hipMemcpyAsync(st.data(), c->delta_status.p, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));                  // (also keeps `sorted` / `run` alive until the copies are done)
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
    HIPCHK(c, hipStreamSynchronize(c->stream));
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
    const char* mode = getenv("NHDFIT_GROUP_REDUCE");
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
            if (hipSetDevice(c->dev) != hipSuccess) { r = ncclSystemError; break; }
            const int b = c->n_fit ? (int)((c->n_fit - 1) % kBufs) : 0;
            if (!c->n) {                                                 // a shard without nodes contributes "no feasible node"
                if (c->score[b].reserve(P) != hipSuccess || hipMemsetAsync(c->score[b].p, 0, (size_t)P * 8, c->stream) != hipSuccess) { r = ncclSystemError; break; }
            }
            r = g_rccl.AllReduce(c->score[b].p, c->score[b].p, P, ncclUint64, ncclMax, g->comm[k], c->stream);
        }
        ncclResult_t r2 = g_rccl.GroupEnd();
        if (r != ncclSuccess || r2 != ncclSuccess) { g->err = std::string("ncclAllReduce (group): ") + g_rccl.GetErrorString(r != ncclSuccess ? r : r2); return NHDFIT_E_RCCL; }
    } else if (n > 1) {                                                  // host max-reduce (debug / test path): D2H, max, H2D
        host_scores.resize(n);
        std::vector<uint64_t> best(P, 0), tmp(P);
        for (size_t k = 0; k < n; ++k) {
            nhdfit_ctx* c = g->ctx[k];
            if (!c->n) continue;
            HIPCHK(c, hipSetDevice(c->dev));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            const int b = (int)((c->n_fit - 1) % kBufs);
            HIPCHK(c, hipMemcpy(tmp.data(), c->score[b].p, (size_t)P * 8, hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < P; ++i) best[i] = std::max(best[i], tmp[i]);
        }
        for (size_t k = 0; k < n; ++k) {
            nhdfit_ctx* c = g->ctx[k];
            if (!c->n) continue;
            HIPCHK(c, hipSetDevice(c->dev));
            const int b = (int)((c->n_fit - 1) % kBufs);
            HIPCHK(c, hipMemcpy(c->score[b].p, best.data(), (size_t)P * 8, hipMemcpyHostToDevice));
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
    *out = c->stats;
    return NHDFIT_OK;
}

int nhdfit_reset_stats(nhdfit_ctx* c) {
    if (!c) return NHDFIT_E_INVAL;
    int rc = nhdfit_sync(c);
    if (rc) return rc;
    c->stats.launches = 0;
    c->stats.fit_ms_total = 0;
    return NHDFIT_OK;
}

}  // extern "C"
