// host_harness.cpp - TEST INFRASTRUCTURE.  Compiles the kernels' shared arithmetic
// (nhd_amd/csrc/fit_core.h, winner_map.h) for the host so that `pytest -m "not gpu"` can check the
// table construction, the per-pair predicate, the selection word and the CPython set model against
// the oracle without a GPU.  It is NOT part of libnhdfit.so and nothing in nhd_amd/ loads it.
#include <algorithm>
#include <cstring>
#include <vector>
#include "../../nhd_amd/csrc/seq_core.h"
#include "../../nhd_amd/csrc/set_states.h"
#include "../../nhd_amd/csrc/dict_stream.h"

using namespace nhdfit;

#include <map>

namespace {
struct Dict {
    uint32_t fcmax, fgmax;
    const uint64_t* gs; uint32_t ngs;
    const double* caps; uint32_t ncls;
    SigDict sig;
};

// table image of one tile (up to 64 pods): the bytes the digest role produces, built pod by pod
void build_tile(const nhdfit_req* reqs, uint32_t npods, const Dict& d, const Layout& L, const std::vector<uint64_t>& xcls,
                uint8_t* img, PodHeader* hdr, const std::vector<uint16_t>* typed = nullptr, int* typed_mismatches = nullptr) {
    std::memset(img, 0, L.bytes);
    std::vector<uint16_t> cover(d.ncls * (kMaxG + 1));
    uint8_t* hot = img + L.off_hot;
    auto set_bits = [&](uint8_t* row, uint32_t v, uint32_t j) {
        for (uint32_t p = 0; p < L.W; ++p)
            if (v >> p & 1) *reinterpret_cast<uint64_t*>(row + p * 8) |= 1ull << j;
    };
    uint64_t m_pci = 0;
    for (uint32_t j = 0; j < (uint32_t)kTile; ++j) {
        nhdfit_req r;
        if (j < npods) r = reqs[j]; else std::memset(&r, 0, sizeof r);
        hdr[j] = pod_header(r);
        if (hdr[j].flags & kPodPci) m_pci |= 1ull << j;
        for (uint32_t k = 0; k < L.hp_rows; ++k)
            if (hp_bit(hdr[j], k)) *reinterpret_cast<uint64_t*>(hot + L.hot_hp + 8 * k) |= 1ull << j;
        for (uint32_t g = 0; g < 1 + 2 * L.ngs; ++g)
            if (gx_bit(hdr[j], g, d.gs)) *reinterpret_cast<uint64_t*>(hot + L.hot_gx + 8 * g) |= 1ull << j;
        if (!(hdr[j].flags & kPodValid)) continue;
        PodSums s;
        pod_sums(r, s);
        for (uint32_t c = 0; c < d.ncls; ++c) class_cover(r, d.caps[c], s.W, s.G, &cover[c * (kMaxG + 1)]);
        for (uint32_t u = 0; u < 2; ++u)
            for (uint32_t smt = 0; smt < 2; ++smt)
                for (uint32_t c = 0; c < L.fc_dim; ++c)
                    for (uint32_t m = 0; m < 2; ++m)
                        set_bits(hot + (u ? L.hot_wc1 : L.hot_wc0) + (smt * L.fc_dim + c) * L.wc_stride + m * L.row, entry_w(s, u, smt, c, m), j);
        for (uint32_t u = 0; u < 2; ++u)
            for (uint32_t f = 0; f < L.fg_dim; ++f) set_bits(img + (u ? L.off_a1 : L.off_a0) + f * L.row, entry_a(s, u, f), j);
        for (uint32_t sig = 0; sig < L.nsig; ++sig) {
            const uint32_t reach = sig_reach(d.sig, sig, cover.data(), s.W);
            if (typed && !typed->empty() && typed_reach(typed->data(), L.nsig, sig, cover.data(), s.W) != reach) ++*typed_mismatches;   // the digest's form by pool type
            set_bits(img + L.off_r0 + sig * L.row, entry_r(reach, s.W, 0), j);
            set_bits(img + L.off_r1 + sig * L.row, entry_r(reach, s.W, 1), j);
        }
    }
    for (uint32_t k = 0; k < xcls.size(); ++k)
        for (uint32_t p = 0; p < L.W; ++p) {
            const uint64_t key = xcls[k];
            const uint32_t u = xkey_u(key);
            const uint64_t av = ld64(img, (u ? L.off_a1 : L.off_a0) + xkey_f(key) * L.row + p * 8);
            const uint64_t rn = ld64(img, (u ? L.off_r1 : L.off_r0) + xkey_sig_numa(key) * L.row + p * 8);
            const uint64_t rp = ld64(img, (u ? L.off_r1 : L.off_r0) + xkey_sig_pci(key) * L.row + p * 8);
            *reinterpret_cast<uint64_t*>(hot + L.hot_x + k * L.x_stride + p * 8) = av & ((rp & m_pci) | (rn & ~m_pci));
        }
}
}  // namespace

extern "C" {

// words of the dictionary's stream by pool type (dict_stream.h), pool types in it; 0 words: the dictionary does not fit the format
int hh_typed_stream(const uint32_t* sig_off, uint32_t nsig, const uint32_t* pool_off, const uint8_t* pool_glimit, const nhdfit_cc* cc,
                    uint32_t* ntypes) {
    const std::vector<uint16_t> f2 = build_typed_stream(SigDict{sig_off, pool_off, pool_glimit, cc, nsig});
    if (ntypes) *ntypes = f2.empty() ? 0u : f2[0];
    return (int)f2.size();
}

// CPU twin of nhdfit_find (mode A), same outputs, chunk-major bitmap [ceil(n/64)][P].  cand: [chunks] node mask.
// Returns the number of (node, tile) pairs whose hot-section verdict differs from the cold-section one (must be 0).
int hh_find(const nhdfit_plane0* p0, const nhdfit_plane1* p1, const nhdfit_plane2* p2, const nhdfit_plane3* p3,
            const nhdfit_plane4* p4, const nhdfit_detail* det, uint32_t n, uint64_t global_base,
            const nhdfit_req* reqs, uint32_t P, double now, uint32_t fcmax, uint32_t fgmax,
            const uint64_t* gs, uint32_t ngs, const double* caps, uint32_t ncls,
            const uint32_t* sig_off, uint32_t nsig, const uint32_t* pool_off, const uint8_t* pool_glimit,
            const nhdfit_cc* cc, const uint64_t* cand, uint64_t* score, uint64_t* bitmap, nhdfit_mapping* maps,
            int force_generic) {
    const uint32_t chunks = (n + 63) / 64;
    int32_t hpmax = 0;
    for (uint32_t p = 0; p < P; ++p) hpmax = reqs[p].hugepages_gb > hpmax ? reqs[p].hugepages_gb : hpmax;
    const Dict d{fcmax, fgmax, gs, ngs, caps, ncls, SigDict{sig_off, pool_off, pool_glimit, cc, nsig}};
    // node classes, interned in node order (the device interns them in a hash table: the ids differ, the rows do not)
    std::map<uint64_t, uint32_t> xid;
    std::vector<uint64_t> xcls;
    std::vector<NodeIdx> nidx(n);
    std::vector<uint32_t> x0(n), x1(n);
    for (uint32_t i = 0; i < n; ++i) {
        nidx[i] = node_index(p0[i], p1[i], p2[i], p4[i], fcmax + 1, fgmax + 1, ngs);
        for (uint32_t u = 0; u < 2; ++u) {
            const uint64_t key = xkey(u, u ? nidx[i].f1 : nidx[i].f0, p3[i].sig_numa[u], p3[i].sig_pci[u]);
            auto it = xid.find(key);
            if (it == xid.end()) { it = xid.emplace(key, (uint32_t)xcls.size()).first; xcls.push_back(key); }
            (u ? x1 : x0)[i] = it->second;
        }
    }
    const uint32_t x_cap = x_capacity((uint32_t)xcls.size());
    const double busy_from = busy_threshold(now);
    int mismatches = 0;
    const std::vector<uint16_t> typed_stream = build_typed_stream(d.sig);      // (empty: the dictionary does not fit the format)
    std::vector<uint8_t> img;
    std::vector<PodHeader> hdr(kTile);
    std::vector<uint8_t> tile_wcls((P + kTile - 1) / kTile, 0);
    for (uint32_t p = 0; p < P; ++p) {
        score[p] = 0;
        if (req_valid(reqs[p])) tile_wcls[p / kTile] = std::max<uint8_t>(tile_wcls[p / kTile], (uint8_t)wclass_of(reqs[p].n_groups));
    }
    for (uint32_t t0 = 0; t0 < P; t0 += kTile) {
        const uint32_t np = P - t0 < (uint32_t)kTile ? P - t0 : kTile;
        const Layout L = make_layout(2u << tile_wcls[t0 / kTile], fcmax, fgmax, nsig, ngs, (uint32_t)hpmax + 2, x_cap);
        img.assign(L.bytes, 0);
        build_tile(reqs + t0, np, d, L, xcls, img.data(), hdr.data(), &typed_stream, &mismatches);
        uint64_t m_need = 0, m_pci = 0;
        for (uint32_t j = 0; j < np; ++j) {
            if (hdr[j].flags & kPodNeedGpu) m_need |= 1ull << j;
            if (hdr[j].flags & kPodPci) m_pci |= 1ull << j;
        }
        // the pair form of the sweep (fit_core.h "pair rows"): free-core counts clamped to the tile's largest demand
        uint32_t max_demand = 0;
        for (uint32_t j = 0; j < np; ++j)
            if (req_valid(reqs[t0 + j])) max_demand = std::max(max_demand, req_max_demand(reqs[t0 + j]));
        const uint32_t pair_D = pair_dim(max_demand, fcmax + 1);
        for (uint32_t c = 0; c < chunks; ++c) {
            uint64_t nogpu = 0;
            uint64_t fm[64];
            const uint32_t cnt = n - c * 64 < 64 ? n - c * 64 : 64;
            for (uint32_t l = 0; l < cnt; ++l) {
                const uint32_t i = c * 64 + l;
                const bool busy = p4[i].busy_time >= busy_from;
                if (busy != ((now - p4[i].busy_time) < kMinBusySecs)) ++mismatches;      // the threshold form of IsBusy
                const NodeRec rec = make_record(nidx[i], x0[i], x1[i], L, l, pair_D);
                // the C row the record carries ready-made (NodeRec::flags, round 6) is the row the free-core counts name, and the flag beside it is intact
                if ((uint32_t)(rec.flags >> 1) != pair_c_row(rec.cc, pair_D) || (bool)(rec.flags & kRecNoGpu) != (bool)nidx[i].nogpu) ++mismatches;
                fm[l] = node_word_hot(img.data() + L.off_hot, L, rec, busy, m_need);
                if (fm[l] != node_word_cold(img.data(), L, nidx[i], p3[i], busy, m_need, m_pci)) ++mismatches;
                if (fm[l] != node_word_pair(img.data() + L.off_hot, L, rec, pair_D, busy, m_need)) ++mismatches;
                if (rec_pos(rec) != l) ++mismatches;
                if (cand && !(cand[c] >> l & 1)) fm[l] = 0;
                if (nidx[i].nogpu) nogpu |= 1ull << l;
            }
            for (uint32_t j = 0; j < np; ++j) {
                uint64_t w = 0;
                for (uint32_t l = 0; l < cnt; ++l)
                    if (fm[l] >> j & 1) w |= 1ull << l;
                if (bitmap) bitmap[(size_t)c * P + t0 + j] = w;
                const uint64_t s = chunk_score(w, nogpu, hdr[j].flags & kPodNeedGpu, global_base + (uint64_t)c * 64);
                if (s > score[t0 + j]) score[t0 + j] = s;
            }
        }
    }
    if (!maps) return mismatches;
    Layout L{};
    for (uint32_t p = 0; p < P; ++p) {
        std::memset(&maps[p], 0, sizeof(nhdfit_mapping));
        if (p % kTile == 0) {
            const uint32_t np = P - p < (uint32_t)kTile ? P - p : kTile;
            L = make_layout(2u << tile_wcls[p / kTile], fcmax, fgmax, nsig, ngs, (uint32_t)hpmax + 2, x_cap);
            img.assign(L.bytes, 0);
            build_tile(reqs + p, np, d, L, xcls, img.data(), hdr.data());
        }
        if (!score[p]) continue;
        const uint64_t gi = NHDFIT_SCORE_INDEX(score[p]);
        if (gi < global_base || gi >= global_base + n) continue;
        const uint32_t i = (uint32_t)(gi - global_base);
        WinnerState w;
        w.U = det[i].numa_nodes;
        w.smt = p2[i].flags & NHDFIT_NF_SMT;
        w.free_c[0] = popc64(p0[i].t0[0] & p1[i].t1[0]);
        w.free_c[1] = popc64(p0[i].t0[1] & p1[i].t1[1]);
        w.free_g[0] = popc32(p2[i].gpu_free & ~p2[i].gpu_numa1);
        w.free_g[1] = popc32(p2[i].gpu_free & p2[i].gpu_numa1);
        w.d = det + i;
        w.caps = caps;
        const uint32_t bits = nic_assignment_bits(img.data(), L, p % kTile, reqs[p].map_type == NHDFIT_MAP_PCI, p3[i]);
        const uint32_t codes = nic_codes_from_table_bits(bits, (int)reqs[p].n_groups, w.U);
        if (force_generic) map_winner_t<GenericOps>(reqs[p], w, codes, maps[p]);
        else map_winner(reqs[p], w, codes, maps[p]);
    }
    return mismatches;
}

// The lone-pod form of the find (fit_core.h lone_pod_fits: the pod's own 16-bit masks looked up per node instead of the
// bit-sliced tile image), pod by pod: scores and chunk-major bitmap as hh_find returns them.  The signature reach families
// come from the dictionary's 16-bit stream (sig_reach_flat), built here as nhdfit_set_dictionary builds it.
int hh_find_lone(const nhdfit_plane0* p0, const nhdfit_plane1* p1, const nhdfit_plane2* p2, const nhdfit_plane3* p3,
                 const nhdfit_plane4* p4, uint32_t n, uint64_t global_base,
                 const nhdfit_req* reqs, uint32_t P, double now, uint32_t fcmax, uint32_t fgmax,
                 const uint64_t* gs, uint32_t ngs, const double* caps, uint32_t ncls,
                 const uint32_t* sig_off, uint32_t nsig, const uint32_t* pool_off, const uint8_t* pool_glimit,
                 const nhdfit_cc* cc, const uint64_t* cand, uint64_t* score, uint64_t* bitmap, uint32_t* nic_bits_of_winner) {
    const uint32_t chunks = (n + 63) / 64, fc_dim = fcmax + 1, fg_dim = fgmax + 1;
    std::vector<uint16_t> flat(nsig + 1, 0);
    for (uint32_t sg = 0; sg < nsig; ++sg) {
        const size_t at = flat.size() - (nsig + 1);
        if (at > 0xFFFFu) return -1;
        flat[sg] = (uint16_t)at;
        flat.push_back((uint16_t)(sig_off[sg + 1] - sig_off[sg]));
        for (uint32_t pl = sig_off[sg]; pl < sig_off[sg + 1]; ++pl) {
            const uint32_t ncc_pl = pool_off[pl + 1] - pool_off[pl];
            if (ncc_pl > 255u) return -1;
            flat.push_back((uint16_t)(pool_glimit[pl] << 8 | ncc_pl));
            for (uint32_t k = pool_off[pl]; k < pool_off[pl + 1]; ++k) flat.push_back((uint16_t)((cc[k].cls & 0xFFu) << 8 | cc[k].cnt));
        }
    }
    const double busy_from = busy_threshold(now);
    int mismatches = 0;
    std::vector<uint16_t> cover(ncls * (kMaxG + 1)), a0(fg_dim), a1(fg_dim), w0(2 * fc_dim * 2), w1(2 * fc_dim * 2), r0(nsig), r1(nsig);
    for (uint32_t p = 0; p < P; ++p) {
        score[p] = 0;
        if (nic_bits_of_winner) nic_bits_of_winner[p] = 0;
        const nhdfit_req& r = reqs[p];
        const PodHeader h = pod_header(r);
        std::fill(a0.begin(), a0.end(), 0); std::fill(a1.begin(), a1.end(), 0);
        std::fill(w0.begin(), w0.end(), 0); std::fill(w1.begin(), w1.end(), 0);
        std::fill(r0.begin(), r0.end(), 0); std::fill(r1.begin(), r1.end(), 0);
        if (h.flags & kPodValid) {
            PodSums s;
            pod_sums(r, s);
            for (uint32_t c = 0; c < ncls; ++c) class_cover(r, caps[c], s.W, s.G, &cover[c * (kMaxG + 1)]);
            for (uint32_t f = 0; f < fg_dim; ++f) { a0[f] = (uint16_t)entry_a(s, 0, f); a1[f] = (uint16_t)entry_a(s, 1, f); }
            for (uint32_t smt = 0; smt < 2; ++smt)
                for (uint32_t c = 0; c < fc_dim; ++c)
                    for (uint32_t m = 0; m < 2; ++m) {
                        w0[(smt * fc_dim + c) * 2 + m] = (uint16_t)entry_w(s, 0, smt, c, m);
                        w1[(smt * fc_dim + c) * 2 + m] = (uint16_t)entry_w(s, 1, smt, c, m);
                    }
            for (uint32_t sig = 0; sig < nsig; ++sig) {
                const uint32_t reach = sig_reach_flat(flat.data(), nsig, sig, cover.data(), s.W);
                r0[sig] = (uint16_t)entry_r(reach, s.W, 0);
                r1[sig] = (uint16_t)entry_r(reach, s.W, 1);
            }
        }
        const LoneMasks t{a0.data(), a1.data(), w0.data(), w1.data(), r0.data(), r1.data()};
        for (uint32_t c = 0; c < chunks; ++c) {
            uint64_t w = 0, nogpu = 0;
            const uint32_t cnt = n - c * 64 < 64 ? n - c * 64 : 64;
            for (uint32_t l = 0; l < cnt; ++l) {
                const uint32_t i = c * 64 + l;
                const NodeIdx ni = node_index(p0[i], p1[i], p2[i], p4[i], fc_dim, fg_dim, ngs);
                if (ni.nogpu) nogpu |= 1ull << l;
                if (cand && !(cand[c] >> l & 1)) continue;
                if (lone_pod_fits(t, h, ni, p3[i], p4[i].busy_time >= busy_from, gs)) w |= 1ull << l;
            }
            if (bitmap) bitmap[(size_t)c * P + p] = w;
            const uint64_t sc = chunk_score(w, nogpu, h.flags & kPodNeedGpu, global_base + (uint64_t)c * 64);
            if (sc > score[p]) score[p] = sc;
        }
        if (nic_bits_of_winner && score[p]) {
            const uint64_t i = NHDFIT_SCORE_INDEX(score[p]) - global_base;
            nic_bits_of_winner[p] = lone_nic_bits(t, (h.flags & kPodPci) != 0, p3[i]);
            // the table form's bits for the same winner: a one-pod tile image, column 0
            const Dict d{fcmax, fgmax, gs, ngs, caps, ncls, SigDict{sig_off, pool_off, pool_glimit, cc, nsig}};
            const Layout L = make_layout(2u << wclass_of(r.n_groups), fcmax, fgmax, nsig, ngs, (uint32_t)(r.hugepages_gb > 0 ? r.hugepages_gb : 0) + 2, kMinXCap);
            std::vector<uint8_t> img(L.bytes);
            std::vector<PodHeader> hdr(kTile);
            build_tile(&r, 1, d, L, std::vector<uint64_t>(), img.data(), hdr.data());
            if (nic_assignment_bits(img.data(), L, 0, (h.flags & kPodPci) != 0, p3[i]) != nic_bits_of_winner[p]) ++mismatches;
        }
    }
    return mismatches;
}

// CPU twin of nhdfit_schedule_batch (mode B with the commit step on the packed state): the algorithm of seq_core.h /
// k_seq, pod by pod, on host copies of the planes (modified in place = apply).  Inputs as hh_find plus the snapshot
// rows / scores hh_find produced.  Returns the number of pods decided (< P: a commit produced an unknown NIC state).
uint32_t hh_schedule(nhdfit_plane0* p0, nhdfit_plane1* p1, nhdfit_plane2* p2, nhdfit_plane3* p3, nhdfit_plane4* p4, nhdfit_detail* det,
                     uint32_t n, uint64_t global_base, const nhdfit_req* reqs, uint32_t P, double now, uint32_t fcmax, uint32_t fgmax,
                     const uint64_t* gs, uint32_t ngs, const double* caps, uint32_t ncls,
                     const uint32_t* sig_off, uint32_t nsig, const uint32_t* pool_off, const uint8_t* pool_glimit, const nhdfit_cc* cc,
                     const uint64_t* score_a, uint64_t* rows, int64_t* node_out, nhdfit_mapping* map_out, nhdfit_placement* place_out,
                     int32_t* status_out) {
    const uint32_t chunks = (n + 63) / 64, tiles = (P + kTile - 1) / kTile;
    int32_t hpmax = 0;
    for (uint32_t p = 0; p < P; ++p) hpmax = reqs[p].hugepages_gb > hpmax ? reqs[p].hugepages_gb : hpmax;
    const Dict d{fcmax, fgmax, gs, ngs, caps, ncls, SigDict{sig_off, pool_off, pool_glimit, cc, nsig}};
    // signature key table, as nhdfit_set_dictionary builds it
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
    // tile images (cold rows are all the column refresh needs; X rows are left empty)
    std::vector<uint8_t> tile_wcls(tiles, 0);
    for (uint32_t p = 0; p < P; ++p)
        if (req_valid(reqs[p])) tile_wcls[p / kTile] = std::max<uint8_t>(tile_wcls[p / kTile], (uint8_t)wclass_of(reqs[p].n_groups));
    std::vector<Layout> L(tiles);
    std::vector<std::vector<uint8_t>> img(tiles);
    std::vector<uint64_t> m_need(tiles, 0), m_pci(tiles, 0);
    std::vector<PodHeader> hdr(kTile);
    const std::vector<uint64_t> no_classes;
    for (uint32_t t = 0; t < tiles; ++t) {
        const uint32_t np = P - t * kTile < (uint32_t)kTile ? P - t * kTile : kTile;
        L[t] = make_layout(2u << tile_wcls[t], fcmax, fgmax, nsig, ngs, (uint32_t)hpmax + 2, kMinXCap);
        img[t].assign(L[t].bytes, 0);
        build_tile(reqs + t * kTile, np, d, L[t], no_classes, img[t].data(), hdr.data());
        for (uint32_t j = 0; j < np; ++j) {
            if (hdr[j].flags & kPodNeedGpu) m_need[t] |= 1ull << j;
            if (hdr[j].flags & kPodPci) m_pci[t] |= 1ull << j;
        }
    }
    const MapTables mt{nullptr, nullptr, SetStates{nullptr, nullptr, nullptr, 0}};
    uint32_t i = 0;
    for (; i < P; ++i) {
        node_out[i] = -1;
        std::memset(&map_out[i], 0, sizeof(nhdfit_mapping));
        if (place_out) std::memset(&place_out[i], 0, sizeof(nhdfit_placement));
        status_out[i] = 0;
        const uint64_t sa = score_a[i];
        if (!sa) continue;
        const int64_t winner_a = (int64_t)(NHDFIT_SCORE_INDEX(sa) - global_base);
        int64_t nd = -1;
        for (int pass = (sa >> 63) ? 0 : 1; pass < 2 && nd < 0; ++pass) {
            const bool pref = pass == 0;
            const int64_t from = pref ? winner_a : ((sa >> 63) ? 0 : winner_a);
            for (uint32_t c = (uint32_t)(from >> 6); c < chunks && nd < 0; ++c) {
                uint64_t w = rows[(size_t)c * P + i];
                if (c == (uint32_t)(from >> 6)) w &= ~0ull << (from & 63);
                while (w && nd < 0) {
                    const uint32_t v = c * 64 + (uint32_t)__builtin_ctzll(w);
                    if (!pref || !(p2[v].flags & NHDFIT_NF_HAS_GPU)) nd = v;
                    w &= w - 1;
                }
            }
        }
        if (nd < 0) continue;
        const uint32_t v = (uint32_t)nd, t = i / kTile;
        NodeState st{p0[v], p1[v], p2[v], p3[v], p4[v]};
        nhdfit_detail dd = det[v];
        const uint32_t bits = nic_assignment_bits(img[t].data(), L[t], i % kTile, reqs[i].map_type == NHDFIT_MAP_PCI, st.p3);
        nhdfit_placement pl;
        std::memset(&pl, 0, sizeof pl);
        node_out[i] = (int64_t)global_base + nd;
        if (map_on_state(reqs[i], st, dd, caps, bits, mt, map_out[i])) status_out[i] = commit_node(st, dd, reqs[i], map_out[i], now, sigs, pl);
        else { status_out[i] = kCommitWouldRaise; pl.status = kCommitWouldRaise; }
        if (place_out) place_out[i] = pl;
        p0[v] = st.p0; p1[v] = st.p1; p2[v] = st.p2; p3[v] = st.p3; p4[v] = st.p4; det[v] = dd;
        if (status_out[i] == kCommitNewSig) { ++i; break; }
        const NodeIdx ni = node_index(st.p0, st.p1, st.p2, st.p4, fcmax + 1, fgmax + 1, ngs);
        const bool busy = (now - st.p4.busy_time) < kMinBusySecs;
        for (uint32_t tt = 0; tt < tiles; ++tt) {
            const uint64_t word = node_word_cold(img[tt].data(), L[tt], ni, st.p3, busy, m_need[tt], m_pci[tt]);
            for (uint32_t j = 0; j < (uint32_t)kTile && tt * kTile + j < P; ++j)
                if (!(word >> j & 1)) rows[(size_t)(v >> 6) * P + tt * kTile + j] &= ~(1ull << (v & 63));
        }
    }
    return i;
}

// the commit step alone (nhdfit_commit) on one node record
int hh_commit(nhdfit_plane0* p0, nhdfit_plane1* p1, nhdfit_plane2* p2, nhdfit_plane3* p3, nhdfit_plane4* p4, nhdfit_detail* det,
              const nhdfit_req* req, const nhdfit_mapping* map, double busy_time,
              const uint32_t* sig_off, uint32_t nsig, const uint32_t* pool_off, const uint8_t* pool_glimit, const nhdfit_cc* cc,
              nhdfit_placement* out) {
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
    NodeState st{*p0, *p1, *p2, *p3, *p4};
    const int rc = commit_node(st, *det, *req, *map, busy_time, SigTable{skeys.data(), sids.data(), slots - 1}, *out);
    *p0 = st.p0; *p1 = st.p1; *p2 = st.p2; *p3 = st.p3; *p4 = st.p4;
    return rc;
}

// list(set(codes inserted in order)) under the CPython model; tuples of length `len`, digits base `base`.
int hh_set_list(const int16_t* codes, int n, int len, int base, int16_t* out) {
    PySet s;
    ps_init(s);
    for (int i = 0; i < n; ++i) ps_add(s, codes[i], py_tuple_hash((uint32_t)codes[i], len, base));
    return ps_list(s, out);
}

// list(set(a) & set(b) & set(c))
int hh_set_isect3(const int16_t* a, int na, const int16_t* b, int nb, const int16_t* c, int nc, int len, int base, int16_t* out) {
    PySet sa, sb, sc, ab, abc;
    ps_init(sa); ps_init(sb); ps_init(sc);
    for (int i = 0; i < na; ++i) ps_add(sa, a[i], py_tuple_hash((uint32_t)a[i], len, base));
    for (int i = 0; i < nb; ++i) ps_add(sb, b[i], py_tuple_hash((uint32_t)b[i], len, base));
    for (int i = 0; i < nc; ++i) ps_add(sc, c[i], py_tuple_hash((uint32_t)c[i], len, base));
    ps_intersect(sa, sb, ab);
    ps_intersect(ab, sc, abc);
    return ps_list(abc, out);
}

// the register-resident model
int hh_small_set_list(const int16_t* codes, int n, int len, int base, int16_t* out) {
    SmallSet s = ss_make(len, base);
    for (int i = 0; i < n; ++i) s = ss_add(s, codes[i]);
    return ss_list(s, out);
}

int hh_small_isect3(const int16_t* a, int na, const int16_t* b, int nb, const int16_t* c, int nc, int len, int base, int16_t* out) {
    SmallSet sa = ss_make(len, base), sb = ss_make(len, base), sc = ss_make(len, base);
    for (int i = 0; i < na; ++i) sa = ss_add(sa, a[i]);
    for (int i = 0; i < nb; ++i) sb = ss_add(sb, b[i]);
    for (int i = 0; i < nc; ++i) sc = ss_add(sc, c[i]);
    return ss_list(ss_intersect(ss_intersect(sa, sb), sc), out);
}

uint64_t hh_tuple_hash(uint32_t code, int len, int base) { return py_tuple_hash(code, len, base); }

// the table of ascending-filled sets (AscEntry), built by the model itself exactly as k_build_asc does on the device
static const AscEntry* host_asc_table() {
    static std::vector<AscEntry> t;
    if (t.empty()) {
        t.resize(kAscEntries);
        for (int len = 1; len <= 4; ++len)
            for (uint32_t sub = 0; sub < (1u << (1u << len)); ++sub) t[kAscOffset[len] + sub] = asc_entry_build(len, sub);
    }
    return t.data();
}

// list(set filled with `subset` in ascending order) read back from the table
int hh_asc_set_list(uint32_t subset, int len, int base, int16_t* out) {
    return ss_list(ss_from_asc(host_asc_table(), len, base, subset), out);
}

// choose_tuples (register model) with / without the table; returns ok, writes gcode / ccode
int hh_choose(int G, int U, uint32_t sg, uint32_t sc, uint32_t nic, int use_table, uint32_t* gcode, int* ccode) {
    uint32_t g = 0; int c = -1;
    const bool ok = choose_tuples<SmallOps>(G, U, sg, sc, nic, g, c, use_table ? host_asc_table() : nullptr);
    *gcode = g; *ccode = c;
    return ok ? 1 : 0;
}

// tabulated choose_tuples (U = 2, G <= 2): table built exactly as k_build_choose does; returns the result word
static const uint8_t* host_choose_table() {
    static std::vector<uint8_t> t;
    if (t.empty()) {
        t.resize(kChooseEntries);
        for (uint32_t e = 0; e < kChooseEntries; ++e) t[e] = choose_entry_build(host_asc_table(), e);
    }
    return t.data();
}
int hh_choose_tabulated(int G, int U) { return choose_tabulated(G, U) ? 1 : 0; }
uint32_t hh_choose_from_table(int G, uint32_t sg, uint32_t sc, uint32_t nic) { return choose_from_table(host_choose_table(), G, sg, sc, nic); }

// choose_tuples for G = 3, U = 2 through the set-layout state machine (set_states.h)
static const SetStates& host_set_states() {
    static std::vector<uint64_t> info;
    static std::vector<uint32_t> next, asc;
    static SetStates t{};
    if (info.empty()) {
        build_set_states(info, next, asc);
        t = SetStates{info.data(), next.data(), asc.data(), (uint32_t)info.size()};
    }
    return t;
}
uint32_t hh_set_state_count() { return host_set_states().n; }
uint32_t hh_choose_g3(uint32_t sg, uint32_t sc, uint32_t nic) { return choose_g3(host_set_states(), host_asc_table(), sg, sc, nic); }

// the same through the generic (PySet) model
int hh_choose_generic(int G, int U, uint32_t sg, uint32_t sc, uint32_t nic, uint32_t* gcode, int* ccode) {
    uint32_t g = 0; int c = -1;
    const bool ok = choose_tuples<GenericOps>(G, U, sg, sc, nic, g, c);
    *gcode = g; *ccode = c;
    return ok ? 1 : 0;
}

}  // extern "C"

// first_nic_choice (depth-first with prefix pruning) next to the plain enumeration it replaces, on one fabricated
// winner: returns (found_pruned) | (found_plain << 1); the picks go to out_pruned / out_plain.
extern "C" int hh_first_nic_choice(const nhdfit_req* r, const nhdfit_detail* d, const double* caps, uint32_t gcode, int pci,
                                   int8_t* out_pruned, int8_t* out_plain) {
    WinnerState w{};
    w.U = d->numa_nodes;
    w.d = d;
    w.caps = caps;
    for (int g = 0; g < kMaxG; ++g) out_pruned[g] = out_plain[g] = -1;
    const bool a = first_nic_choice(*r, w, gcode, pci != 0, out_pruned);
    const bool b = first_nic_choice_plain(*r, w, gcode, pci != 0, out_plain);
    return (a ? 1 : 0) | (b ? 2 : 0);
}

// CPU twin of nhdfit_apply_deltas (K3): the deltas in array order on host copies of the planes.
extern "C" int hh_apply_deltas(nhdfit_plane0* p0, nhdfit_plane1* p1, nhdfit_plane2* p2, nhdfit_plane3* p3, nhdfit_plane4* p4, nhdfit_detail* det,
                               nhdfit_origin* origin, uint32_t n_nodes, const nhdfit_delta* deltas, uint32_t n,
                               const uint32_t* sig_off, uint32_t nsig, const uint32_t* pool_off, const uint8_t* pool_glimit, const nhdfit_cc* cc,
                               uint8_t* status_out) {
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
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t v = deltas[i].node;
        if (v >= n_nodes) return -1;
        NodeState s;
        s.p0 = p0[v]; s.p1 = p1[v]; s.p2 = p2[v]; s.p3 = p3[v]; s.p4 = p4[v];
        status_out[i] = (uint8_t)apply_delta(s, det[v], origin[v], deltas[i], sigs);
        p0[v] = s.p0; p1[v] = s.p1; p2[v] = s.p2; p3[v] = s.p3; p4[v] = s.p4;
    }
    return 0;
}

// ---- the general path (wide_core.h) on the host ------------------------------------------------------------------------
#include "../../nhd_amd/csrc/wide_core.h"

// what k_wide_eval computes: verdict byte per (wide node, pod), scores max-merged into score[] (caller's pod order)
extern "C" void hh_wide_eval(const nhdfit_wide_node* wide, uint32_t n_wide, const nhdfit_req* reqs, uint32_t P, double now, const double* caps,
                             const uint64_t* cand, uint64_t global_base, uint8_t* fits /* [n_wide][P] */, uint64_t* score /* in-out */,
                             const nhdfit_wide_share* share /* optional [n_wide]: ENABLE_SHARING */) {
    const double busy_from = busy_threshold(now);
    for (uint32_t w = 0; w < n_wide; ++w)
        for (uint32_t i = 0; i < P; ++i) {
            const nhdfit_wide_node& n = wide[w];
            bool ok = !(cand && !(cand[n.index >> 6] >> (n.index & 63) & 1ull));
            const bool busy = n.busy_time >= busy_from;
            if (busy != ((now - n.busy_time) < kMinBusySecs)) std::abort();       // both forms of IsBusy agree
            ok = ok && wide_fits(n, reqs[i], busy, WideCaps(caps, share ? share + w : nullptr));
            fits[(size_t)w * P + i] = ok ? 1 : 0;
            if (!ok) continue;
            uint32_t want = 0;
            for (uint32_t g = 0; g < reqs[i].n_groups; ++g) want += reqs[i].gpus[g];
            const uint64_t s = score_of(want == 0 && n.n_gpus == 0, global_base + n.index);
            if (s > score[i]) score[i] = s;
        }
}
extern "C" int hh_wide_map(const nhdfit_wide_node* n, const nhdfit_req* r, const double* caps, nhdfit_mapping* out, const nhdfit_wide_share* share) {
    std::vector<int16_t> scratch(kWideScratchWords);
    return wide_map(*n, *r, WideCaps(caps, share), scratch.data(), *out);
}
extern "C" int hh_wide_commit(nhdfit_wide_node* n, const nhdfit_req* r, const nhdfit_mapping* m, double busy_time, nhdfit_wide_placement* out, nhdfit_wide_share* share) {
    const int st = wide_commit(*n, *r, *m, busy_time, *out, share);
    out->pod = 0; out->node = n->index;
    return st;
}
extern "C" uint64_t hh_wide_tuple_hash(uint32_t code, uint32_t len, uint32_t U) { return wide_tuple_hash(code, len, U); }
// list(set) after adding `codes` one by one; returns the length, -1 on table overflow
extern "C" int hh_wide_set_list(const int16_t* codes, int n, uint32_t len, uint32_t U, int16_t* out) {
    std::vector<int16_t> mem(kWideSetSlotsC), tmp(kWideSetSlotsC);
    WideSet s;
    ws_init(s, mem.data(), kWideSetSlotsC, len, U);
    for (int i = 0; i < n; ++i) ws_add(s, codes[i], tmp.data());
    if (s.overflow) return -1;
    int k = 0;
    for (int32_t i = ws_next(s, 0); i >= 0; i = ws_next(s, i + 1)) out[k++] = s.key[i];
    return k;
}
// list(set(a) & set(b) & set(c)), each operand built by adding its codes in the order given
extern "C" int hh_wide_isect3(const int16_t* a, int na, const int16_t* b, int nb, const int16_t* c, int nc, uint32_t len, uint32_t U, int16_t* out) {
    std::vector<int16_t> mem(5 * kWideSetSlotsC), tmp(kWideSetSlotsC);
    WideSet A, B, C, AB, ABC;
    ws_init(A, mem.data(), kWideSetSlotsC, len, U); ws_init(B, mem.data() + kWideSetSlotsC, kWideSetSlotsC, len, U);
    ws_init(C, mem.data() + 2 * kWideSetSlotsC, kWideSetSlotsC, len, U); ws_init(AB, mem.data() + 3 * kWideSetSlotsC, kWideSetSlotsC, len, U);
    ws_init(ABC, mem.data() + 4 * kWideSetSlotsC, kWideSetSlotsC, len, U);
    for (int i = 0; i < na; ++i) ws_add(A, a[i], tmp.data());
    for (int i = 0; i < nb; ++i) ws_add(B, b[i], tmp.data());
    for (int i = 0; i < nc; ++i) ws_add(C, c[i], tmp.data());
    ws_intersect(A, B, AB, tmp.data());
    ws_intersect(AB, C, ABC, tmp.data());
    int k = 0;
    for (int32_t i = ws_next(ABC, 0); i >= 0; i = ws_next(ABC, i + 1)) out[k++] = ABC.key[i];
    return k;
}

// ---- big requests (5..8 processing groups): the general path over every node (wide_core.h templates, commit_core.h commit_node_t) ----
// fits [n][P] by node index (a wide node's verdict at its own index), scores max-merged; flags[1] = a pair ran out of NIC budget
extern "C" void hh_big_eval(const nhdfit_plane0* p0, const nhdfit_plane1* p1, const nhdfit_plane2* p2, const nhdfit_plane3* p3,
                            const nhdfit_plane4* p4, const nhdfit_detail* det, uint32_t n, const nhdfit_wide_node* wide, uint32_t n_wide,
                            const nhdfit_big_req* reqs, uint32_t P, double now, const double* caps, const uint64_t* cand,
                            uint64_t global_base, uint8_t* fits, uint64_t* score, uint32_t* flags, uint32_t budget /* 0: NHDFIT_BIG_NIC_BUDGET */,
                            const nhdfit_wide_share* share /* optional [n_wide] */) {
    const double busy_from = busy_threshold(now);
    for (uint32_t v = 0; v < n + n_wide; ++v) {
        nhdfit_wide_node view;
        if (v < n) wide_view(p0[v], p1[v], p2[v], p3[v], p4[v], det[v], v, view);
        else view = wide[v - n];
        if (cand && !(cand[view.index >> 6] >> (view.index & 63) & 1ull)) continue;
        const bool busy = view.busy_time >= busy_from;
        for (uint32_t i = 0; i < P; ++i) {
            NicSearch ns{budget ? budget : NHDFIT_BIG_NIC_BUDGET, false};
            const bool ok = wide_fits(view, reqs[i], busy, WideCaps(caps, share && v >= n ? share + (v - n) : nullptr), &ns);
            if (ns.exhausted) flags[1] = 1;
            flags[2] += (budget ? budget : NHDFIT_BIG_NIC_BUDGET) - ns.left;      // search steps spent (diagnostics)
            if (!ok) continue;
            fits[(size_t)view.index * P + i] = 1;
            uint32_t want = 0;
            for (uint32_t g = 0; g < reqs[i].n_groups; ++g) want += reqs[i].gpus[g];
            const uint64_t s = score_of(want == 0 && view.n_gpus == 0, global_base + view.index);
            if (s > score[i]) score[i] = s;
        }
    }
}
// mapping of one winner: `wide` != NULL: that record, else node `v` of the planes
extern "C" int hh_big_map(const nhdfit_plane0* p0, const nhdfit_plane1* p1, const nhdfit_plane2* p2, const nhdfit_plane3* p3,
                          const nhdfit_plane4* p4, const nhdfit_detail* det, uint32_t v, const nhdfit_wide_node* wide,
                          const nhdfit_big_req* r, const double* caps, nhdfit_big_mapping* out, const nhdfit_wide_share* share) {
    nhdfit_wide_node view;
    if (wide) view = *wide; else wide_view(p0[v], p1[v], p2[v], p3[v], p4[v], det[v], v, view);
    const uint32_t U = view.numa_nodes ? view.numa_nodes : 1, G = r->n_groups <= NHDFIT_BIG_MAX_GROUPS ? r->n_groups : NHDFIT_BIG_MAX_GROUPS;
    std::vector<int32_t> scratch(big_scratch_words(U, G));           // (the device sizes them for the mirror's widest node: any size >= the need gives the same answer)
    return wide_map(view, *r, WideCaps(caps, wide ? share : nullptr), scratch.data(), *out, (int32_t)wide_table_slots(wide_ipow(U, G)), (int32_t)wide_table_slots(wide_ipow(U, G + 1)));
}
extern "C" int hh_big_commit_wide(nhdfit_wide_node* n, const nhdfit_big_req* r, const nhdfit_big_mapping* m, double busy_time, nhdfit_big_placement* out,
                                  nhdfit_wide_share* share) {
    const int st = wide_commit(*n, *r, *m, busy_time, *out, share);
    out->pod = 0; out->node = n->index;
    return st;
}
extern "C" int hh_big_commit(nhdfit_plane0* p0, nhdfit_plane1* p1, nhdfit_plane2* p2, nhdfit_plane3* p3, nhdfit_plane4* p4, nhdfit_detail* det,
                             const nhdfit_big_req* req, const nhdfit_big_mapping* map, double busy_time,
                             const uint32_t* sig_off, uint32_t nsig, const uint32_t* pool_off, const uint8_t* pool_glimit, const nhdfit_cc* cc,
                             nhdfit_big_placement* out) {
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
    NodeState st{*p0, *p1, *p2, *p3, *p4};
    std::memset(out, 0, sizeof *out);
    const int rc = commit_node_t<nhdfit_big_req, nhdfit_big_placement>(st, *det, *req, *map, busy_time, SigTable{skeys.data(), sids.data(), slots - 1}, *out);
    *p0 = st.p0; *p1 = st.p1; *p2 = st.p2; *p3 = st.p3; *p4 = st.p4;
    return rc;
}
// list(set) after adding `codes` one by one with the big requests' tables (int32 keys, slots sized by wide_table_slots); -1 on overflow
extern "C" int hh_big_set_list(const int32_t* codes, int n, uint32_t len, uint32_t U, int32_t* out) {
    const uint32_t slots = wide_table_slots((uint32_t)n);
    std::vector<int32_t> mem(slots), tmp(slots);
    WideSetT<int32_t> s;
    ws_init(s, mem.data(), (int32_t)slots, len, U);
    for (int i = 0; i < n; ++i) ws_add(s, codes[i], tmp.data());
    if (s.overflow) return -1;
    int k = 0;
    for (int32_t i = ws_next(s, 0); i >= 0; i = ws_next(s, i + 1)) out[k++] = s.key[i];
    return k;
}
extern "C" uint32_t hh_table_slots(uint32_t keys) { return wide_table_slots(keys); }
