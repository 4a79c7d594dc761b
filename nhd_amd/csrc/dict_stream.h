// dict_stream.h - the NIC-signature dictionary as streams of 16-bit words for the digest role (host code, shared by
// nhdfit_set_dictionary and the host twin of the tests).
//
// DictView::flat2 (step_digest.h): the dictionary BY POOL TYPE.  A signature's reach family is the disjoint union over its
// pools; the operation is commutative and associative and a dictionary holds few distinct pools, so the digest forms each type's
// family and its 2-, 3-, 4-fold unions once per pod - as many of them as the dictionary ever asks for (a type only ever met once
// per signature needs one; more than four pools of a kind add nothing: a pod has at most four groups to spread over them) - each
// in a SLOT of the digest block's LDS, and a signature is the list of the slots to unite.
//   [0] ntypes, [1] nslots, type offsets [ntypes + 1], signature offsets [nsig + 1], then the records (offsets count from there):
//   type = { glimit << 8 | #cc, kmax << 8 | first slot, #cc x (cls << 8 | cnt) }   slot (first + k) = the (k + 1)-fold union, k < kmax
//   signature = { #entries, entries x slot }
// typed_reach() below is that arithmetic restated for the host (tests/harness compares it with the pool-by-pool sig_reach on every
// pod and signature of every CPU test); the kernel's form is role_digest's typed branch.
#pragma once
#include <algorithm>
#include <map>
#include <utility>
#include <vector>
#include "fit_core.h"

namespace nhdfit {

constexpr uint32_t kPoolSlotsMax = 128;              // == kPoolSlots of step_digest.h (unions a digest block keeps in LDS: 128 x 64 pods x 2 bytes)

// Returns the stream, or an empty vector when the dictionary does not fit the format (more than kPoolSlotsMax slots, a pool of
// more than 255 classes, offsets beyond 16 bits): the digest then walks pool by pool.
inline std::vector<uint16_t> build_typed_stream(const SigDict& d) {
    const uint32_t nsig = d.nsig;
    std::map<std::vector<uint16_t>, uint32_t> type_of;
    std::vector<std::vector<uint16_t>> types;                     // { head, cc words sorted }
    std::vector<uint32_t> kmax;
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> sig_ent(nsig);   // (type, count <= kMaxG) in order of first appearance
    for (uint32_t sg = 0; sg < nsig; ++sg) {
        auto& ent = sig_ent[sg];
        for (uint32_t pl = d.sig_off[sg]; pl < d.sig_off[sg + 1]; ++pl) {
            const uint32_t ncc_pl = d.pool_off[pl + 1] - d.pool_off[pl];
            if (ncc_pl == 0) continue;                            // a pool without NICs hosts the empty set only: the neutral element
            if (ncc_pl > 255u) return {};
            std::vector<uint16_t> body;
            for (uint32_t k = d.pool_off[pl]; k < d.pool_off[pl + 1]; ++k) body.push_back((uint16_t)((d.cc[k].cls & 0xFFu) << 8 | d.cc[k].cnt));
            std::sort(body.begin(), body.end());
            std::vector<uint16_t> rec{(uint16_t)(d.pool_glimit[pl] << 8 | ncc_pl)};
            rec.insert(rec.end(), body.begin(), body.end());
            auto it = type_of.find(rec);
            if (it == type_of.end()) {
                it = type_of.emplace(rec, (uint32_t)types.size()).first;
                types.push_back(rec);
                kmax.push_back(0);
            }
            bool seen = false;
            for (auto& e : ent)
                if (e.first == it->second) { if (e.second < (uint32_t)kMaxG) e.second++; seen = true; }
            if (!seen) ent.emplace_back(it->second, 1u);
        }
        for (auto& e : ent) kmax[e.first] = std::max(kmax[e.first], e.second);
    }
    const uint32_t nt = (uint32_t)types.size();
    std::vector<uint32_t> first(nt + 1, 0);
    for (uint32_t t = 0; t < nt; ++t) first[t + 1] = first[t] + kmax[t];
    if (first[nt] > kPoolSlotsMax || nt > 0xFFFFu) return {};
    std::vector<uint16_t> f2(2 + (nt + 1) + (nsig + 1), 0);
    f2[0] = (uint16_t)nt;
    f2[1] = (uint16_t)first[nt];
    const size_t recs = f2.size();
    for (uint32_t t = 0; t < nt; ++t) {
        if (f2.size() - recs > 0xFFFFu) return {};
        f2[2 + t] = (uint16_t)(f2.size() - recs);
        f2.push_back(types[t][0]);
        f2.push_back((uint16_t)(kmax[t] << 8 | first[t]));
        f2.insert(f2.end(), types[t].begin() + 1, types[t].end());
    }
    if (f2.size() - recs > 0xFFFFu) return {};
    f2[2 + nt] = (uint16_t)(f2.size() - recs);
    for (uint32_t sg = 0; sg < nsig; ++sg) {
        if (f2.size() - recs > 0xFFFFu || sig_ent[sg].size() > 63u) return {};      // (a record is read by one wavefront: lane = word)
        f2[2 + nt + 1 + sg] = (uint16_t)(f2.size() - recs);
        f2.push_back((uint16_t)sig_ent[sg].size());
        for (auto& e : sig_ent[sg]) f2.push_back((uint16_t)(first[e.first] + e.second - 1u));
    }
    if (f2.size() - recs > 0xFFFFu) return {};
    f2[2 + nt + 1 + nsig] = (uint16_t)(f2.size() - recs);
    if (f2.size() & 1) f2.push_back(0);
    return f2;
}

// reach family of signature `sig` for one pod from the typed stream; cover = [ncls][kMaxG + 1] (class_cover)
inline uint32_t typed_reach(const uint16_t* f2, uint32_t nsig, uint32_t sig, const uint16_t* cover, uint32_t W) {
    const uint32_t ntypes = f2[0], nslots = f2[1], t_off = 2, s_off = t_off + ntypes + 1, recs = s_off + nsig + 1;
    std::vector<uint32_t> slot(nslots ? nslots : 1, 0);           // (the kernel fills them once per pod and tile, phase A)
    for (uint32_t t = 0; t < ntypes; ++t) {
        uint32_t at = recs + f2[t_off + t];
        const uint32_t head = f2[at++], ncc = head & 0xFFu, glimit = head >> 8;
        const uint32_t ks = f2[at++], kmax = ks >> 8, first = ks & 0xFFu;
        uint32_t pool = 1;
        for (uint32_t i = 0; i < ncc; ++i) {
            const uint32_t e = f2[at++], cnt = e & 0xFFu, cls = e >> 8;
            pool = dunion(pool, cover[cls * (kMaxG + 1) + (cnt > (uint32_t)kMaxG ? kMaxG : cnt)], W);
        }
        if (glimit != NHDFIT_GLIMIT_NONE) pool &= size_le_mask(W, glimit);
        uint32_t pw = pool;
        for (uint32_t k = 0; k < kmax; ++k) { slot[first + k] = pw; pw = dunion(pw, pool, W); }
    }
    uint32_t at = recs + f2[s_off + sig];
    const uint32_t nent = f2[at++];
    uint32_t reach = 1;
    for (uint32_t e = 0; e < nent; ++e) reach = dunion(reach, slot[f2[at++]], W);
    return reach;
}

}  // namespace nhdfit
