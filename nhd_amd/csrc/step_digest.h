// step_digest.h - request digest role of the step kernel (k_step): pod requests -> bit-sliced tile image.
// Device code of libnhdfit.so; included by nhdfit.hip inside its anonymous namespace, in this order: step_digest.h,
// step_fit.h, step_map.h, step_kernel.h, seq_kernel.h (one translation unit: the roles are fused into one kernel).
// gfx950 only.
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
    // The dictionary by POOL TYPE (dict_stream.h, built by nhdfit_set_dictionary): a signature's reach family is the disjoint union
    // over its pools, the operation is commutative and associative, and a dictionary holds few distinct pools (config 5: eight
    // one-NIC pools per PCI-mode signature, a handful of kinds).  Per pod the digest forms each type's family and as many of its
    // 2-, 3-, 4-fold unions as the dictionary asks for, each in a slot of LDS, and a signature is the list of the slots to unite:
    // one union per distinct type instead of two per pool.  Stream layout: dict_stream.h.
    const uint16_t* flat2;
    uint32_t flat2_words;            // 0: not available (more than kPoolSlots slots, offsets beyond 16 bits)
};
constexpr uint32_t kPoolSlots = 128;                 // unions a digest block keeps in LDS (x 64 pods x 2 bytes = 16 KB)

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
    uint32_t wc_parts;               // blocks per tile that share its CPU rows (free-core count c = part mod wc_parts)
    uint32_t sig_parts;              // blocks per tile that share its NIC signature rows (signature = part mod sig_parts; 1 for small
                                     // dictionaries); the last of them to finish derives the X rows
    uint32_t* count;                 // [tiles] arrival counters of those blocks, zero before and after a launch
    // The signatures some node class of the mirror refers to, ascending (ensure_records, nhdfit.hip), or null = all of them.  A
    // dictionary closed under claims (mode B needs every state a commit can produce) holds many signatures no node is in yet -
    // config 5: 121 of 272 - and a snapshot step reads R rows through node classes (X rows) and winners (mapping) only.
    const uint16_t* sig_list;
    uint32_t n_sig_list;
};
constexpr uint32_t kDictLdsWords = 6144;             // 12 KB for the staged signature stream (c5: 151 signatures = 1.5 K words)
// the request copies come last: once the covers are formed they are dead, and the per-type unions (kPoolSlots x 64 x 2 bytes) take
// their place plus the pad behind them
constexpr size_t kPoolLds = (size_t)kPoolSlots * kTile * sizeof(uint16_t);
constexpr size_t kDigestLds = lds_slice(kTile * sizeof(PodSums)) +
                              lds_slice(kTile * NHDFIT_MAX_CLASSES * (kMaxG + 1) * sizeof(uint16_t)) + lds_slice(kTile * sizeof(PodHeader)) +
                              lds_slice(kDictLdsWords * sizeof(uint16_t)) + 16 +    // (+ the last-arriver flag; no static LDS in the step kernel:
                                                                                     //  its dynamic allocation may ask for all 160 KB)
                              (lds_slice(kTile * sizeof(PaddedReq)) > kPoolLds ? lds_slice(kTile * sizeof(PaddedReq)) : kPoolLds);
constexpr uint32_t kWcPartsDefault = 4;              // blocks per tile: part 0 = GPU / NIC rows (cold section + X), parts 1..wc_parts = CPU rows, the last one also HP / GX

// Request digest, 1 + wc_parts blocks per 64-pod tile: per-pod subset sums / NIC covers in LDS, then the table rows
// (lane = pod, one ballot per assignment).  The role is a chain of dependent phases, not a lot of work: it is cut
// into parts by table so that the chain of each block stays short.
template <int THREADS>
__device__ __forceinline__ void role_digest(const DigestArgs& a, uint32_t blk, uint8_t* lds) {
    PodSums* s_sum = carve<PodSums>(lds, kTile);
    uint16_t (*s_cover)[NHDFIT_MAX_CLASSES][kMaxG + 1] =
        reinterpret_cast<uint16_t (*)[NHDFIT_MAX_CLASSES][kMaxG + 1]>(carve<uint16_t>(lds, kTile * NHDFIT_MAX_CLASSES * (kMaxG + 1)));
    PodHeader* s_hdr = carve<PodHeader>(lds, kTile);
    uint16_t* s_flat = carve<uint16_t>(lds, kDictLdsWords);
    uint32_t& s_last = *carve<uint32_t>(lds, 4);
    PaddedReq* s_req = reinterpret_cast<PaddedReq*>(lds);                 // (last: see kDigestLds)
    uint16_t* s_pw = reinterpret_cast<uint16_t*>(lds);                    // [slot][pod]: a pool type's family or one of its k-fold unions - over s_req, after the covers

    // blocks of a tile: sig_parts blocks for the GPU / NIC rows (part 0), then wc_parts blocks for the CPU rows (parts 1 ..)
    const uint32_t per_tile = a.sig_parts + a.wc_parts, tile = blk / per_tile, idx = blk % per_tile;
    const uint32_t part = idx < a.sig_parts ? 0u : idx - a.sig_parts + 1u, sig_part = idx < a.sig_parts ? idx : 0u;
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
        if (part == 0 && sig_part == 0) {
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
    auto emit_row = [&](uint8_t* row, uint32_t v) __attribute__((always_inline)) {
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
        // (the tile's row width W is uniform over the block: a one-group tile compares 2 demands per row, not 16 - the guards below are
        // scalar branches.  One instantiation on purpose: four, one per width, cost the whole step kernel 1.5 KB of scratch per lane)
        for (uint32_t g = wave; g < 8; g += NW) {
            const uint32_t u = g >> 2, m = (g >> 1) & 1, smt = g & 1;
            uint32_t t[1 << kMaxG];
#pragma unroll
            for (uint32_t p = 0; p < (1u << kMaxG); ++p) {
                t[p] = 0xFFFFFFFFu;
                if (p < W && valid && p < s_sum[lane].W) {
                    const uint32_t* sum = smt ? s_sum[lane].cpu_smt : s_sum[lane].cpu_nosmt;
                    const uint32_t extra = m ? (smt ? s_sum[lane].misc_smt : s_sum[lane].misc_nosmt) : 0;
                    t[p] = sum[u ? p : (~p & s_sum[lane].full)] + extra;
                }
            }
            uint8_t* base = hot + (u ? L.hot_wc1 : L.hot_wc0) + smt * L.fc_dim * L.wc_stride + m * L.row;
            for (uint32_t c = part - 1; c < L.fc_dim; c += a.wc_parts) {
                uint32_t v = 0;
#pragma unroll
                for (uint32_t p = 0; p < (1u << kMaxG); ++p)
                    if (p < W) v |= (t[p] <= c ? 1u : 0u) << p;
                emit_row(base + c * L.wc_stride, v);
            }
        }
        if (part != a.wc_parts) return;
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
    const bool typed = a.d.flat2_words != 0 && a.d.flat2_words <= kDictLdsWords;      // block-uniform
    const bool staged = !typed && a.d.flat_words != 0 && a.d.flat_words <= kDictLdsWords;
    if (typed)
        for (uint32_t w = tid; w < a.d.flat2_words / 2; w += THREADS)       // (the streams are padded to an even word count)
            reinterpret_cast<uint32_t*>(s_flat)[w] = reinterpret_cast<const uint32_t*>(a.d.flat2)[w];
    if (staged)
        for (uint32_t w = tid; w < a.d.flat_words / 2; w += THREADS)
            reinterpret_cast<uint32_t*>(s_flat)[w] = reinterpret_cast<const uint32_t*>(a.d.flat)[w];
    auto covers_and_sig_rows = [&](auto width) __attribute__((always_inline)) {
        constexpr uint32_t WW = decltype(width)::value;
        for (uint32_t w = tid; w < kTile * a.d.ncls; w += THREADS) {
            const uint32_t j = w % kTile, c = w / kTile;
            if (s_hdr[j].flags & kPodValid) class_cover_w<WW>(s_req[j].r, a.d.caps[c], s_sum[j].W, s_sum[j].G, s_cover[j][c]);
        }
        __syncthreads();
        if (typed) {
            // (A) every pool type's family and the k-fold unions the dictionary asks for, lane = pod, a type per wavefront iteration
            auto bword = [&](uint32_t i) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)s_flat[i]); };
            const uint32_t ntypes = bword(0), nsig_d = a.d.sig.nsig;
            const uint32_t t_off = 2, s_off = t_off + ntypes + 1, recs = s_off + nsig_d + 1;
            for (uint32_t t = wave; t < ntypes; t += NW) {
                uint32_t at = recs + bword(t_off + t);
                const uint32_t head = bword(at++), ncc = head & 0xFFu, glimit = head >> 8;
                const uint32_t ks = bword(at++), kmax = ks >> 8, first = ks & 0xFFu;
                uint32_t pool = 1;
                for (uint32_t k = 0; k < ncc; ++k) {
                    const uint32_t e = bword(at++), cnt = e & 0xFFu, cls = e >> 8;
                    pool = dunion_n<WW>(pool, s_cover[lane][cls][cnt > (uint32_t)kMaxG ? kMaxG : cnt]);
                }
                if (glimit != NHDFIT_GLIMIT_NONE) pool &= size_le_mask(s_sum[lane].W, glimit);
                uint32_t pw = pool;
                for (uint32_t k = 0; k < kmax; ++k) {
                    s_pw[(first + k) * kTile + lane] = (uint16_t)pw;              // (s_req is dead: the covers are formed, the barrier passed)
                    pw = dunion_n<WW>(pw, pool);
                }
            }
            __syncthreads();
            // (B) a signature = the union over its slots; record offsets and records by bulk load + v_readlane
            const uint32_t sig_first = sig_part * NW + wave, sig_step = a.sig_parts * NW;
            uint32_t offs_lo = 0, offs_hi = 0, seq = 0;
            const uint32_t n_sigs = a.sig_list ? a.n_sig_list : L.nsig;
            uint32_t sig_ids = 0;                                                   // lane i: the signature at position idx + i * sig_step
            for (uint32_t idx = sig_first; idx < n_sigs; idx += sig_step, ++seq) {
                if ((seq & 63u) == 0) {
                    const uint32_t at = idx + lane * sig_step;
                    const uint32_t mine = at < n_sigs ? (a.sig_list ? (uint32_t)a.sig_list[at] : at) : L.nsig;
                    sig_ids = mine;
                    offs_lo = mine < L.nsig ? s_flat[s_off + mine] : 0u;
                    offs_hi = mine < L.nsig ? s_flat[s_off + mine + 1] : 0u;
                }
                const uint32_t sig = (uint32_t)__builtin_amdgcn_readlane((int)sig_ids, (int)(seq & 63u));
                const uint32_t rec0 = (uint32_t)__builtin_amdgcn_readlane((int)offs_lo, (int)(seq & 63u));
                const uint32_t rec1 = (uint32_t)__builtin_amdgcn_readlane((int)offs_hi, (int)(seq & 63u));
                const uint32_t len = rec1 - rec0;                                   // 1 + entries, <= 64 (dict_stream.h)
                const uint32_t my_word = lane < len ? s_flat[recs + rec0 + lane] : 0u;
                const uint32_t nent = (uint32_t)__builtin_amdgcn_readlane((int)my_word, 0);
                uint32_t reach = 1;
                for (uint32_t e = 0; e < nent; ++e) {
                    const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)my_word, (int)(1 + e));
                    reach = dunion_n<WW>(reach, s_pw[slot * kTile + lane]);
                }
                if (!valid) reach = 0;
                emit_row(img + L.off_r0 + sig * L.row, valid ? entry_r(reach, s_sum[lane].W, 0) : 0u);
                emit_row(img + L.off_r1 + sig * L.row, valid ? entry_r(reach, s_sum[lane].W, 1) : 0u);
            }
            return;
        }
        // Walking a signature's record word by word through LDS is a chain of dependent round trips (broadcast read, wait,
        // readfirstlane: ~17 of them for a signature of eight one-NIC pools, ~2 000 cycles before any union is formed - what made
        // config 5's digest the longest role of its step).  Instead: the offsets of the wavefront's next 64 signatures are read
        // with one load (lane = signature of the wavefront's sequence), a signature's whole record with one more (lane = word),
        // and the walk hands words to the scalar unit with v_readlane - two LDS round trips per signature.  Records longer than a
        // wavefront keep the word-by-word walk.
        const uint32_t sig_first = sig_part * NW + wave, sig_step = a.sig_parts * NW;
        uint32_t offs_lo = 0, offs_hi = 0;                                   // lane i: record start / end of signature sig_first + (base + i) * sig_step
        uint32_t seq = 0;
        const uint32_t n_sigs = a.sig_list ? a.n_sig_list : L.nsig;
        uint32_t sig_ids = 0;
        for (uint32_t idx = sig_first; idx < n_sigs; idx += sig_step, ++seq) {    // one reach family per (signature, pod), both sockets' rows from it
            uint32_t reach = 0;
            if ((seq & 63u) == 0) {
                const uint32_t at = idx + lane * sig_step;
                sig_ids = at < n_sigs ? (a.sig_list ? (uint32_t)a.sig_list[at] : at) : L.nsig;
            }
            const uint32_t sig = (uint32_t)__builtin_amdgcn_readlane((int)sig_ids, (int)(seq & 63u));
            if (staged) {
                if ((seq & 63u) == 0) {
                    const uint32_t mine = sig_ids;
                    offs_lo = mine < L.nsig ? s_flat[mine] : 0u;
                    offs_hi = mine + 1 < L.nsig ? s_flat[mine + 1] : a.d.flat_words - (a.d.sig.nsig + 1);   // (the last record ends with the stream, padding included)
                }
                const uint32_t rec0 = (uint32_t)__builtin_amdgcn_readlane((int)offs_lo, (int)(seq & 63u));
                const uint32_t rec1 = (uint32_t)__builtin_amdgcn_readlane((int)offs_hi, (int)(seq & 63u));
                const uint32_t base = a.d.sig.nsig + 1 + rec0;
                reach = 1;
                if (rec1 - rec0 <= 64u) {
                    const uint32_t my_word = lane < rec1 - rec0 ? s_flat[base + lane] : 0u;
                    auto word = [&](uint32_t i) { return (uint32_t)__builtin_amdgcn_readlane((int)my_word, (int)i); };
                    uint32_t at = 0;
                    const uint32_t npools = word(at++);
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
                } else {
                    // the record is read with the same address in every lane (LDS broadcast); readfirstlane hands the loop
                    // bounds to the scalar unit so the walk stays wave-uniform
                    auto word = [&](uint32_t i) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)s_flat[i]); };
                    uint32_t at = base;
                    const uint32_t npools = word(at++);
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
    if (sig_part == 0)
        for (uint32_t k = wave; k < 2 * L.fg_dim; k += NW) {
            const uint32_t u = k >= L.fg_dim, f = u ? k - L.fg_dim : k;
            emit_row(img + (u ? L.off_a1 : L.off_a0) + f * L.row, valid ? entry_a(s_sum[lane], u, f) : 0u);
        }
    // The X rows need every cold row of the tile.  One block per tile: a barrier.  Several (large dictionaries - config 5
    // has 151 signatures and as many again under claims): each block publishes its rows and takes a ticket; the last one
    // to arrive derives the X rows (reading the others' rows past its own L1) and leaves the counter at zero.
    if (a.sig_parts > 1) {
        __threadfence();
        __syncthreads();
        if (tid == 0) {
            const uint32_t t = atomicAdd(&a.count[tile], 1u);
            s_last = t == a.sig_parts - 1u ? 1u : 0u;
            if (s_last) a.count[tile] = 0u;
        }
        __syncthreads();
        if (!s_last) return;
        __threadfence();
    } else __syncthreads();                                     // the block reads back the cold rows it just wrote
    // hot rows X[class] = A_u[f] & (PCI-mode pods: R_u[sigPCI], NUMA-mode pods: R_u[sigNUMA]) - pure word
    // operations on the cold rows, one lane per (class, assignment)
    auto cold64 = [&](const uint8_t* base, uint32_t off) {
        const uint64_t* ptr = reinterpret_cast<const uint64_t*>(base + off);
        return a.sig_parts > 1 ? __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *ptr;
    };
    const uint64_t m_pci = __ballot((s_hdr[lane].flags & kPodPci) != 0);
    const uint32_t nx = a.nx[0] < L.x_cap ? a.nx[0] : L.x_cap;
    for (uint32_t i = tid; i < nx * W; i += THREADS) {
        const uint32_t k = i / W, p = i % W;
        const uint64_t key = a.xcls[k];
        const uint32_t u = xkey_u(key);
        const uint8_t* rbase = img + (u ? L.off_r1 : L.off_r0) + p * 8;
        const uint64_t av = cold64(img, (u ? L.off_a1 : L.off_a0) + xkey_f(key) * L.row + p * 8);
        const uint64_t rn = cold64(rbase, xkey_sig_numa(key) * L.row), rp = cold64(rbase, xkey_sig_pci(key) * L.row);
        *reinterpret_cast<uint64_t*>(hot + L.hot_x + k * L.x_stride + p * 8) = av & ((rp & m_pci) | (rn & ~m_pci));
    }
}
