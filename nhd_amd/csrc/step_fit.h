// step_fit.h - fit role of the step kernel: the P x N pass over node records against the staged tile image.
// Device code of libnhdfit.so; included by nhdfit.hip inside its anonymous namespace, in this order: step_digest.h,
// step_fit.h, step_map.h, step_kernel.h, seq_kernel.h (one translation unit: the roles are fused into one kernel).
// gfx950 only.
struct FitItem { uint32_t tile, wcls, c_begin, c_end; };
#ifndef NHDFIT_FIT_TRACK_SKIP
#define NHDFIT_FIT_TRACK_SKIP 1     // once every pod of the tile has its winners of a wavefront's run, the winner test of a chunk is one scalar branch
#endif
#ifndef NHDFIT_FIT_CROW
#define NHDFIT_FIT_CROW 1           // the pipelined loop takes the C row's address from the record (NodeRec::flags); a launch whose records were written
                                    // for another pair-table dimension (FitArgs::crow_ok = 0) sweeps six rows instead.  NOT a further instantiation of the
                                    // loop: with four pipelined loops instead of two the register allocation of k_step spilled in the three-group tiles' loop
#endif
#ifndef NHDFIT_FIT_SWP
#define NHDFIT_FIT_SWP 3            // bit 0: software-pipelined chunk loop for the W = 2 pair form, bit 1: for W = 4 (A/B builds: -DNHDFIT_FIT_SWP=n;
                                    // profiles/r06/fit_swp_ab.log: steady step 14.94 -> 14.34 us with both, 14.59 with the first alone)
#endif       // one block of the fit role: chunks [c_begin, c_end) of a tile
#ifdef NHDFIT_TUNING
constexpr bool kTuning = true;        // ablation switches (FitArgs::dbg_skip) are compiled into the tuning build only
#else
constexpr bool kTuning = false;
#endif

struct FitArgs {
    const NodeRec* rec[kWClasses];   // node records per row width (k_xrecords), padded to a multiple of 64 nodes
    const double* bt[kWClasses];     // busy times in the records' lane order (k_xrecords / k_xorder), padded likewise
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
    // pair form of the sweep (fit_core.h "pair rows"), per row width: pair_D = 0: off; else C[2][D][D] is derived in LDS behind
    // the winner scratch (W / 2 planes of 16-byte pieces).  Set for W = 2 and 4 only.
    uint32_t pair_D[2];
    uint32_t hot_wc1[2];             // where the second socket's CPU records start in the hot section
    uint32_t crow_ok[2];             // the records' flags hold the C row for exactly pair_D (NodeRec::flags): the pipelined sweep takes the address from
                                     // there; 0 (a small batch behind a larger one of another dimension): that width sweeps six rows this launch
    uint32_t fc_dim;
    uint32_t dbg_skip;          // tuning aid (NHDFIT_FIT_SKIP): 1 no table sweep, 2 no winner tracking, 4 constant record, 8 no predicate rows
    unsigned long long* clk;    // tuning aid (NHDFIT_FIT_PHASES): per fit block, summed and latest - [0/1] staged, [2/3] pair table derived, [4/5] sweep done,
                                // [6/7] scores out (10 ns ticks after the block's start), [8] blocks
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

// Pair form (fit_core.h "pair rows"): one C row instead of four CPU rows.  a_c = address of the row's piece 0; piece q lies one
// plane (`plane` bytes) further.
template <int W>
__device__ __forceinline__ uint64_t sweep_pair_c(const uint8_t* lds, uint32_t a_c, uint32_t plane, uint32_t a_x0, uint32_t a_x1) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int q = 0; q < W / 2; ++q) {
        const uint32_t o = q * 16;
        const uint4 cp = lds16(lds, a_c + q * plane), x0 = lds16(lds, a_x0 + o), x1 = lds16(lds, a_x1 + o);
        lo |= __builtin_amdgcn_bitop3_b32(cp.x, x0.x, x1.x, 0x80) | __builtin_amdgcn_bitop3_b32(cp.z, x0.z, x1.z, 0x80);
        hi |= __builtin_amdgcn_bitop3_b32(cp.y, x0.y, x1.y, 0x80) | __builtin_amdgcn_bitop3_b32(cp.w, x0.w, x1.w, 0x80);
    }
    return ((uint64_t)hi << 32) | lo;
}

// The verdict matrix is written once per step and never read back by the launch: NHDFIT_NM_NT = 1 marks the store non-temporal
// (streaming: 33.5 MB per step that need not displace the node records in L2 nor wait for the launch's end to leave it).
#ifndef NHDFIT_NM_NT
#define NHDFIT_NM_NT 1
#endif
__device__ __forceinline__ void verdict_store(uint64_t* p, uint64_t v) {
#if NHDFIT_NM_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// Address of a node's C row (piece 0): row ((smt * D + min(c0, D - 1)) * D + min(c1, D - 1)) of 16 bytes behind off_c.  Every factor
// is below 2^24: v_min_u32 and v_mad_u32_u24 (full rate) instead of compare + select and the quarter-rate 64-bit multiply-add the
// plain expression compiles to (round 6).
__device__ __forceinline__ uint32_t pair_row_addr(uint32_t rec_w, uint32_t pD, uint32_t off_c) {
    const uint32_t cw = rec_w >> 16, c0 = cw & 127u, c1 = (cw >> 7) & 127u, top = pD - 1u;
    const uint32_t r0 = ((0u - (cw >> 14)) & pD) + (c0 < top ? c0 : top);        // (smt is 0 or 1)
    return off_c + ((__umul24(r0, pD) + (c1 < top ? c1 : top)) << 4);
}

// One chunk's rows in flight for the software-pipelined sweep (pair form): everything the verdict word of a node needs from LDS.
template <int W>
struct PairRows { uint4 cp[W / 2], x0[W / 2], x1[W / 2]; uint2 gx, hp; };
template <int W>
__device__ __forceinline__ PairRows<W> fetch_pair_rows(const uint8_t* lds, const uint4 rv, uint32_t pD, uint32_t off_c, uint32_t c_plane,
                                                       uint32_t hot_hp, uint32_t hp_last) {
    PairRows<W> r;
    const uint32_t a_x0 = (rv.y & 0xFFFFu) << 3, a_x1 = (rv.y >> 16) << 3, a_gx = (rv.z & 0xFFFFu) << 3;
    const uint32_t hp = (rv.z >> 16) & 1023u;
    const uint32_t a_hp = hot_hp + (hp < hp_last ? hp : hp_last) * 8;
    const uint32_t a_c = NHDFIT_FIT_CROW ? off_c + ((rv.w & 0xFFFEu) << 3) : pair_row_addr(rv.w, pD, off_c);   // (row << 1 in the flags half, 16 bytes a row)
#pragma unroll
    for (int q = 0; q < W / 2; ++q) {
        r.cp[q] = lds16(lds, a_c + q * c_plane);
        r.x0[q] = lds16(lds, a_x0 + q * 16);
        r.x1[q] = lds16(lds, a_x1 + q * 16);
    }
    r.gx = lds8(lds, a_gx);
    r.hp = lds8(lds, a_hp);
    return r;
}
template <int W>
__device__ __forceinline__ uint64_t pair_rows_word(const PairRows<W>& r) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int q = 0; q < W / 2; ++q) {
        lo |= __builtin_amdgcn_bitop3_b32(r.cp[q].x, r.x0[q].x, r.x1[q].x, 0x80) | __builtin_amdgcn_bitop3_b32(r.cp[q].z, r.x0[q].z, r.x1[q].z, 0x80);
        hi |= __builtin_amdgcn_bitop3_b32(r.cp[q].y, r.x0[q].y, r.x1[q].y, 0x80) | __builtin_amdgcn_bitop3_b32(r.cp[q].w, r.x0[q].w, r.x1[q].w, 0x80);
    }
    return ((uint64_t)(hi & r.gx.y & r.hp.y) << 32) | (lo & r.gx.x & r.hp.x);
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
// PAIR: 0 = six row fetches per pair of assignments; 1 = C tabulated in LDS (three).
// The lanes of a chunk work in the records' order (NodeRec::hp names the node's position in the chunk, fit_core.h "lane order").
// SWP (pair form only): the chunk loop software-pipelined - the NEXT chunk's row fetches are in the LDS queue while this chunk's words are
// combined, stored and tracked, the record after that is on its way from L2 (round 6; the form of round 2 was bound by an LDS pipe that
// spent half of its cycles on bank conflicts - they are gone since round 5).
template <int BLOCK, int W, bool SPILL, int PAIR = 0, bool SWP = false>
__device__ __forceinline__ void role_fit_w(const FitArgs& a, const double busy_from, const FitItem it, uint8_t* lds) {
    static_assert(PAIR == 0 || (!SPILL && W <= 4), "pair tables: narrow tiles, whole hot section staged");
    static_assert(!SWP || PAIR == 1, "the pipelined loop is the pair form's");
    constexpr int NW = BLOCK / 64;
    const uint32_t dbg = kTuning ? a.dbg_skip : 0u;
    const unsigned long long t_blk = kTuning && a.clk ? (unsigned long long)wall_clock64() : 0ull;
    auto phase = [&](int k) {                                      // tuning aid (FitArgs::clk): time since the block entered the role, summed and latest
        if (kTuning && a.clk && threadIdx.x == 0) {
            const unsigned long long d = (unsigned long long)wall_clock64() - t_blk;
            atomicAdd(&a.clk[2 * k], d);
            atomicMax(&a.clk[2 * k + 1], d);
        }
    };
    // the argument block may live behind a pointer (k_step_p): what the chunk loop uses is read once, here
    const uint64_t* __restrict__ cand = a.cand;
    uint64_t* __restrict__ nm = a.nm;
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
    const double* __restrict__ bts = a.bt[WC];
    const uint32_t len = it.c_end - it.c_begin, per = (len + NW - 1) / NW;
    const uint32_t c_first = it.c_begin + wave * per;
    const uint32_t c_last = (dbg & 64) ? c_first : c_first + per < it.c_end ? c_first + per : it.c_end;

    // Everything the block needs first is requested before anything is waited for: the tile's request headers, the
    // wavefront's first node records and the hot section of the table image are independent L2 round trips - issued one
    // after the other behind a barrier they would add up.
    const PodHeader my_h = a.hdr[pod0 + lane];
    uint4 rv = make_uint4(0u, 0u, 0u, 0u);
    double bt = 0.0;
    if (c_first < c_last) {
        rv = *reinterpret_cast<const uint4*>(recs + c_first * 64 + lane);           // {w0,w1}, {x0,x1}, {gx,hp}, {flags,pad}
        bt = bts[c_first * 64 + lane];
    }
    const uint8_t* hot_global = a.tabs + (size_t)tile * a.pitch + a.off_hot[WC];
    {   // stage the hot section of the tile's table image in LDS (16 B per lane, fully coalesced)
        const uint4* src = reinterpret_cast<const uint4*>(hot_global);
        uint4* dst = reinterpret_cast<uint4*>(hot);
        // (a plain loop: load - wait - store per 8 KB.  Four loads in flight per thread were measured in round 6: staged 2.2 -> 1.8 us per
        //  block, but the step 14.05 -> 14.2-14.3 us - the blocks then reach their pair-table derivation together)
        if (!(dbg & 16)) for (uint32_t i = threadIdx.x; i < staged / 16; i += BLOCK) dst[i] = src[i];
        if (spill) {                                                                 // the HP rows go right behind the prefix
            const uint4* hsrc = reinterpret_cast<const uint4*>(hot_global + a.hot_hp[WC]);
            for (uint32_t i = threadIdx.x; i < a.hp_bytes / 16; i += BLOCK) dst[staged / 16 + i] = hsrc[i];
        }
    }
    __syncthreads();
    phase(0);
    // pair table behind the winner scratch: C[2][D][D] as W / 2 planes of 16-byte pieces (fit_core.h "pair rows").  Derived by every
    // block of the tile for itself: ONE derivation per tile in the digest role with the fit blocks copying the table from L2 was built
    // and measured twice in round 6 (plain copy loop; four loads in flight per thread) - the copy (37 KB per four-assignment block)
    // costs ~4 us from L2 with both launches' blocks on the chip, more than the derivation it replaces: step 14.4 -> 16.4 us
    // (profiles/r06/pair_table_in_digest_*withdrawn.log)
    const uint32_t pD = PAIR ? a.pair_D[WC & 1] : 0u;
    const uint32_t off_c = (uint32_t)lds_slice(hot_bytes) + NW * 64 * (uint32_t)sizeof(unsigned long long), c_plane = 2 * pD * pD * 16;
    if constexpr (PAIR != 0) {
        const uint32_t hot_wc1 = a.hot_wc1[WC & 1], fc_dim = a.fc_dim;
        constexpr uint32_t kWcStride = 2 * W * 8 + 16;                    // wc_stride_of(W)
        for (uint32_t r = threadIdx.x; r < 2 * pD * pD; r += BLOCK) {
            const uint32_t smt = r >= pD * pD ? 1u : 0u, e = r - smt * pD * pD, c0 = e / pD, c1 = e - c0 * pD;
            const uint32_t a_w0 = (smt * fc_dim + c0) * kWcStride, a_w1 = hot_wc1 + (smt * fc_dim + c1) * kWcStride;
#pragma unroll
            for (int q = 0; q < W / 2; ++q) {
                const uint32_t o = q * 16;
                const uint4 w0 = lds16(hot, a_w0 + o), w0m = lds16(hot, a_w0 + W * 8 + o);
                const uint4 w1 = lds16(hot, a_w1 + o), w1m = lds16(hot, a_w1 + W * 8 + o);
                uint4 cp;
                cp.x = __builtin_amdgcn_bitop3_b32(w0.x, w1m.x, w0m.x & w1.x, 0xEA);
                cp.y = __builtin_amdgcn_bitop3_b32(w0.y, w1m.y, w0m.y & w1.y, 0xEA);
                cp.z = __builtin_amdgcn_bitop3_b32(w0.z, w1m.z, w0m.z & w1.z, 0xEA);
                cp.w = __builtin_amdgcn_bitop3_b32(w0.w, w1m.w, w0m.w & w1.w, 0xEA);
                *reinterpret_cast<uint4*>(__builtin_assume_aligned(lds + off_c + q * c_plane + r * 16, 16)) = cp;
            }
        }
        __syncthreads();
    }
    phase(1);

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
    if constexpr (SWP) {
        // Two register sets that swap roles every chunk (the loop is unrolled by two: rotating ONE set through copies would make every
        // iteration wait for the loads it has just issued).  At the top of a step for chunk c: rows(c) are in the LDS queue or back (qa),
        // record(c) is back (ra / ba), record(c + 1) is on its way (rb / bb).  The step requests record(c + 2) into the registers of
        // record(c) - whose address fields were consumed by the row fetch one step ago; position, GPU-less flag and busy time are
        // copied out first -, fetches rows(c + 1), then combines / stores / tracks chunk c.  Past the run's end the last chunk's record
        // is requested again (harmless).
        const uint32_t c_stop = c_last - 1u;
        auto rec_of = [&](uint32_t c) { return *reinterpret_cast<const uint4*>(recs + (c < c_stop ? c : c_stop) * 64 + lane); };
        auto bt_of = [&](uint32_t c) { return bts[(c < c_stop ? c : c_stop) * 64 + lane]; };
        auto step = [&](const uint32_t c, const PairRows<W>& qa, uint4& ra, double& ba, PairRows<W>& qb, const uint4& rb) {
            const uint32_t pos = ra.z >> 26;
            const bool nogpu = (ra.w & kRecNoGpu) != 0;
            const bool busy = ba >= busy_from;                                  // Node.IsBusy, nhd/Node.py:847-850
            ra = rec_of(c + 2);
            ba = bt_of(c + 2);
            qb = fetch_pair_rows<W>(lds, rb, pD, off_c, c_plane, hot_hp, hp_last);
            const uint64_t okm = pair_rows_word<W>(qa);
            uint32_t wlo = (uint32_t)okm, whi = (uint32_t)(okm >> 32);
            if (busy) { wlo &= ~(uint32_t)m_need; whi &= ~(uint32_t)(m_need >> 32); }      // Matcher.py:107-111
            if (cand) {
                const uint64_t cw = cand[c];
                if (!(cw >> pos & 1)) wlo = whi = 0;
            }
            if (nm) verdict_store(&nm[(size_t)tile * npad + c * 64 + pos], ((uint64_t)whi << 32) | wlo);
            const uint32_t nlo = (uint32_t)need_any | (nogpu ? (uint32_t)need_pref : 0u);
            const uint32_t nhi = (uint32_t)(need_any >> 32) | (nogpu ? (uint32_t)(need_pref >> 32) : 0u);
            if (NHDFIT_FIT_TRACK_SKIP && !(need_any | need_pref)) return;      // every pod of the tile has its winner of this run: nothing left to track (wave-uniform)
            if (__ballot(((wlo & nlo) | (whi & nhi)) != 0)) {
                wlo = (uint32_t)__builtin_amdgcn_ds_permute((int)(pos << 2), (int)wlo);
                whi = (uint32_t)__builtin_amdgcn_ds_permute((int)(pos << 2), (int)whi);
                const uint64_t nogpu_mask = __ballot(__builtin_amdgcn_ds_permute((int)(pos << 2), nogpu ? 1 : 0) != 0);
                transpose64(wlo, whi);
                const uint64_t word = ((uint64_t)whi << 32) | wlo;
                const uint64_t pref = my_pod_needs_gpu ? 0ull : word & nogpu_mask;
                if (word && best_any == ~0u) best_any = c * 64 + (uint32_t)__builtin_ctzll(word);
                if (pref && best_pref == ~0u) best_pref = c * 64 + (uint32_t)__builtin_ctzll(pref);
                need_any &= ~__ballot(word != 0);
                need_pref &= ~__ballot(pref != 0);
            }
        };
        if (c_first < c_last) {
            uint4 r0 = rv, r1 = rec_of(c_first + 1);
            double b0 = bt, b1 = bt_of(c_first + 1);
            PairRows<W> q0 = fetch_pair_rows<W>(lds, r0, pD, off_c, c_plane, hot_hp, hp_last), q1;
            uint32_t c = c_first;
            for (; c + 1 < c_last; c += 2) {
                step(c, q0, r0, b0, q1, r1);
                step(c + 1, q1, r1, b1, q0, r0);
            }
            if (c < c_last) step(c, q0, r0, b0, q1, r1);
        }
    } else
    for (uint32_t c = c_first; c < c_last; ++c) {
        const uint32_t i = c * 64 + lane;
        uint4 rv_next = rv;
        double bt_next = bt;
        if (c + 1 < c_last && !(dbg & 4)) {      // (a second record in flight for the narrow tiles: measured in round 5, no gain - profiles/r05)
            rv_next = *reinterpret_cast<const uint4*>(recs + i + 64);
            bt_next = bts[i + 64];
        }
        const uint32_t a_w0 = (rv.x & 0xFFFFu) << 3, a_w1 = (rv.x >> 16) << 3;
        const uint32_t a_x0 = (rv.y & 0xFFFFu) << 3, a_x1 = (rv.y >> 16) << 3;
        const uint32_t a_gx = (rv.z & 0xFFFFu) << 3;
        const uint32_t hp = (rv.z >> 16) & 1023u, pos = rv.z >> 26;       // pos: this lane's node is node c * 64 + pos
        const uint32_t a_hp = hot_hp + (hp < hp_last ? hp : hp_last) * 8;
        const bool nogpu = (rv.w & kRecNoGpu) != 0;

        // (1) NUMA-assignment feasibility against all 64 pods (bit-sliced tables), (2) scalar predicates
        uint64_t okm;
        if (SPILL && spill && __ballot(a_x0 + W * 8 > staged || a_x1 + W * 8 > staged))
            okm = sweep_assignments_spill<W>(hot, hot_global, staged, a_w0, a_w1, a_x0, a_x1);
        else if constexpr (PAIR == 0)
            okm = (dbg & 1) ? ((uint64_t)rv.y << 32 | rv.x) : sweep_assignments<W>(hot, a_w0, a_w1, a_x0, a_x1);
        else {
            okm = sweep_pair_c<W>(lds, pair_row_addr(rv.w, pD, off_c), c_plane, a_x0, a_x1);
        }
        const uint2 gx = (dbg & 8) ? make_uint2(rv.z, rv.w) : lds8(hot, a_gx), hpw = (dbg & 8) ? make_uint2(~0u, ~0u) : lds8(hot, a_hp);
        const bool busy = bt >= busy_from;                                  // Node.IsBusy, nhd/Node.py:847-850
        uint32_t wlo = (uint32_t)okm & gx.x & hpw.x, whi = (uint32_t)(okm >> 32) & gx.y & hpw.y;
        if (busy) { wlo &= ~(uint32_t)m_need; whi &= ~(uint32_t)(m_need >> 32); }      // Matcher.py:107-111
        if (cand) {                                                           // candidate dict of the call (FindNode's nl)
            const uint64_t cw = cand[c];
            if (!(cw >> pos & 1)) wlo = whi = 0;
        }
        if (nm) verdict_store(&nm[(size_t)tile * npad + c * 64 + pos], ((uint64_t)whi << 32) | wlo);      // (the chunk's 512 bytes, whatever the lane order)

        // (3) does this chunk change any pod's winner?
        const uint32_t nlo = (uint32_t)need_any | (nogpu ? (uint32_t)need_pref : 0u);
        const uint32_t nhi = (uint32_t)(need_any >> 32) | (nogpu ? (uint32_t)(need_pref >> 32) : 0u);
        if (!(dbg & 2) && (!NHDFIT_FIT_TRACK_SKIP || (need_any | need_pref)) && __ballot(((wlo & nlo) | (whi & nhi)) != 0)) {
            // back to node order first (rare path): lane p receives the words of the lane that worked on node c * 64 + p
            wlo = (uint32_t)__builtin_amdgcn_ds_permute((int)(pos << 2), (int)wlo);
            whi = (uint32_t)__builtin_amdgcn_ds_permute((int)(pos << 2), (int)whi);
            const uint64_t nogpu_mask = __ballot(__builtin_amdgcn_ds_permute((int)(pos << 2), nogpu ? 1 : 0) != 0);
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
    phase(2);
    unsigned long long best = 0;
    if (best_pref != ~0u) best = score_of(true, a.global_base + best_pref);
    else if (best_any != ~0u) best = score_of(false, a.global_base + best_any);
    if (dbg & 32) return;
    s_best[wave][lane] = best;
    __syncthreads();
    if (wave == 0 && my_pod_live) {
        unsigned long long m = s_best[0][lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) m = s_best[w][lane] > m ? s_best[w][lane] : m;
        if (m) atomicMax(&a.score[pod0 + lane], m);
    }
    phase(3);
    if (kTuning && a.clk && threadIdx.x == 0) atomicAdd(&a.clk[8], 1ull);
}

// (the same for a work item computed by the caller: the single-launch find, k_find)
template <int BLOCK, bool SPILL = false>
__device__ __forceinline__ void role_fit_item(const FitArgs& a, const double busy_from, FitItem it, uint8_t* lds) {
    it.tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.tile);      // block-uniform: keep it in scalar registers
    it.wcls = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.wcls);
    it.c_begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.c_begin);
    it.c_end = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.c_end);
    switch (it.wcls) {
        case 0: role_fit_w<BLOCK, 2, SPILL>(a, busy_from, it, lds); break;
        case 1: role_fit_w<BLOCK, 4, SPILL>(a, busy_from, it, lds); break;
        case 2: role_fit_w<BLOCK, 8, SPILL>(a, busy_from, it, lds); break;
        default: role_fit_w<BLOCK, 16, SPILL>(a, busy_from, it, lds); break;
    }
}
template <int BLOCK, bool SPILL = false>
__device__ __forceinline__ void role_fit(const FitArgs& a, const double busy_from, uint32_t blk, uint8_t* lds) {
    FitItem it = a.items[blk];                      // block-uniform: keep it in scalar registers
    it.tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.tile);
    it.wcls = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.wcls);
    it.c_begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.c_begin);
    it.c_end = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.c_end);
    if constexpr (!SPILL) {                          // narrow tiles whose pair table fits the launch's LDS (refresh_layouts)
        if (it.wcls <= 1 && a.pair_D[it.wcls] && (!NHDFIT_FIT_CROW || !((NHDFIT_FIT_SWP >> it.wcls) & 1) || a.crow_ok[it.wcls])) {
            if (it.wcls == 0) role_fit_w<BLOCK, 2, false, 1, (NHDFIT_FIT_SWP & 1) != 0>(a, busy_from, it, lds);
            else role_fit_w<BLOCK, 4, false, 1, (NHDFIT_FIT_SWP & 2) != 0>(a, busy_from, it, lds);
            return;
        }
    }
    switch (it.wcls) {
        case 0: role_fit_w<BLOCK, 2, SPILL>(a, busy_from, it, lds); break;
        case 1: role_fit_w<BLOCK, 4, SPILL>(a, busy_from, it, lds); break;
        case 2: role_fit_w<BLOCK, 8, SPILL>(a, busy_from, it, lds); break;
        default: role_fit_w<BLOCK, 16, SPILL>(a, busy_from, it, lds); break;
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
