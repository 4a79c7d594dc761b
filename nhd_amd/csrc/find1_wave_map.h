// find1_wave_map.h - the mapping tail of the one-pod launch (k_find1, step_kernel.h) on ONE wavefront whose lanes work together.
// Round 4 ran map_one_tile<BLOCK, true> there - the tile machinery with a single live lane walking candidate_masks and the NIC
// choices one after the other, 12-20 us of the 35 us call.  The sequential kernels already own the wave-cooperative form of the
// same arithmetic (seq_kernel.h map_on_state_wave: a lane per tuple code / NIC choice); for a lone pod its inputs are at hand - the
// winner's planes and detail, the capacity classes, and the NIC-feasible assignments from the pod's own masks (fit_core.h
// lone_nic_bits) instead of a tile image.  Measured in round 5's first GPU call (profiles/r05/candidates.md): per nhdfit_find call
// with one pod 27.9 -> 24.8 us at 4 096 nodes, 30.1 -> 27.3 at 16 384, 37.1 -> 34.3 at 65 536, 35.4 -> 30.7 on the config-5 shard,
// single-launch and lone-pod parity tests green - adopted.  tests/test_wave_commit_emulation.py
// (test_lone_pod_winner_mapped_by_the_wavefront_form) runs exactly this composition on emulated lanes against the table pass's mapping.
// map_on_state_wave (seq_kernel.h) for a pod of at most three groups: the same steps without the call into the generic set model
// (four groups), whose scratch arrays would size the private segment of every k_find1 launch (10 KB per lane)
__device__ __forceinline__ bool map_on_state_wave_small(const nhdfit_req& r, const NodeState& s, const nhdfit_detail& d, const double* caps, uint32_t nic_bits,
                                                        const MapTables& t, uint32_t lane, nhdfit_mapping& m) {
    const WinnerState w = state_view(s, d, caps);
    const int G = (int)r.n_groups, U = w.U;
    m = nhdfit_mapping{};
    if (G > 3) return false;
    const uint32_t codes = nic_codes_from_table_bits(nic_bits, G, U);
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
    uint32_t nic_nibbles = 0;
    const bool nic_ok = first_nic_choice_wave(r, w, gcode, r.map_type == NHDFIT_MAP_PCI, lane, nic_nibbles);
#pragma unroll
    for (int g = 0; g < kMaxG; ++g) {
        const bool in = g < G;
        m.gpu[g] = in ? (int8_t)tup_digit(gcode, G, U, g) : (int8_t)-1;
        m.nic_numa[g] = m.gpu[g];
        m.nic_idx[g] = in ? (int8_t)nib_get(nic_nibbles, g) : (int8_t)-1;
    }
#pragma unroll
    for (int g = 0; g <= kMaxG; ++g) m.cpu[g] = g <= G ? (int8_t)tup_digit((uint32_t)ccode, G + 1, U, g) : (int8_t)-1;
    m.valid = nic_ok ? 1 : 0;
    if (!nic_ok) m = nhdfit_mapping{};
    return nic_ok;
}
__device__ __forceinline__ void map_lone_pod_wave(const MapArgs& a, const ShapeArgs& h, const LoneMasks& t, const nhdfit_req& r,
                                                  const double* __restrict__ caps, uint8_t* lds) {
    NodeState* st = carve<NodeState>(lds, 1);
    nhdfit_detail* dd = carve<nhdfit_detail>(lds, 1);
    double* l_caps = carve<double>(lds, NHDFIT_MAX_CLASSES);
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    if (tid < 64u) {                                                      // wavefront 0; the others wait at the barrier below
        nhdfit_mapping mp = nhdfit_mapping{};
        const unsigned long long s = __hip_atomic_load(a.score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t gi = s ? NHDFIT_SCORE_INDEX(s) : ~0ull;
        if (s && gi >= a.global_base && gi < a.global_base + a.n && r.n_groups <= 3u) {      // (wave-uniform) this shard's node
            const uint32_t i = (uint32_t)(gi - a.global_base);
            uint32_t* sw = reinterpret_cast<uint32_t*>(st);
            if (lane < 4u) {                                             // the winner's planes: a 16-byte load per lane
                const uint4 q = lane == 0u ? *reinterpret_cast<const uint4*>(a.p0 + i) : lane == 1u ? *reinterpret_cast<const uint4*>(a.p1 + i) :
                                lane == 2u ? *reinterpret_cast<const uint4*>(a.p2 + i) : *reinterpret_cast<const uint4*>(a.p3 + i);
                sw[lane * 4 + 0] = q.x; sw[lane * 4 + 1] = q.y; sw[lane * 4 + 2] = q.z; sw[lane * 4 + 3] = q.w;
            }
            if (lane == 4u) { sw[16] = 0u; sw[17] = 0u; sw[18] = 0u; sw[19] = 0u; }         // plane 4 (busy time, group set): not read by the mapping
            if (lane >= 8u && lane < 16u) {
                const uint4 q = reinterpret_cast<const uint4*>(a.det + i)[lane - 8u];
                uint32_t* dw = reinterpret_cast<uint32_t*>(dd) + (lane - 8u) * 4u;
                dw[0] = q.x; dw[1] = q.y; dw[2] = q.z; dw[3] = q.w;
            }
            if (lane >= 16u && lane < 16u + (uint32_t)NHDFIT_MAX_CLASSES) l_caps[lane - 16u] = caps[lane - 16u];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t bits = lone_nic_bits(t, r.map_type == NHDFIT_MAP_PCI, st->p3);
            const MapTables mt{h.asc, h.choose_tab, h.st};
            map_on_state_wave_small(r, *st, *dd, l_caps, bits, mt, lane, mp);   // every lane: the same mapping (zeros when nothing fits)
        }
        if (lane == 0u && r.n_groups <= 3u) a.out[0] = mp;                // (a pod with four groups is mapped by k_map<true>, as before)
    }
    __syncthreads();
}
