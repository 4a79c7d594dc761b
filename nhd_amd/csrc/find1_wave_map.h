// find1_wave_map.h - CANDIDATE for the next GPU measurement, NOT part of libnhdfit.so yet (nothing includes it but
// tools/probe_wave.hip).  The mapping tail of the one-pod launch (k_find1, step_kernel.h) on ONE wavefront whose lanes work
// together: today the block with the last ticket runs map_one_tile<BLOCK, true> - the tile machinery with a single live
// lane walking candidate_masks and the NIC choices one after the other, 12-20 us of the 35 us call (DESIGN.md section 4
// "One pod, one launch").  The sequential kernels already own the wave-cooperative form of the same arithmetic
// (seq_kernel.h map_on_state_wave: a lane per tuple code / NIC choice); for a lone pod its inputs are at hand - the winner's
// planes and detail, the capacity classes, and the NIC-feasible assignments from the pod's own masks (fit_core.h
// lone_nic_bits) instead of a tile image.  tests/test_wave_commit_emulation.py (test_lone_pod_winner_mapped_by_the_wavefront_form)
// runs exactly this composition on emulated lanes against the table pass's mapping.
// To try it on the device: include this file behind seq_kernel.h (it needs map_on_state_wave), replace k_find1's
//     map_one_tile<BLOCK, true>(m, a.h, 0, lds_map, &t);
// by  map_lone_pod_wave(m, a.h, t, *s_req, a.d.caps, lds_map);
// run tests/test_gpu_parity.py -k "single or lone or find" and tools/time_single_find.py.
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
            map_on_state_wave(r, *st, *dd, l_caps, bits, mt, lane, mp);   // every lane: the same mapping (zeros when nothing fits)
        }
        if (lane == 0u && r.n_groups <= 3u) a.out[0] = mp;                // (a pod with four groups is mapped by k_map<true>, as before)
    }
    __syncthreads();
}
