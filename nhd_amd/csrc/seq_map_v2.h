// seq_map_v2.h - CANDIDATE for the next GPU measurement, compiled OUT of libnhdfit.so (seq2_kernel.h includes it only under
// -DNHDFIT_CAND_MAP_V2, which nhd_amd/build.py does not pass; tools/r05_candidates.sh builds and measures it).  The verification of
// a candidate node on mode B's chain (seq_kernel.h map_on_state_wave) with the NIC walk's wave-uniform operands read ONCE:
// tools/probe_wave_isa.sh shows the shipped first_nic_choice_wave re-reading nic_cnt[] (a byte, seven places) and the groups'
// rx / tx (doubles, inside the per-group loops) from LDS at every use - each a round trip with a full wait on a chain one
// wavefront walks alone.  Here the two NIC counts and the eight bandwidths are loaded up front (one batch of independent reads)
// and picked by the run-time group number with selects.  The per-lane gathers (a lane's NIC class, its capacity, its switch)
// stay what they are.  Same arithmetic, same order of the f64 subtractions: tests/test_wave_commit_emulation.py runs this text on
// emulated lanes against the scalar map_on_state and the shipped wavefront form.
// Needs candidate_masks_wave / choose_model_cold / map_generic_cold of seq_kernel.h in front of it.
__device__ __forceinline__ double sel4_f64(int h, double a0, double a1, double a2, double a3) {
    return h == 0 ? a0 : h == 1 ? a1 : h == 2 ? a2 : a3;
}
__device__ __forceinline__ bool first_nic_choice_wave_v2(const nhdfit_req& r, const WinnerState& w, uint32_t gcode, bool pci, uint32_t lane, uint32_t& nic_nibbles) {
    static_assert(kMaxG == 4, "sel4_f64 picks among four groups");
    const int G = (int)r.n_groups;
    const uint32_t cnt0 = w.d->nic_cnt[0], cnt1 = w.d->nic_cnt[1];
    const double rx0 = r.rx[0], rx1 = r.rx[1], rx2 = r.rx[2], rx3 = r.rx[3];
    const double tx0 = r.tx[0], tx1 = r.tx[1], tx2 = r.tx[2], tx3 = r.tx[3];
    uint32_t order = 0, numa = 0;
    int n = 0;
    for (int u = 0; u < w.U; ++u)
        for (int g = 0; g < G; ++g)
            if (tup_digit(gcode, G, w.U, g) == u) { order = nib_set(order, n, (uint32_t)g); numa |= (uint32_t)u << g; ++n; }
    uint32_t total = 1;
    for (int g = 0; g < G; ++g) {
        const uint32_t k = (numa >> g) & 1 ? cnt1 : cnt0;
        if (k == 0) return false;
        total *= k;
    }
    for (uint32_t base = 0; base < total; base += 64) {
        uint32_t rem = base + lane, pick = 0;
        const bool live = rem < total;
        for (int pos = G - 1; pos >= 0; --pos) {
            const int g = (int)nib_get(order, pos);
            const uint32_t k = (numa >> g) & 1 ? cnt1 : cnt0;
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
                if (((numa >> h) & 1) == u && nib_get(pick, h) == k) { rx = rx - sel4_f64(h, rx0, rx1, rx2, rx3); tx = tx - sel4_f64(h, tx0, tx1, tx2, tx3); }
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
            nic_nibbles = (uint32_t)__builtin_amdgcn_readlane((int)pick, __builtin_ctzll(any));
            return true;
        }
    }
    return false;
}
// map_on_state_wave (seq_kernel.h) with the NIC walk above
__device__ __forceinline__ bool map_on_state_wave_v2(const nhdfit_req& r, const NodeState& s, const nhdfit_detail& d, const double* caps, uint32_t nic_bits,
                                                     const MapTables& t, uint32_t lane, nhdfit_mapping& m) {
    const WinnerState w = state_view(s, d, caps);
    const int G = (int)r.n_groups, U = w.U;
    m = nhdfit_mapping{};
    const uint32_t codes = nic_codes_from_table_bits(nic_bits, G, U);
    if (G > 3) {                                                          // (copies: nothing the hot path keeps in registers has its address taken)
        WinnerState wc = w;
        nhdfit_mapping tmp = nhdfit_mapping{};
        const bool ok = map_generic_cold(&r, &wc, codes, &tmp);
        m = tmp;
        return ok;
    }
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
    const bool nic_ok = first_nic_choice_wave_v2(r, w, gcode, r.map_type == NHDFIT_MAP_PCI, lane, nic_nibbles);
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
