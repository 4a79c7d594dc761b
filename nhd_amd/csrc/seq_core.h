// seq_core.h - mode B: sequential-commit semantics of a batch.
//
// The real scheduler commits pod k's winner before it matches pod k+1 (nhd/NHDScheduler.py:289-304, 425-437).
// A commit only ever takes resources away from ONE node, so it can only turn that node's column of the snapshot
// feasibility matrix from 1 to 0 for later pods.  The batch is therefore decided in one pass over the pods, in the
// caller's order, on top of the snapshot the fit role produced:
//   1. pod k's node = first set bit of its (up-to-date) row, with SelectNode's GPU-less preference (Matcher.py:393-421)
//   2. its mapping = FindNode's winner-only tail (Matcher.py:337-452) against the node's CURRENT packed state
//   3. the commit on the packed state (commit_core.h): physical core / GPU ids out, bitmaps, GPU mask, NIC classes,
//      hugepages, busy time and NIC signatures updated
//   4. the node's column is re-evaluated for every tile from the COLD table rows (fit_core.h node_word_cold) and the
//      rows of the pods that lost the node are patched.
// Shared by the device kernel (nhdfit.hip k_seq) and the host twin of the tests.
#pragma once
#include "commit_core.h"
#include "set_states.h"

namespace nhdfit {

struct SeqResult {
    int64_t node;                // global node index or -1
    nhdfit_mapping map;
    int32_t status;              // NHDFIT_COMMIT_*
};

struct MapTables {               // optional accelerators of the order-dependent core (null: run the set model)
    const AscEntry* asc;
    const uint8_t* choose_tab;
    SetStates st;
};

NHD_HD WinnerState state_view(const NodeState& s, const nhdfit_detail& d, const double* caps) {
    WinnerState w;
    w.d = &d;
    w.U = d.numa_nodes;
    w.smt = (s.p2.flags & NHDFIT_NF_SMT) != 0;
    w.free_c[0] = popc64(s.p0.t0[0] & s.p1.t1[0]);
    w.free_c[1] = popc64(s.p0.t0[1] & s.p1.t1[1]);
    w.free_g[0] = popc32(s.p2.gpu_free & ~s.p2.gpu_numa1);
    w.free_g[1] = popc32(s.p2.gpu_free & s.p2.gpu_numa1);
    w.caps = caps;
    return w;
}

// The mapping FindNode returns for pod `r` on a node in state (s, d).  nic_bits: NIC-feasible assignments of the pair
// (bit p), read from the cold R rows of the pod's tile image for the node's current signatures.
NHD_HD bool map_on_state(const nhdfit_req& r, const NodeState& s, const nhdfit_detail& d, const double* caps, uint32_t nic_bits,
                         const MapTables& t, nhdfit_mapping& m) {
    const WinnerState w = state_view(s, d, caps);
    const int G = (int)r.n_groups, U = w.U;
    m.valid = 0;
    const uint32_t codes = nic_codes_from_table_bits(nic_bits, G, U);
    if (G > 3) return map_winner_t<GenericOps>(r, w, codes, m);
    uint32_t sg, sc;
    candidate_masks(r, w, sg, sc);
    const uint32_t nG = ipow(U, G);
    const uint32_t cd = codes & ((1u << nG) - 1u);
    if (!sg || !sc || !cd) return false;
    uint32_t res;
    if (t.choose_tab && choose_tabulated(G, U)) res = choose_from_table(t.choose_tab, G, sg, sc, cd);
    else if (t.st.info && G == 3 && U == 2) res = choose_g3(t.st, t.asc, sg, sc, cd);
    else {
        uint32_t gcode = 0;
        int ccode = -1;
        const bool ok = choose_tuples<SmallOps>(G, U, sg, sc, cd, gcode, ccode, t.asc);
        res = choose_result_word(ok, gcode, ccode);
    }
    if (!(res >> 8 & 1)) return false;
    return finish_mapping(r, w, (res >> 4) & 7u, (int)(res & 15u), m);
}

}  // namespace nhdfit
