// seq2_commit_v2.h - CANDIDATE for the next GPU measurement, compiled OUT of libnhdfit.so (seq2_kernel.h includes it only under
// -DNHDFIT_CAND_COMMIT_V2, which nhd_amd/build.py does not pass; tools/r05_candidates.sh builds and measures it).  The wavefront form of the commit step (seq2_kernel.h
// commit_node_wave) with the request read ONCE: tools/probe_wave_isa.sh shows the shipped form fetching the request's byte
// fields one ds_read_u8 + s_waitcnt lgkmcnt(0) at a time - five or six LDS round trips per processing group on the chain
// of mode B's GPU-less pods.  Here every lane reads one dword of the 128-byte record, the eight dwords that hold the
// commit's fields are broadcast with v_readlane (wave-uniform values in scalar registers), and a group's counts are
// shifts of those words; the NIC's switch is only looked up for a group that asks for GPUs.  Same arithmetic, same
// results: tests/test_wave_commit_emulation.py runs this text on emulated lanes against the scalar commit_node and
// against the shipped wavefront form.
// Needs lowest_bits_wave / take_batch_wave / sig_keys_wave of seq2_kernel.h in front of it.
struct ReqWords {                   // the commit's fields of a nhdfit_req, wave-uniform
    uint32_t G, map_type, np4, nh4, n_misc, smt_bits, misc_smt_enabled, nic_use;
    int32_t hp;
    uint64_t gp;                    // gpus[g] = 16 bits each
};
__device__ __forceinline__ ReqWords req_words_wave(const nhdfit_req& r, uint32_t lane) {
    static_assert(sizeof(nhdfit_req) == 128 && offsetof(nhdfit_req, n_groups) == 0 && offsetof(nhdfit_req, map_type) == 4 &&
                  offsetof(nhdfit_req, hugepages_gb) == 8 && offsetof(nhdfit_req, gpus) == 24 && offsetof(nhdfit_req, n_proc) == 52 &&
                  offsetof(nhdfit_req, n_help) == 120 && offsetof(nhdfit_req, n_misc) == 124 && offsetof(nhdfit_req, smt_bits) == 125 &&
                  offsetof(nhdfit_req, misc_smt_enabled) == 126 && offsetof(nhdfit_req, nic_use) == 127 && kMaxG == 4,
                  "req_words_wave reads the record by dword");
    const int v = (int)reinterpret_cast<const uint32_t*>(&r)[lane & 31u];       // ONE LDS read per lane, no dependence between them
    ReqWords q;
    q.G = (uint32_t)__builtin_amdgcn_readlane(v, 0);
    q.map_type = (uint32_t)__builtin_amdgcn_readlane(v, 1);
    q.hp = (int32_t)__builtin_amdgcn_readlane(v, 2);
    q.gp = (uint64_t)(uint32_t)__builtin_amdgcn_readlane(v, 6) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(v, 7) << 32);
    q.np4 = (uint32_t)__builtin_amdgcn_readlane(v, 13);
    q.nh4 = (uint32_t)__builtin_amdgcn_readlane(v, 30);
    const uint32_t tail = (uint32_t)__builtin_amdgcn_readlane(v, 31);
    q.n_misc = tail & 255u; q.smt_bits = (tail >> 8) & 255u; q.misc_smt_enabled = (tail >> 16) & 255u; q.nic_use = tail >> 24;
    return q;
}
// `s` / `d` / `out` live in LDS (one copy per wavefront); every lane returns the same status
__device__ __forceinline__ int commit_node_wave_v2(NodeState& s, nhdfit_detail& d, const nhdfit_req& r, const nhdfit_mapping& m, double busy_time,
                                                   const SigTable& sigs, uint32_t ncls, nhdfit_placement& out, uint32_t lane) {
    const ReqWords q = req_words_wave(r, lane);
    const int G = (int)q.G;
    int status = kCommitOk;
    {   // the placement record: zeros, 0xFF for the GPU list and the NUMA entries (bytes 144 .. 180 of the 256)
        const uint32_t o = lane * 4u;
        reinterpret_cast<uint32_t*>(&out)[lane] = o >= 144u && o < 180u ? 0xFFFFFFFFu : o == 180u ? 0x000000FFu : 0u;
    }
    uint64_t t0[2] = {s.p0.t0[0], s.p0.t0[1]}, t1[2] = {s.p1.t1[0], s.p1.t1[1]};
    const bool smt_node = (s.p2.flags & NHDFIT_NF_SMT) != 0;
    uint32_t gpu_free = s.p2.gpu_free;
    const uint32_t gpu_numa1 = s.p2.gpu_numa1;
    const bool any_gpu = q.gp != 0;                                       // (a pod without GPUs never looks at the GPU lists)
    const uint32_t n_gpus = any_gpu ? d.n_gpus : 0u;
    const uint32_t my_gsw = any_gpu && lane < n_gpus && lane < (uint32_t)NHDFIT_MAX_GPUS ? d.gpu_sw[lane & 31u] : 0xFFu;
    uint32_t claimed0 = 0, claimed1 = 0;
    bool gpu_taken = false;
    uint32_t m_gpu = 0, m_nnuma = 0, m_nidx = 0, mu = 0;
#pragma unroll
    for (int g = 0; g < kMaxG; ++g) {
        m_gpu |= ((uint32_t)m.gpu[g] & 1u) << g; m_nnuma |= ((uint32_t)m.nic_numa[g] & 1u) << g; m_nidx |= ((uint32_t)m.nic_idx[g] & 15u) << (4 * g);
    }
#pragma unroll
    for (int g = 0; g <= kMaxG; ++g) if (g == G) mu = (uint32_t)m.cpu[g] & 1u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int g = 0; g < G; ++g) {
        const uint32_t u = (m_gpu >> g) & 1u;
        const WaveBatch pb = take_batch_wave(t0[u], t1[u], smt_node, (q.np4 >> (8 * g)) & 255u, (q.smt_bits >> g & 1) != 0, lane);
        if (!pb.ok) status = kCommitWouldRaise;
        const uint32_t nu = (m_nnuma >> g) & 1u, nk = (m_nidx >> (4 * g)) & 15u;
        const uint32_t want = (uint32_t)(q.gp >> (16 * g)) & 0xFFFFu;
        if (want) {
            const uint32_t sw = d.nic_sw[nu][nk];
            for (uint32_t k = 0; k < want; ++k) {
                const bool mine_free = lane < n_gpus && (gpu_free >> lane & 1);
                uint64_t cand = __ballot(mine_free && my_gsw == sw);          // GetFreePciGpuFromNic, Node.py:648-655
                if (!cand && q.map_type != NHDFIT_MAP_PCI) cand = __ballot(mine_free && (gpu_numa1 >> lane & 1) == u);   // GetNextGpuFree, Node.py:495-500
                if (!cand) { status = kCommitWouldRaise; continue; }
                const uint32_t pick = (uint32_t)__builtin_ctzll(cand);
                gpu_free &= ~(1u << pick);
                gpu_taken = true;
                if (lane == 0) {
                    const uint32_t psw = d.gpu_sw[pick];
                    if (d.sw_free[psw]) d.sw_free[psw]--;
                    if (k < (uint32_t)NHDFIT_PLACEMENT_GPUS) out.gpu[g][k] = (uint8_t)pick;
                }
            }
        }
        const WaveBatch hb = take_batch_wave(t0[u], t1[u], smt_node, (q.nh4 >> (8 * g)) & 255u, (q.smt_bits >> (4 + g) & 1) != 0, lane);
        if (!hb.ok) status = kCommitWouldRaise;
        if (q.nic_use >> g & 1) { if (nu) claimed1 |= 1u << nk; else claimed0 |= 1u << nk; }
        if (lane == 0) {
            out.numa[g] = (int8_t)u;
            out.proc_take[g] = pb.take; out.proc_pair[g] = pb.pair; out.proc_late[g] = pb.late;
            out.help_take[g] = hb.take; out.help_pair[g] = hb.pair; out.help_late[g] = hb.late;
        }
    }
    const WaveBatch mb = take_batch_wave(t0[mu], t1[mu], smt_node, q.n_misc, q.misc_smt_enabled != 0, lane);     // Node.py:799
    if (!mb.ok) status = kCommitWouldRaise;
    if (lane == 0) {
        out.numa[kMaxG] = (int8_t)mu;
        out.misc_take = mb.take; out.misc_pair = mb.pair; out.misc_late = mb.late;
        s.p0.t0[0] = t0[0]; s.p0.t0[1] = t0[1]; s.p1.t1[0] = t1[0]; s.p1.t1[1] = t1[1];
        s.p2.gpu_free = gpu_free;
        if (q.hp > 0) s.p2.hp_free -= q.hp;                                  // Node.py:794-796
        s.p4.busy_time = busy_time;                                          // SetBusy, nhd/Node.py:843-845
        for (uint32_t cl = claimed0 | (claimed1 << 16); cl; cl &= cl - 1u) {  // ClaimPodNICResources (commit_core.h: the same rule; every NIC touched once)
            const uint32_t b = (uint32_t)__builtin_ctz(cl), u = b >> 4, k = b & 15u;
            if (pods_get(d, u, k) != kPodsLost && pods_add(d, u, k, 1) != 0) d.nic_cls[u][k] = 0;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t u = 0; u < 2; ++u) {
        if (!(u ? claimed1 : claimed0) && !gpu_taken) continue;
        uint64_t kn, kp;
        uint32_t idn = 0, idp = 0;
        sig_keys_wave(d, u, lane, ncls, kn, kp);
        if (!sig_lookup(sigs, kn, idn) || !sig_lookup(sigs, kp, idp)) { if (status == kCommitOk) status = kCommitNewSig; }
        if (lane == 0) { s.p3.sig_numa[u] = (uint16_t)idn; s.p3.sig_pci[u] = (uint16_t)idp; }
    }
    if (lane == 0) out.status = (uint8_t)status;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return status;
}
