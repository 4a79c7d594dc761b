// probe_wave.hip - NOT part of the product: the wavefront forms of the commit step and of the mapping on a node state (what the
// chain of mode B's GPU-less pods executes pod after pod) as stand-alone kernels, so that their ISA can be read and counted without a GPU
// (tools/probe_wave_isa.sh: hipcc --cuda-device-only -S; instruction histogram, LDS reads followed by a full wait).
#include "../nhd_amd/csrc/nhdfit.hip"
namespace {
__global__ __launch_bounds__(256) void probe_lone_map(MapArgs m, ShapeArgs h, LoneMasks t, const nhdfit_req* r, const double* caps) {
    extern __shared__ __align__(16) uint8_t lds_probe[];
    __shared__ nhdfit_req lr;
    if (threadIdx.x < sizeof(nhdfit_req) / 16) reinterpret_cast<uint4*>(&lr)[threadIdx.x] = reinterpret_cast<const uint4*>(r)[threadIdx.x];
    __syncthreads();
    map_lone_pod_wave(m, h, t, lr, caps, lds_probe);
}
__global__ __launch_bounds__(256) void probe_lone_map_shipped(MapArgs m, ShapeArgs h, LoneMasks t, const nhdfit_req* r) {
    extern __shared__ __align__(16) uint8_t lds_probe2[];
    __shared__ nhdfit_req lr;
    if (threadIdx.x < sizeof(nhdfit_req) / 16) reinterpret_cast<uint4*>(&lr)[threadIdx.x] = reinterpret_cast<const uint4*>(r)[threadIdx.x];
    __syncthreads();
    m.reqs = &lr;
    map_one_tile<256, true>(m, h, 0, lds_probe2, &t);
}
__global__ __launch_bounds__(64) void probe_summary(NodeState* s, nhdfit_detail* d, const nhdfit_req* r, const nhdfit_mapping* m, double bt,
                                                    SigTable sigs, uint32_t ncls, nhdfit_placement* out, int* st) {
    // the two-stage commit of k_decide's speculators (round 5): stage 1 on the chain, stage 2 behind it
    __shared__ NodeState ls; __shared__ nhdfit_detail ld; __shared__ nhdfit_placement lo; __shared__ PaddedReq lr;
    const uint32_t lane = threadIdx.x;
    if (lane == 0) { ls = *s; ld = *d; }
    if (lane < sizeof(nhdfit_req) / 16) reinterpret_cast<uint4*>(&lr)[lane] = reinterpret_cast<const uint4*>(r)[lane];
    __syncthreads();
    const nhdfit_mapping lm = *m;
    uint64_t f0, f1, c0, c1;
    int status = commit_summary_wave(ls, ld, reinterpret_cast<const nhdfit_req&>(lr), lm, bt, sigs, ncls, lane, f0, f1);
    asm volatile("; ---- end of stage 1");
    const int s2 = commit_picks_wave(f0, f1, (ls.p2.flags & NHDFIT_NF_SMT) != 0, reinterpret_cast<const nhdfit_req&>(lr), lm, lo, lane, c0, c1);
    if (s2 == kCommitWouldRaise) status = s2;
    __syncthreads();
    if (lane == 0) { ls.p1.t1[0] &= ~c0; ls.p1.t1[1] &= ~c1; *s = ls; *d = ld; *out = lo; *st = status; }
}
__global__ __launch_bounds__(64) void probe_commit(NodeState* s, nhdfit_detail* d, const nhdfit_req* r, const nhdfit_mapping* m, double bt,
                                                   SigTable sigs, uint32_t ncls, nhdfit_placement* out, int* st) {
    __shared__ NodeState ls; __shared__ nhdfit_detail ld; __shared__ nhdfit_placement lo; __shared__ PaddedReq lr; __shared__ nhdfit_mapping lm;
    const uint32_t lane = threadIdx.x;
    if (lane == 0) { ls = *s; ld = *d; lm = *m; }
    if (lane < sizeof(nhdfit_req) / 16) reinterpret_cast<uint4*>(&lr)[lane] = reinterpret_cast<const uint4*>(r)[lane];
    __syncthreads();
    const int status = commit_node_wave(ls, ld, reinterpret_cast<const nhdfit_req&>(lr), lm, bt, sigs, ncls, lo, lane);
    __syncthreads();
    if (lane == 0) { *s = ls; *d = ld; *out = lo; *st = status; }
}
__global__ __launch_bounds__(64) void probe_map(const NodeState* s, const nhdfit_detail* d, const nhdfit_req* r, const double* caps, uint32_t bits, MapTables mt,
                                                nhdfit_mapping* out, int* ok) {
    __shared__ NodeState ls; __shared__ nhdfit_detail ld; __shared__ PaddedReq lr; __shared__ double lc[NHDFIT_MAX_CLASSES];
    const uint32_t lane = threadIdx.x;
    if (lane == 0) { ls = *s; ld = *d; }
    if (lane < NHDFIT_MAX_CLASSES) lc[lane] = caps[lane];
    if (lane < sizeof(nhdfit_req) / 16) reinterpret_cast<uint4*>(&lr)[lane] = reinterpret_cast<const uint4*>(r)[lane];
    __syncthreads();
    nhdfit_mapping mp;
    const bool k = map_on_state_wave(reinterpret_cast<const nhdfit_req&>(lr), ls, ld, lc, bits, mt, lane, mp);
    if (lane == 0) { *out = mp; *ok = k; }
}
}
