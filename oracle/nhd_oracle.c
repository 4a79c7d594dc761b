/*
 * nhd_oracle.c - plain-C restatement of the reference's node filter-and-score path.
 *
 * *** TEST INFRASTRUCTURE - NOT PRODUCT CODE ***  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.  Nothing under nhd_amd/ does.
 *
 * It evaluates every (pod, node) pair the way the reference does - by explicit enumeration of NUMA
 * assignments and NIC choices over per-core / per-GPU / per-NIC records - and picks the winner.
 * It shares no code and no data layout with the HIP path (which works on bitmaps, interned NIC
 * signatures and per-pod tables).  The winner's resource *mapping* is restated in
 * oracle/nhd_oracle.py only (it depends on CPython set order).
 *
 * Parity pin: tests/test_c_oracle.py checks this file against oracle/nhd_oracle.py (itself pinned
 * to the unmodified reference) and against tests/golden/ (reference outputs).
 *
 * Reference lines followed:
 *   FilterPodResources            nhd/Matcher.py:65-84
 *   GPU / CPU / NIC stages        nhd/Matcher.py:95-149, 152-222, 224-280
 *   PCI pruning, intersection     nhd/Matcher.py:294-335, 337-391
 *   SelectNode                    nhd/Matcher.py:393-421
 *   GetFreeCpuCores               nhd/Node.py:250-264      GetFreeNumaGPUs     nhd/Node.py:456-462
 *   GetFreeNumaNicResources       nhd/Node.py:283-296      GetFreeGPUPCICount  nhd/Node.py:266-273
 *   IsBusy                        nhd/Node.py:847-850      InitialNodeFilter   nhd/NHDScheduler.py:235-247
 *   GetTotal{Gpus,Cpus,NICs}Requested  nhd/CfgTopology.py:199-232
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define MAXG 8
#define MAXU 4
#define MAXNIC 64

typedef struct {
    int32_t numa_nodes, smt, n_scan;      /* n_scan = cores_per_proc * sockets (Node.py:257) */
    int32_t core_off, n_cores;            /* into core_used / core_socket / core_sibling     */
    int32_t gpu_off, n_gpus;              /* into gpu_used / gpu_numa / gpu_sw               */
    int32_t nic_off, n_nics;              /* into nic_numa / nic_speed / nic_pods / nic_sw   */
    int32_t hp_free, maintenance, active;
    uint64_t groups;
    double busy_time;
} onode;

typedef struct {
    int32_t G, map_type, hp, n_misc, misc_smt_truthy, use_filter, pad0, pad1;
    int32_t n_gpus[MAXG], n_proc[MAXG], proc_smt[MAXG], n_help[MAXG], help_smt[MAXG];
    double rx[MAXG], tx[MAXG];
    uint64_t groups;
} opod;

typedef struct {
    const onode* nodes;
    int64_t n;
    const uint8_t* core_used; const int32_t* core_socket; const int32_t* core_sibling;
    const uint8_t* gpu_used;  const int32_t* gpu_numa;    const int32_t* gpu_sw;
    const int32_t* nic_numa;  const double* nic_speed;    const int32_t* nic_pods; const int32_t* nic_sw;
} ocluster;

static int half_up(int n) { return (int)ceil(n / 2.0); }

/* digits of `code` in base U, most significant first (= itertools.product order) */
static void digits(int code, int U, int len, int* out) {
    for (int i = len - 1; i >= 0; --i) { out[i] = code % U; code /= U; }
}

static int ipow(int b, int e) { int r = 1; while (e-- > 0) r *= b; return r; }

int oracle_feasible(const ocluster* c, int64_t idx, const opod* p, double now) {
    const onode* nd = &c->nodes[idx];
    const int U = nd->numa_nodes, G = p->G;
    if (p->map_type != 1 && p->map_type != 2) return 0;                 /* Matcher.py:45-47 */
    if (p->use_filter && !((nd->groups & p->groups) && nd->active)) return 0;
    if (nd->maintenance) return 0;                                      /* Matcher.py:71 */
    if (p->hp > nd->hp_free) return 0;                                  /* Matcher.py:78 */
    if (G < 1 || G > MAXG || U < 1 || U > MAXU) return 0;

    int sum_g = 0;
    for (int g = 0; g < G; ++g) sum_g += p->n_gpus[g];
    if (sum_g > 0 && (now - nd->busy_time) < 30.0) return 0;            /* Matcher.py:107-111 */

    /* free GPUs / cores per NUMA node */
    int free_g[MAXU] = {0}, free_c[MAXU] = {0};
    for (int g = 0; g < nd->n_gpus; ++g)
        if (!c->gpu_used[nd->gpu_off + g]) free_g[c->gpu_numa[nd->gpu_off + g]]++;
    for (int k = 0; k < nd->n_scan; ++k) {
        const int o = nd->core_off + k;
        if (c->core_used[o]) continue;
        if (nd->smt && c->core_used[nd->core_off + c->core_sibling[o]]) continue;
        free_c[c->core_socket[o]]++;
    }
    /* per-group physical core demand (Matcher.py:178-204) */
    int want_c[MAXG + 1];
    for (int g = 0; g < G; ++g) {
        if (nd->smt)
            want_c[g] = (p->proc_smt[g] ? half_up(p->n_proc[g]) : p->n_proc[g]) +
                        (p->help_smt[g] ? half_up(p->n_help[g]) : p->n_help[g]);
        else
            want_c[g] = p->n_proc[g] + p->n_help[g];
    }
    want_c[G] = nd->smt ? (p->misc_smt_truthy ? half_up(p->n_misc) : p->n_misc) : p->n_misc;

    /* NIC capacities per NUMA in node.nics order (Node.py:283-296, sharing disabled) */
    double cap[MAXU][MAXNIC];
    int sw[MAXU][MAXNIC], K[MAXU] = {0};
    for (int k = 0; k < nd->n_nics; ++k) {
        const int o = nd->nic_off + k, u = c->nic_numa[o];
        if (u < 0 || u >= U || K[u] >= MAXNIC) continue;
        cap[u][K[u]] = c->nic_pods[o] > 0 ? 0.0 : c->nic_speed[o] * 0.9;
        sw[u][K[u]] = c->nic_sw[o];
        K[u]++;
    }

    const int nG = ipow(U, G);
    int a[MAXG + 1];
    for (int code = 0; code < nG; ++code) {
        digits(code, U, G, a);
        /* GPU stage for this assignment */
        int tot[MAXU] = {0}, ok = 1;
        for (int g = 0; g < G; ++g) tot[a[g]] += p->n_gpus[g];
        for (int u = 0; u < U; ++u) if (tot[u] > free_g[u]) ok = 0;
        if (!ok) continue;
        /* CPU stage: any NUMA node for the misc cores */
        int cpu_ok = 0;
        for (int m = 0; m < U && !cpu_ok; ++m) {
            int t[MAXU] = {0}, fine = 1;
            for (int g = 0; g < G; ++g) t[a[g]] += want_c[g];
            t[m] += want_c[G];
            for (int u = 0; u < U; ++u) if (t[u] > free_c[u]) fine = 0;
            cpu_ok = fine;
        }
        if (!cpu_ok) continue;
        /* NIC stage: every choice of one NIC per group on its NUMA node */
        int pick[MAXG] = {0}, has = 1;
        for (int g = 0; g < G; ++g) if (K[a[g]] == 0) has = 0;
        if (!has) continue;
        for (;;) {
            double rx[MAXU][MAXNIC], tx[MAXU][MAXNIC];
            for (int u = 0; u < U; ++u) for (int k = 0; k < K[u]; ++k) rx[u][k] = tx[u][k] = cap[u][k];
            for (int g = 0; g < G; ++g) {
                rx[a[g]][pick[g]] -= p->rx[g];
                tx[a[g]][pick[g]] -= p->tx[g];
            }
            int fine = 1;
            for (int u = 0; u < U && fine; ++u)
                for (int k = 0; k < K[u]; ++k) if (rx[u][k] < 0 || tx[u][k] < 0) { fine = 0; break; }
            if (fine && p->map_type == 2) {
                /* groups per PCIe switch <= free GPUs on that switch (Matcher.py:312-322) */
                for (int g = 0; g < G && fine; ++g) {
                    const int s = sw[a[g]][pick[g]];
                    int need = 0, have = 0;
                    for (int h = 0; h < G; ++h) if (sw[a[h]][pick[h]] == s) need++;
                    for (int q = 0; q < nd->n_gpus; ++q)
                        if (!c->gpu_used[nd->gpu_off + q] && c->gpu_sw[nd->gpu_off + q] == s) have++;
                    if (have < need) fine = 0;
                }
            }
            if (fine) return 1;
            int pos = G - 1;
            while (pos >= 0) { if (++pick[pos] < K[a[pos]]) break; pick[pos] = 0; --pos; }
            if (pos < 0) break;
        }
    }
    return 0;
}

/* winner[p] = node index or -1; feas (optional) = P x N bytes */
void oracle_find(const ocluster* c, const opod* pods, int64_t P, double now, int64_t* winner, uint8_t* feas) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t p = 0; p < P; ++p) {
        int any_gpu = 0;
        for (int g = 0; g < pods[p].G; ++g) if (pods[p].n_gpus[g] > 0) any_gpu = 1;
        int64_t first = -1, first_nogpu = -1;
        for (int64_t i = 0; i < c->n; ++i) {
            const int f = oracle_feasible(c, i, &pods[p], now);
            if (feas) feas[p * c->n + i] = (uint8_t)f;
            if (!f) continue;
            if (first < 0) first = i;
            if (first_nogpu < 0 && c->nodes[i].n_gpus == 0) first_nogpu = i;
        }
        winner[p] = (!any_gpu && first_nogpu >= 0) ? first_nogpu : first;   /* Matcher.py:401-421 */
    }
}

/* ---- mode B support: first-fit scan with early exit, and the commit step on the flat records ---------------------
 * Reference lines followed:
 *   SelectNode (first candidate / first GPU-less candidate)     nhd/Matcher.py:393-421
 *   SetBusy                                                     nhd/Node.py:843-845
 *   SetPhysicalIdsFromMapping                                   nhd/Node.py:663-841
 *   GetFreeCpuBatch                                             nhd/Node.py:502-519
 *   GetNicObjFromIndex / GetFreePciGpuFromNic / GetNextGpuFree  nhd/Node.py:657-661, 648-655, 495-500
 *   ClaimPodNICResources                                        nhd/Node.py:644-646, nhd/NHDScheduler.py:302-304
 * The mapping itself (GetNumaGroupIdx, CPython set order) stays in oracle/nhd_oracle.py: oracle/seq_oracle.py asks it
 * for the winner only. */

/* first feasible node in [0, n) - restricted to nodes without GPUs installed when only_nogpu - or -1.
 * Blocks of nodes are handed out in ascending order; a thread stops once its block starts past the best hit. */
int64_t oracle_first_feasible(const ocluster* c, const opod* p, double now, int only_nogpu) {
    const int64_t B = 256, nblocks = (c->n + B - 1) / B;
    int64_t best = c->n, next = 0;
#pragma omp parallel
    {
        for (;;) {
            int64_t b, cur;
#pragma omp atomic capture
            b = next++;
#pragma omp atomic read
            cur = best;
            if (b >= nblocks || b * B >= cur) break;
            const int64_t hi = (b + 1) * B < c->n ? (b + 1) * B : c->n;
            for (int64_t i = b * B; i < hi; ++i) {
                if (only_nogpu && c->nodes[i].n_gpus != 0) continue;
                if (!oracle_feasible(c, i, p, now)) continue;
#pragma omp critical
                { if (i < best) best = i; }
                break;
            }
        }
    }
    return best < c->n ? best : -1;
}

/* GetFreeCpuBatch(numa, num, smt) on the flat core records of node nd (nhd/Node.py:502-519).  Scans EVERY core in
 * index order - sibling range included - and marks nothing while it scans, exactly like the reference. */
static int free_cpu_batch(const ocluster* c, const onode* nd, int numa, int num, int smt_requested, int32_t* out) {
    int got = 0;
    for (int ci = 0; ci < nd->n_cores; ++ci) {
        if (num == 0) break;
        const int o = nd->core_off + ci;
        if (c->core_socket[o] != numa || c->core_used[o]) continue;
        if (nd->smt) {
            const int sib = c->core_sibling[o];
            if (c->core_used[nd->core_off + sib]) continue;
            if (smt_requested && num >= 2) { out[got++] = ci; out[got++] = sib; num -= 2; }
            else { out[got++] = ci; num -= 1; }
        } else { out[got++] = ci; num -= 1; }
    }
    return got;
}

/* SetBusy + SetPhysicalIdsFromMapping + ClaimPodNICResources on node idx.
 *   map_numa[g] = mapping['gpu'][g], misc_numa = mapping['cpu'][-1], nic_numa/nic_idx[g] = mapping['nic'][g]
 *   nic_use[g]  = group g has an RX or TX core (its NIC is claimed)
 *   ids: per group its core batch (GPU cpu_cores first, then proc_cores - the batch in order), helper batch, GPU list
 *        positions; then the misc batch.  counts[3g..3g+2] = their lengths, counts[3G] = misc length.
 * Returns 0, or 1 where the reference raises IndexError (its unwind path is itself broken: parity undefined). */
int oracle_commit(const ocluster* c, int64_t idx, const opod* p, const int32_t* map_numa, int32_t misc_numa,
                  const int32_t* nic_numa, const int32_t* nic_idx, const int32_t* nic_use, int32_t misc_smt_enabled,
                  double now, int32_t* ids, int32_t* counts) {
    onode* nd = (onode*)&c->nodes[idx];
    uint8_t* core_used = (uint8_t*)c->core_used;
    uint8_t* gpu_used = (uint8_t*)c->gpu_used;
    int32_t* nic_pods = (int32_t*)c->nic_pods;
    int n_ids = 0;
    int claimed[MAXNIC * MAXU], n_claimed = 0;
    nd->busy_time = now;                                                    /* SetBusy */
    for (int g = 0; g < p->G; ++g) {
        const int numa = map_numa[g];
        int32_t batch[512];
        const int want = p->n_proc[g];                                       /* len(proc_cores) + sum(len(gpu.cpu_cores)) */
        const int got = free_cpu_batch(c, nd, numa, want, p->proc_smt[g], batch);
        if (got != want) return 1;
        /* GetNicObjFromIndex: idx is the per-NUMA ordinal in node.nics order (nhd/Node.py:413-418) */
        int nic_pos = -1, ord = 0;
        for (int k = 0; k < nd->n_nics && nic_pos < 0; ++k) {
            if (c->nic_numa[nd->nic_off + k] != nic_numa[g]) continue;
            if (ord == nic_idx[g]) nic_pos = k;
            ++ord;
        }
        if (nic_pos < 0) return 1;
        const int nsw = c->nic_sw[nd->nic_off + nic_pos];
        int n_gpu_ids = 0;
        int32_t gpu_ids[MAXNIC];
        for (int q = 0; q < p->n_gpus[g]; ++q) {
            int dev = -1;
            for (int x = 0; x < nd->n_gpus && dev < 0; ++x)                  /* GetFreePciGpuFromNic */
                if (c->gpu_sw[nd->gpu_off + x] == nsw && !gpu_used[nd->gpu_off + x]) dev = x;
            if (dev < 0) {
                if (p->map_type == 2) return 1;
                for (int x = 0; x < nd->n_gpus && dev < 0; ++x)              /* GetNextGpuFree */
                    if (c->gpu_numa[nd->gpu_off + x] == numa && !gpu_used[nd->gpu_off + x]) dev = x;
            }
            if (dev < 0) return 1;
            gpu_used[nd->gpu_off + dev] = 1;
            gpu_ids[n_gpu_ids++] = dev;
        }
        for (int k = 0; k < got; ++k) { core_used[nd->core_off + batch[k]] = 1; ids[n_ids++] = batch[k]; }
        if (nic_use[g]) claimed[n_claimed++] = nic_pos;
        int32_t helpers[512];
        const int hgot = free_cpu_batch(c, nd, numa, p->n_help[g], p->help_smt[g], helpers);
        if (hgot != p->n_help[g]) return 1;
        for (int k = 0; k < hgot; ++k) { core_used[nd->core_off + helpers[k]] = 1; ids[n_ids++] = helpers[k]; }
        for (int k = 0; k < n_gpu_ids; ++k) ids[n_ids++] = gpu_ids[k];
        counts[3 * g] = got; counts[3 * g + 1] = hgot; counts[3 * g + 2] = n_gpu_ids;
    }
    if (p->hp > 0) nd->hp_free -= p->hp;
    int32_t misc[512];
    const int mgot = free_cpu_batch(c, nd, misc_numa, p->n_misc, misc_smt_enabled, misc);
    if (mgot != p->n_misc) return 1;
    for (int k = 0; k < mgot; ++k) { core_used[nd->core_off + misc[k]] = 1; ids[n_ids++] = misc[k]; }
    counts[3 * p->G] = mgot;
    for (int k = 0; k < n_claimed; ++k) {                                    /* distinct NIC list indices, pods_used += 1 */
        int dup = 0;
        for (int j = 0; j < k; ++j) if (claimed[j] == claimed[k]) dup = 1;
        if (!dup) nic_pods[nd->nic_off + claimed[k]] += 1;
    }
    return 0;
}


int oracle_sizeof_node(void) { return (int)sizeof(onode); }
int oracle_sizeof_pod(void) { return (int)sizeof(opod); }
