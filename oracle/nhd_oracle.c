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

int oracle_sizeof_node(void) { return (int)sizeof(onode); }
int oracle_sizeof_pod(void) { return (int)sizeof(opod); }
