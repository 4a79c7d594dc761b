/*
 * nhdfit.h - C ABI of libnhdfit.so: the MI355X (gfx950) node filter-and-score engine that
 * replaces the per-node Python loop of the NHD scheduler's Matcher.FindNode.
 *
 * Boundary.  The reference has exactly one call site for this path,
 *     match = self.matcher.FindNode(filt_nodes, top)          nhd/NHDScheduler.py:277
 * and is pure Python, so the binding a maintainer adds is a ctypes stub; the
 * in-tree one is nhd_amd/_lib.py and the Matcher-compatible class is nhd_amd/matcher.py.
 *
 * Conventions: every function returns 0 on success or a negative NHDFIT_E_* code and stores a
 * message retrievable with nhdfit_last_error(); nothing throws across the boundary and nothing
 * calls abort().  The caller owns every host buffer; the library copies what it needs before
 * returning.  One context = one GPU = one caller thread at a time (the reference calls FindNode
 * from its single scheduler thread only, nhd/NHDScheduler.py:43,277).  Multi-GPU = one process
 * (one context) per GPU, joined by nhdfit_comm_init (RCCL all-reduce(max) of packed scores).
 *
 * There is deliberately no CPU implementation behind this ABI: if the HIP runtime or a gfx950
 * device is missing, nhdfit_create fails.
 */
#ifndef NHDFIT_H
#define NHDFIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NHDFIT_ABI_VERSION        9
#define NHDFIT_MAX_GROUPS         4      /* proc groups per pod (G) of the table-driven pass; 5..8: nhdfit_big_req below */
#define NHDFIT_MAX_NUMA           2      /* NUMA nodes (= sockets, nhd/Node.py:336) per node (U)  */
#define NHDFIT_MAX_CORES_PER_NUMA 64     /* physical cores per socket (one uint64 mask)           */
#define NHDFIT_MAX_GPUS           32     /* GPUs per node (one uint32 mask)                       */
#define NHDFIT_MAX_GPUS_PER_NUMA  8      /* GPUs on one NUMA node (pair-indexed GPU table)        */
#define NHDFIT_MAX_NICS_PER_NUMA  16
#define NHDFIT_MAX_SWITCHES       14     /* distinct PCIe switches per node                       */
#define NHDFIT_MAX_CLASSES        16     /* distinct NIC capacity values cluster-wide             */
#define NHDFIT_TILE               64     /* pods per tile = wavefront width                       */
#define NHDFIT_GLIMIT_NONE        255    /* pool without a GPU-per-switch limit (NUMA mode)       */

#define NHDFIT_OK              0
#define NHDFIT_E_INVAL        -1
#define NHDFIT_E_NODEVICE     -2
#define NHDFIT_E_HIP          -3
#define NHDFIT_E_NOMEM        -4
#define NHDFIT_E_STATE        -5
#define NHDFIT_E_LIMIT        -6         /* a compile-time capacity above was exceeded            */
#define NHDFIT_E_RCCL         -7

/* node flags (plane2.flags) */
#define NHDFIT_NF_MAINTENANCE  0x01u     /* Node.maintenance          nhd/Matcher.py:71           */
#define NHDFIT_NF_ACTIVE       0x02u     /* Node.active               nhd/NHDScheduler.py:242     */
#define NHDFIT_NF_SMT          0x04u     /* Node.smt_enabled          nhd/Node.py:225             */
#define NHDFIT_NF_HAS_GPU      0x08u     /* len(Node.gpus) > 0        nhd/Matcher.py:411          */

/* request flags */
#define NHDFIT_RF_INITIAL_FILTER 0x01u   /* apply InitialNodeFilter (active && groups intersect), nhd/NHDScheduler.py:235-247 */
#define NHDFIT_RF_NIC_SPLIT      0x02u   /* informational, set by the digests and ignored by the kernels: some group has several RX (or TX)
                                            cores, its rx / tx are sums of their speeds.  Matters under ENABLE_SHARING only, where the
                                            reference's commit adds the speeds to speed_used one by one (nhd/Node.py:754): the
                                            caller decides whether the sum is the same f64 value (nhd_amd/pack.py, Packer.share_exact) */
#define NHDFIT_RF_NIC_SPLIT_DYADIC 0x04u /* beside NHDFIT_RF_NIC_SPLIT: every speed of such a group is a non-negative multiple of 2^-20 below
                                            2^31 - sums of such values are exact in f64 in any order while they stay below 2^33 */

/* map types = values of nhd.CfgTopology.TopologyMapType (nhd/CfgTopology.py:41-45) */
#define NHDFIT_MAP_INVALID 0u    /* never matches (Matcher.py:45-47) */
#define NHDFIT_MAP_NUMA 1u
#define NHDFIT_MAP_PCI  2u

/* ---- packed node state: five structure-of-array planes of 16 bytes per node --------------- */
typedef struct { uint64_t t0[NHDFIT_MAX_NUMA]; } nhdfit_plane0;   /* thread-0 "core unused" bit per physical core of socket u
                                                                      (logical id u*cores_per_proc+i), nhd/Node.py:250-264 */
typedef struct { uint64_t t1[NHDFIT_MAX_NUMA]; } nhdfit_plane1;   /* thread-1 (SMT sibling id + num_cores); all ones without SMT */
typedef struct {
    uint32_t gpu_free;      /* bit g: Node.gpus[g] unused                  nhd/Node.py:456-462 */
    uint32_t gpu_numa1;     /* bit g: Node.gpus[g].numa_node == 1                               */
    int32_t  hp_free;       /* Node.mem.free_hugepages_gb                  nhd/Matcher.py:78   */
    uint32_t flags;         /* NHDFIT_NF_*                                                      */
} nhdfit_plane2;
typedef struct {
    uint64_t groups;        /* interned NHD_GROUP set                      nhd/Node.py:308-321 */
    uint16_t sig_numa[NHDFIT_MAX_NUMA];   /* NIC signature id of NUMA u, all NICs in one pool   */
    uint16_t sig_pci[NHDFIT_MAX_NUMA];    /* ... one pool per PCIe switch with its free-GPU cap */
} nhdfit_plane3;
typedef struct {
    double   busy_time;     /* Node.busy_time (monotonic seconds)          nhd/Node.py:843-850 */
    uint32_t group_set;     /* id of the node's interned NHD_GROUP set (row of the per-tile group table) */
    uint32_t reserved;
} nhdfit_plane4;

/* cold per-node detail, gathered only for winners (mapping step, nhd/Matcher.py:423-452) */
typedef struct {
    uint8_t nic_cnt[NHDFIT_MAX_NUMA];
    uint8_t sw_free[NHDFIT_MAX_SWITCHES];                          /* free GPUs on local switch id s, nhd/Node.py:266-273 */
    uint8_t nic_cls[NHDFIT_MAX_NUMA][NHDFIT_MAX_NICS_PER_NUMA];    /* capacity class of NIC (numa, idx)                   */
    uint8_t nic_sw[NHDFIT_MAX_NUMA][NHDFIT_MAX_NICS_PER_NUMA];     /* local switch id of NIC (numa, idx), nhd/Node.py:275 */
    uint8_t numa_nodes;                                            /* Node.numa_nodes (1 or 2)                            */
    uint8_t n_gpus;                                                /* len(Node.gpus)                                      */
    uint8_t nic_pods[12];                                          /* Node.nics[].pods_used as 32 three-bit counters, NIC (numa, idx)
                                                                      = counter numa*16+idx, two's complement -3 .. 3; the pattern
                                                                      4 = out of range, sticky: the host re-packs the node
                                                                      (NHDFIT_DELTA_REPACK).  nhd/Node.py:292,644-646            */
    uint8_t pad[2];
    uint8_t gpu_sw[NHDFIT_MAX_GPUS];                               /* local switch id of Node.gpus[g] (commit step, nhd/Node.py:648-655) */
} nhdfit_detail;                                                   /* 128 bytes */

/* cold per-node record no commit changes: what ResetResources goes back to (nhd/Node.py:144-161) and each NIC's own
 * capacity class (its class whenever pods_used <= 0, nhd/Node.py:292) */
typedef struct {
    uint64_t t0[NHDFIT_MAX_NUMA], t1[NHDFIT_MAX_NUMA];             /* planes 0 / 1 with every core outside reserved_cores unused */
    uint8_t  nic_base[NHDFIT_MAX_NUMA][NHDFIT_MAX_NICS_PER_NUMA];  /* capacity class of speed * 0.9 of NIC (numa, idx)           */
    int32_t  hp_total;                                             /* Node.mem.ttl_hugepages_gb                                   */
    uint8_t  pad[12];
} nhdfit_origin;                                                   /* 80 bytes */

/* ---- one change of a node outside the commit step ("K3 delta update"): the scheduler's
 * release / reclaim paths and its writes to scalar node fields, applied to the device mirror in place of re-packing
 * and re-uploading the node.  The host turns the ids a CfgTopology holds into masks over the planes. */
#define NHDFIT_DELTA_TAKE          1u   /* Node.RemoveResourcesFromTopology  nhd/Node.py:530-585 (start-up replay, NHDScheduler.py:135) */
#define NHDFIT_DELTA_GIVE          2u   /* Node.AddResourcesFromTopology     nhd/Node.py:587-636 (pod deleted, NHDScheduler.py:203)     */
#define NHDFIT_DELTA_RESET         3u   /* Node.ResetResources               nhd/Node.py:144-161                                        */
#define NHDFIT_DELTA_SET_FLAGS     4u   /* node.active / node.maintenance    nhd/NHDScheduler.py:533-566                                */
#define NHDFIT_DELTA_SET_GROUPS    5u   /* Node.SetGroups                    nhd/Node.py:308-310, NHDScheduler.py:570                   */
#define NHDFIT_DELTA_SET_BUSY      6u   /* Node.SetBusy / busy_time          nhd/Node.py:843-845                                        */
#define NHDFIT_DELTA_SET_HUGEPAGES 7u   /* Node.SetHugepages(alloc, free)    nhd/Node.py:489-493                                        */
#define NHDFIT_DELTA_MAX_NICS      15
typedef struct {
    uint32_t node;                           /* local index in the mirror                                                     */
    uint32_t op;                             /* NHDFIT_DELTA_*                                                                 */
    uint64_t t0[NHDFIT_MAX_NUMA], t1[NHDFIT_MAX_NUMA];   /* TAKE / GIVE: the cores the topology names, as bits of planes 0 / 1   */
    uint32_t gpus;                           /* TAKE / GIVE: bit x = Node.gpus[x] named by a group_gpus entry (by device id)      */
    int32_t  hugepages_gb;                   /* TAKE / GIVE: top.hugepages_gb (applied when > 0); SET_HUGEPAGES: the new free count */
    int32_t  hp_total;                       /* SET_HUGEPAGES: alloc                                                           */
    uint32_t flags_mask, flags_value;        /* SET_FLAGS: which of NHDFIT_NF_MAINTENANCE / NHDFIT_NF_ACTIVE, and their values   */
    uint32_t group_set;                      /* SET_GROUPS: id of the interned set                                             */
    uint64_t groups;                         /* SET_GROUPS: interned bits                                                      */
    double   busy_time;                      /* SET_BUSY                                                                       */
    uint8_t  nic_n;                          /* TAKE / GIVE: nic_core_pairing entries whose MAC is one of the node's NICs       */
    uint8_t  nic[NHDFIT_DELTA_MAX_NICS];     /* (numa << 4) | idx per entry: pods_used +1 / -1 each (a NIC may repeat)          */
} nhdfit_delta;                              /* 96 bytes */
#define NHDFIT_DELTA_OK      0
#define NHDFIT_DELTA_REPACK  1   /* a pods_used counter left the range the packed form tracks: only the Node object knows the
                                    NIC's state now - re-pack and upload the node                                            */
#define NHDFIT_DELTA_NEW_SIG 2   /* applied, but the node's new NIC state has no signature in the dictionary yet: intern it
                                    (download the node, re-derive its signatures - or re-pack it) and upload its plane 3     */

/* ---- NIC signature dictionary (cluster-wide, interned by the host packer) ----------------- */
typedef struct { uint8_t cls; uint8_t cnt; } nhdfit_cc;           /* cnt NICs (capped at NHDFIT_MAX_GROUPS) of capacity class cls */

/* ---- one pending pod: the digest of a CfgTopology (nhd/CfgTopology.py:126-232) ------------ */
typedef struct {
    uint32_t n_groups;                       /* len(top.proc_groups), 1..NHDFIT_MAX_GROUPS                         */
    uint32_t map_type;                       /* TopologyMapType value; anything but NUMA/PCI never matches (Matcher.py:45-47) */
    int32_t  hugepages_gb;                   /* top.hugepages_gb                                                   */
    uint32_t flags;                          /* NHDFIT_RF_*                                                        */
    uint64_t groups;                         /* interned pod node-group set (K8SMgr.py:152-165); used with RF_INITIAL_FILTER */
    uint16_t gpus[NHDFIT_MAX_GROUPS];        /* GetTotalGpusRequested, CfgTopology.py:199                          */
    uint16_t cpu_smt[NHDFIT_MAX_GROUPS];     /* physical cores group i needs on an SMT node   (Matcher.py:178-196) */
    uint16_t cpu_nosmt[NHDFIT_MAX_GROUPS];   /* ... on a non-SMT node                                              */
    uint16_t misc_smt;                       /* pod-level misc cores on an SMT node (always halved, Matcher.py:198)*/
    uint16_t misc_nosmt;
    /* raw counts for the commit step (Node.SetPhysicalIdsFromMapping, nhd/Node.py:663-841) */
    uint8_t  n_proc[NHDFIT_MAX_GROUPS];      /* len(proc_cores) + sum(len(gpu.cpu_cores)): one GetFreeCpuBatch with proc_smt */
    double   rx[NHDFIT_MAX_GROUPS];          /* GetTotalNICsRequested, CfgTopology.py:219-232 (Gb/s, Python float) */
    double   tx[NHDFIT_MAX_GROUPS];
    uint8_t  n_help[NHDFIT_MAX_GROUPS];      /* len(group.misc_cores): second batch with helper_smt                  */
    uint8_t  n_misc;                         /* len(top.misc_cores): last batch with the real misc_cores_smt flag     */
    uint8_t  smt_bits;                       /* bit g: group g proc_smt enabled; bit 4+g: helper_smt enabled           */
    uint8_t  misc_smt_enabled;               /* top.misc_cores_smt == SMT_ENABLED (Node.py:799)                        */
    uint8_t  nic_use;                        /* bit g: group g has RX/TX cores -> its NIC is claimed (Node.py:744-764, 644) */
} nhdfit_req;                                /* 128 bytes */

/* ---- resource mapping of one placement = the dict FindNode returns (Matcher.py:452) ------- */
typedef struct {
    int8_t gpu[NHDFIT_MAX_GROUPS];           /* mapping['gpu'][i]  : NUMA node of group i       */
    int8_t cpu[NHDFIT_MAX_GROUPS + 1];       /* mapping['cpu']     : ... plus the misc cores'   */
    int8_t nic_numa[NHDFIT_MAX_GROUPS];      /* mapping['nic'][i][0]                            */
    int8_t nic_idx[NHDFIT_MAX_GROUPS];       /* mapping['nic'][i][1] : per-NUMA NIC ordinal     */
    int8_t valid;                            /* 0 = pod not placed                              */
    int8_t pad[2];
} nhdfit_mapping;                            /* 20 bytes */

/* ---- physical ids of one committed placement = what Node.SetPhysicalIdsFromMapping writes into the pod's
 * CfgTopology (nhd/Node.py:663-841).  A core batch (one GetFreeCpuBatch call, nhd/Node.py:502-519) is three masks
 * over the physical cores of its socket: `take` = cores whose thread 0 was handed out, `pair` = those whose SMT
 * sibling was handed out with it, `late` = those whose sibling was handed out as a core of its own when the walk ran
 * on into the sibling range (a request without the SMT flag that the thread-0 cores could not fill - the pod-level
 * misc cores under quirk Q1).  The reference's list is, for ascending core b in `take`: logical id
 * numa * cores_per_proc + b, then - if b is in `pair` - its sibling (id + num_cores); then, for ascending b in `late`,
 * the sibling id.  It assigns that list to the group's GPU cpu_cores first, then to its proc_cores (nhd/Node.py:729-742). */
#define NHDFIT_PLACEMENT_GPUS 8
typedef struct {
    uint64_t proc_take[NHDFIT_MAX_GROUPS], proc_pair[NHDFIT_MAX_GROUPS];   /* batch of len(proc_cores) + GPU cores, proc_smt   */
    uint64_t help_take[NHDFIT_MAX_GROUPS], help_pair[NHDFIT_MAX_GROUPS];   /* batch of the group's helper cores, helper_smt    */
    uint64_t misc_take, misc_pair;                                         /* pod-level misc cores, misc_cores_smt (Node.py:799) */
    uint8_t  gpu[NHDFIT_MAX_GROUPS][NHDFIT_PLACEMENT_GPUS];                /* position in Node.gpus of the k-th GPU of group g, 0xFF = none */
    int8_t   numa[NHDFIT_MAX_GROUPS + 1];                                  /* NUMA node of group g / of the misc cores          */
    uint8_t  status;                                                       /* NHDFIT_COMMIT_*                                   */
    uint8_t  pad[2];
    uint64_t proc_late[NHDFIT_MAX_GROUPS], help_late[NHDFIT_MAX_GROUPS], misc_late;   /* second threads handed out by the run-on walk */
} nhdfit_placement;                                                        /* 256 bytes */
#define NHDFIT_COMMIT_OK          0
#define NHDFIT_COMMIT_WOULD_RAISE 1   /* the reference's commit raises IndexError (a batch short after the walk over both thread
                                         ranges, no free GPU on a PCI-mode group's switch, no such NIC) or hands an SMT request
                                         its own cores twice: parity undefined from here on; the mirror is left as the walk left it */
#define NHDFIT_COMMIT_WIDE        3   /* nhdfit_schedule_batch: the pod landed on a wide node - its physical ids are in nhdfit_wide_placements */
#define NHDFIT_COMMIT_NEW_SIG     2   /* committed, but the node's new NIC state has no signature in the dictionary yet: intern
                                         it (download the node, re-derive its signatures) and upload its plane 3           */

/* ---- nodes beyond the fast layout: the general path ("wide" nodes) -------------------------------------------------------
 * The reference enumerates range(v.numa_nodes) for any socket count and repeat=len(req) for any group count
 * (nhd/Matcher.py:118,203,242) over any cores_per_proc (nhd/Node.py:257,336-350).  The planes above hold two sockets of up to
 * 64 physical cores - what the table-driven P x N pass is built for.  A node outside that shape (3 or 4 sockets, 65..128
 * physical cores per socket) is mirrored TWICE: as a placeholder in the planes (flags = NHDFIT_NF_MAINTENANCE: the table pass
 * never matches it, node indices stay what they are) and as one self-contained record below, which the general path evaluates
 * by explicit enumeration - lane = (wide node, pod) - with the reference's own arithmetic (wide_core.h): its verdict bit lands
 * in the same verdict matrix, its score in the same score word (atomicMax: the word carries the global node index, so the first
 * feasible node of the whole candidate order wins wherever it is mirrored), its mapping comes from the general CPython set
 * model (tuples over range(U), tables of up to 4 096 slots).  The commit step and mode B are served for these nodes too
 * (nhdfit_wide_commit; nhdfit_schedule_batch decides pod by pod when the mirror holds wide nodes).  Release / reclaim / reset
 * re-upload the node's record (640 bytes) instead of travelling as a delta.  Slower per (pod, node) pair than the table pass by
 * three orders of magnitude - and exact. */
#define NHDFIT_WIDE_MAX_NUMA      4      /* sockets (= NUMA nodes, nhd/Node.py:336) of a wide node                         */
#define NHDFIT_WIDE_CORE_WORDS    8      /* flat bitmaps over the node's physical cores: 512 = 4 sockets x 128            */
#define NHDFIT_WIDE_MAX_CORES_PER_NUMA 128
typedef struct {
    uint64_t t0[NHDFIT_WIDE_CORE_WORDS];      /* bit c: logical core c (thread 0 of physical core c, socket c / cores_per_proc) unused */
    uint64_t t1[NHDFIT_WIDE_CORE_WORDS];      /* bit c: its SMT sibling (logical id c + num_cores) unused; all ones without SMT         */
    uint64_t o0[NHDFIT_WIDE_CORE_WORDS];      /* t0 / t1 as ResetResources leaves them (nhd/Node.py:144-161)                            */
    uint64_t o1[NHDFIT_WIDE_CORE_WORDS];
    uint64_t groups;                          /* interned NHD_GROUP set                                                                  */
    double   busy_time;                       /* Node.busy_time                                                                          */
    uint32_t gpu_free;                        /* bit g: Node.gpus[g] unused                                                              */
    uint32_t flags;                           /* NHDFIT_NF_*                                                                             */
    int32_t  hp_free, hp_total;               /* Node.mem.free_hugepages_gb / ttl_hugepages_gb                                           */
    uint32_t index;                           /* the node's local index in the mirror (candidate order)                                  */
    uint16_t cores_per_proc;                  /* physical cores per socket, 1..128                                                       */
    uint8_t  numa_nodes;                      /* 1..4                                                                                    */
    uint8_t  n_gpus;                          /* len(Node.gpus), <= 32                                                                   */
    uint8_t  nic_cnt[NHDFIT_WIDE_MAX_NUMA];
    uint8_t  gpu_numa[NHDFIT_MAX_GPUS];       /* Node.gpus[g].numa_node                                                                  */
    uint8_t  gpu_sw[NHDFIT_MAX_GPUS];         /* local id of its PCIe switch (ids are per node: equal id = same switch)                  */
    uint8_t  nic_cls[NHDFIT_WIDE_MAX_NUMA][NHDFIT_MAX_NICS_PER_NUMA];    /* capacity class of NIC (numa, idx) as it is now              */
    uint8_t  nic_base[NHDFIT_WIDE_MAX_NUMA][NHDFIT_MAX_NICS_PER_NUMA];   /* ... of speed * pct: its class whenever pods_used <= 0        */
    uint8_t  nic_sw[NHDFIT_WIDE_MAX_NUMA][NHDFIT_MAX_NICS_PER_NUMA];     /* local switch id                                               */
    int8_t   nic_pods[NHDFIT_WIDE_MAX_NUMA][NHDFIT_MAX_NICS_PER_NUMA];   /* Node.nics[].pods_used, saturating at +-127                    */
    uint8_t  pad[20];
} nhdfit_wide_node;                           /* 640 bytes */

/* physical ids of one placement on a wide node: nhdfit_placement with two mask words per batch (bit b of the pair = physical
 * core b of the batch's socket, logical id numa * cores_per_proc + b; sibling id + num_cores) */
typedef struct {
    uint64_t proc_take[NHDFIT_MAX_GROUPS][2], proc_pair[NHDFIT_MAX_GROUPS][2], proc_late[NHDFIT_MAX_GROUPS][2];
    uint64_t help_take[NHDFIT_MAX_GROUPS][2], help_pair[NHDFIT_MAX_GROUPS][2], help_late[NHDFIT_MAX_GROUPS][2];
    uint64_t misc_take[2], misc_pair[2], misc_late[2];
    uint8_t  gpu[NHDFIT_MAX_GROUPS][NHDFIT_PLACEMENT_GPUS];
    int8_t   numa[NHDFIT_MAX_GROUPS + 1];
    uint8_t  status;                          /* NHDFIT_COMMIT_OK / NHDFIT_COMMIT_WOULD_RAISE */
    uint8_t  pad[2];
    uint32_t pod;                             /* nhdfit_wide_placements: index of the pod in the batch */
    uint32_t node;                            /* ... and the local index of the node                    */
} nhdfit_wide_placement;                      /* 480 bytes */

/* ---- nhd/Node.py:20 ENABLE_SHARING = True (ABI 9) ----------------------------------------------------------------------------
 * The shipped reference prices a NIC at its full capacity until a pod uses it, then at nothing (pods_used).  With the module
 * constant flipped, GetFreeNumaNicResources (nhd/Node.py:289-291) prices it per direction at
 *     n.speed * NIC_BW_AVAIL_PERCENT - n.speed_used[x]            x = 0 (rx), 1 (tx)
 * and the commit step adds every RX / TX core's speed to speed_used (nhd/Node.py:754).  Such capacities are sums of whatever the
 * pods asked for, not a handful of classes, so a cluster whose node module has the switch on is mirrored for the GENERAL path
 * only: every node a wide record, and beside record k one nhdfit_wide_share with the NICs' speed_used as they are - the path's own
 * f64 arithmetic then computes `capacity(nic_base) - used[x]` exactly as the reference does, before the per-group subtractions
 * (wide_core.h).  nhdfit_wide_share_upload(NULL) switches a context back to the shipped arithmetic. */
typedef struct {
    double used[NHDFIT_WIDE_MAX_NUMA][NHDFIT_MAX_NICS_PER_NUMA][2];   /* Node.nics[..].speed_used[0 / 1] of NIC (numa, idx) */
} nhdfit_wide_share;                              /* 1024 bytes */

/* ---- pods with more than NHDFIT_MAX_GROUPS processing groups: the general path for requests ("big" requests) ---------------
 * The reference enumerates itertools.product(range(numa_nodes), repeat=len(req)) for whatever len(top.proc_groups) is
 * (nhd/Matcher.py:118,203,242).  The table-driven pass is built around masks over 2^G <= 16 assignments; a pod with 5..8
 * processing groups is carried as a record of its own kind and answered by the general path - explicit enumeration with the
 * reference's own arithmetic (wide_core.h), lane = node, against EVERY node of the mirror (ordinary nodes are read through the
 * same view a wide node's record gives) - its winner by the same score word, its mapping from the general CPython set model,
 * its commit step on whichever form the winner is mirrored in.  Every node either layout holds (<= 4 sockets) answers for any
 * G <= 8: up to NHDFIT_BIG_MAX_TUPLES = 4^9 assignment tuples, the set model's tables sized per call.  One bound: a search
 * budget per (pod, node) for the NIC stage (NHDFIT_BIG_NIC_BUDGET steps of the pruned depth-first search; the reference's own
 * enumeration at such a pair is K^G deepcopies): exceeding it fails the call (NHDFIT_E_LIMIT), it is never answered wrong. */
#define NHDFIT_BIG_MAX_GROUPS     8
#define NHDFIT_BIG_MAX_TUPLES     262144
#define NHDFIT_BIG_NIC_BUDGET     (1u << 22)
typedef struct {
    uint32_t n_groups;                           /* len(top.proc_groups), 1..NHDFIT_BIG_MAX_GROUPS                       */
    uint32_t map_type;
    int32_t  hugepages_gb;
    uint32_t flags;                              /* NHDFIT_RF_*                                                          */
    uint64_t groups;
    uint16_t gpus[NHDFIT_BIG_MAX_GROUPS];
    uint16_t cpu_smt[NHDFIT_BIG_MAX_GROUPS];
    uint16_t cpu_nosmt[NHDFIT_BIG_MAX_GROUPS];
    uint16_t misc_smt, misc_nosmt;
    uint16_t smt_bits;                           /* bit g: group g proc_smt enabled; bit 8+g: helper_smt enabled          */
    uint8_t  n_misc;
    uint8_t  misc_smt_enabled;
    double   rx[NHDFIT_BIG_MAX_GROUPS];
    double   tx[NHDFIT_BIG_MAX_GROUPS];
    uint8_t  n_proc[NHDFIT_BIG_MAX_GROUPS];
    uint8_t  n_help[NHDFIT_BIG_MAX_GROUPS];
    uint8_t  nic_use;                            /* bit g: group g has RX/TX cores                                       */
    uint8_t  pad[31];
} nhdfit_big_req;                                /* 256 bytes; field for field nhdfit_req with eight groups               */
typedef struct {
    int8_t gpu[NHDFIT_BIG_MAX_GROUPS];
    int8_t cpu[NHDFIT_BIG_MAX_GROUPS + 1];
    int8_t nic_numa[NHDFIT_BIG_MAX_GROUPS];
    int8_t nic_idx[NHDFIT_BIG_MAX_GROUPS];
    int8_t valid;
    int8_t pad[2];
} nhdfit_big_mapping;                            /* 36 bytes */
/* physical ids of a big request's placement: nhdfit_wide_placement with eight groups (an ordinary node uses word 0 of each pair) */
typedef struct {
    uint64_t proc_take[NHDFIT_BIG_MAX_GROUPS][2], proc_pair[NHDFIT_BIG_MAX_GROUPS][2], proc_late[NHDFIT_BIG_MAX_GROUPS][2];
    uint64_t help_take[NHDFIT_BIG_MAX_GROUPS][2], help_pair[NHDFIT_BIG_MAX_GROUPS][2], help_late[NHDFIT_BIG_MAX_GROUPS][2];
    uint64_t misc_take[2], misc_pair[2], misc_late[2];
    uint8_t  gpu[NHDFIT_BIG_MAX_GROUPS][NHDFIT_PLACEMENT_GPUS];
    int8_t   numa[NHDFIT_BIG_MAX_GROUPS + 1];
    uint8_t  status;                             /* NHDFIT_COMMIT_OK / _WOULD_RAISE / _NEW_SIG (ordinary node: as nhdfit_commit) */
    uint8_t  pad[2];
    uint32_t pod;
    uint32_t node;
    uint8_t  pad2[4];
} nhdfit_big_placement;                          /* 904 bytes */

typedef struct {
    uint64_t launches;          /* step-kernel launches carrying a fit role that were timed (every 8th step)   */
    double   fit_ms_total;      /* sum of their HIP-event durations (ms): the whole fused launch - fit role plus
                                   the digest / mapping roles of neighbouring steps that share it              */
    double   fit_ms_last;
    double   digest_ms_last;    /* last digest-only launch (the first step after staging has nothing to share) */
    double   step_ms_last;      /* = fit_ms_last (one launch per step)                                         */
    uint64_t evals_last;        /* pod x node evaluations of the last step                                      */
    uint64_t bytes_last;        /* algorithmic bytes of the last fit role                 */
    uint32_t nodes, nsig, ncls, lds_bytes;
    uint32_t pipes;             /* step launches in flight at a time: the pipelined form alternates its steps between this many
                                   independent pipelines on as many streams - a launch's HIP-event duration then overlaps its
                                   neighbour's, and throughput is pipes x (work per launch) / duration                          */
    uint32_t small_finds;       /* nhdfit_find calls answered by the single-launch form (at most one pod tile, no verdict matrix) */
    uint32_t big_nic_steps_max; /* nhdfit_big_find: the most NIC-search steps any (pod, node) pair took since the last reset - to be read against
                                   NHDFIT_BIG_NIC_BUDGET, at which a call fails (ABI 9) */
    uint32_t batch_finds;       /* nhdfit_find calls of more than one pod tile answered by ONE launch (k_findn): digest, fit and mapping tile by
                                   tile inside it, results through a fine-grained host block (the word was padding before: same layout) */
} nhdfit_stats;

typedef struct nhdfit_ctx nhdfit_ctx;

/* score word: 0 = no feasible node, else bit63 = preference hit (GPU-less node for a GPU-less pod,
 * Matcher.py:401-413), low 63 bits = 0x7FFFFFFFFFFFFFFF - global node index, so that
 * max(score) = first feasible node in candidate order with the preference applied (Matcher.py:415-421). */
#define NHDFIT_SCORE_NONE 0ull
#define NHDFIT_SCORE_INDEX(s) (0x7FFFFFFFFFFFFFFFull - ((s) & 0x7FFFFFFFFFFFFFFFull))

int  nhdfit_abi_version(void);
int  nhdfit_device_count(void);
int  nhdfit_create(int device_id, nhdfit_ctx** out);
void nhdfit_destroy(nhdfit_ctx* ctx);
const char* nhdfit_last_error(nhdfit_ctx* ctx);            /* ctx may be NULL: last error of a failed create */

/* NIC capacity classes + signature dictionary (CSR: sig -> pools -> (class,count) pairs) and the
 * largest number of physical cores per socket / GPUs per NUMA node anywhere in the cluster (they
 * size the per-pod tables).  Signature 0 must be the empty signature.  May be called again when the
 * dictionary grows. */
int nhdfit_set_dictionary(nhdfit_ctx* ctx, uint32_t max_cores_per_numa, uint32_t max_gpus_per_numa,
                          const uint64_t* group_sets, uint32_t n_group_sets,   /* distinct node-group bit sets, id = index */
                          const double* caps, uint32_t ncls,
                          const uint32_t* sig_off, uint32_t nsig,
                          const uint32_t* pool_off, const uint8_t* pool_glimit, uint32_t npools,
                          const nhdfit_cc* cc, uint32_t ncc);

/* (Re)size the device mirror to `capacity` nodes of which this rank's shard starts at global
 * index `global_base` (candidate order of the whole cluster; used for the score word).
 * Growing the capacity discards the mirror's contents (node count becomes 0). */
int nhdfit_reserve_nodes(nhdfit_ctx* ctx, uint32_t capacity, uint64_t global_base);
/* Full or delta upload of `count` node records starting at local index `first`. */
int nhdfit_upload_nodes(nhdfit_ctx* ctx, uint32_t first, uint32_t count,
                        const nhdfit_plane0* p0, const nhdfit_plane1* p1, const nhdfit_plane2* p2,
                        const nhdfit_plane3* p3, const nhdfit_plane4* p4, const nhdfit_detail* detail);
int nhdfit_set_node_count(nhdfit_ctx* ctx, uint32_t n_nodes);

/* The wide records of the mirror's nodes [first, first + count): `n_wide` records (ascending `index`, each inside the range)
 * replace whatever wide records the mirror held for that range (none: the range holds ordinary nodes only).  Call it for every
 * range nhdfit_upload_nodes is called for when the packer produced wide nodes (their planes carry the placeholder). */
int nhdfit_wide_upload(nhdfit_ctx* ctx, uint32_t first, uint32_t count, const nhdfit_wide_node* wide, uint32_t n_wide);
/* ENABLE_SHARING = True (nhd/Node.py:20, 289-291; see nhdfit_wide_share): record k = the NICs' speed_used of wide record k, for
 * ALL wide records of the mirror, which must be all of its nodes; sent again after every nhdfit_wide_upload.  NULL: back to the
 * shipped arithmetic.  nhdfit_wide_share_download reads them back as the device's commits left them. */
int nhdfit_wide_share_upload(nhdfit_ctx* ctx, const nhdfit_wide_share* share, uint32_t n_wide);
int nhdfit_wide_share_download(nhdfit_ctx* ctx, nhdfit_wide_share* out, uint32_t cap, uint32_t* n_wide);
/* how many wide records the mirror holds / read them back (ascending index) */
int nhdfit_wide_count(nhdfit_ctx* ctx, uint32_t* n_wide);
int nhdfit_wide_download(nhdfit_ctx* ctx, nhdfit_wide_node* out, uint32_t cap, uint32_t* n_wide);
/* nhdfit_commit for a wide node (`node` = its local index in the mirror): updates its record, returns the physical ids */
int nhdfit_wide_commit(nhdfit_ctx* ctx, uint32_t node, const nhdfit_req* req, const nhdfit_mapping* map, double busy_time,
                       nhdfit_wide_placement* place_out);
/* placements nhdfit_schedule_batch's last call made on wide nodes (their nhdfit_placement entries carry status
 * NHDFIT_COMMIT_WIDE): up to `cap` records, *n = how many there are */
int nhdfit_wide_placements(nhdfit_ctx* ctx, nhdfit_wide_placement* out, uint32_t cap, uint32_t* n);

/* FindNode for `P` big requests (mode A: one snapshot), every node of the mirror - ordinary and wide - by the general path.
 * cand / score_out / map_out as nhdfit_find (the score word is the same: with a communicator attached the P words are
 * all-reduced(max) and the owner of each winner maps it).  NHDFIT_E_LIMIT: a (pod, node) pair ran out of NIC search budget. */
int nhdfit_big_find(nhdfit_ctx* ctx, const nhdfit_big_req* reqs, uint32_t P, double now, const uint64_t* cand,
                    uint64_t* score_out, nhdfit_big_mapping* map_out);
/* nhdfit_commit / nhdfit_wide_commit for a big request, on whichever form `node` (local index) is mirrored in */
int nhdfit_big_commit(nhdfit_ctx* ctx, uint32_t node, const nhdfit_big_req* req, const nhdfit_big_mapping* map, double busy_time,
                      nhdfit_big_placement* place_out);

/* The nhdfit_origin records of nodes [first, first+count) (needed by nhdfit_apply_deltas; uploaded next to the planes). */
int nhdfit_upload_origin(nhdfit_ctx* ctx, uint32_t first, uint32_t count, const nhdfit_origin* origin);

/* K3: apply `n` deltas to the mirror, in array order (deltas of one node keep their order; different nodes are independent).
 * Runs on the context's stream behind every step in flight; blocks until status_out (n bytes, NHDFIT_DELTA_*) is on the host. */
int nhdfit_apply_deltas(nhdfit_ctx* ctx, const nhdfit_delta* deltas, uint32_t n, uint8_t* status_out);

/* Read node records back (after device-side commits). */
int nhdfit_download_nodes(nhdfit_ctx* ctx, uint32_t first, uint32_t count,
                          nhdfit_plane0* p0, nhdfit_plane1* p1, nhdfit_plane2* p2,
                          nhdfit_plane3* p3, nhdfit_plane4* p4, nhdfit_detail* detail);

/* Evaluate P pending pods against the mirror (mode A: every pod sees the same snapshot).
 *   now        monotonic clock sampled once per call (IsBusy, nhd/Node.py:847-850)
 *   cand       optional, [ceil(n/64)] words, bit = node: restrict the call to these nodes
 *              (the dict the scheduler passes to FindNode may be a filtered subset of the mirror)
 *   score_out  P score words (after the all-reduce when a communicator is attached)
 *   bitmap_out optional, chunk-major [ceil(n/64)][P] feasibility words of THIS shard
 *   map_out    optional, P mappings (valid only for winners owned by this shard)
 * A call with at most one pod tile (64 pods), no bitmap_out, no communicator and no pod with four processing groups - the
 * scheduler's pod-at-a-time FindNode (nhd/NHDScheduler.py:277) - is ONE kernel launch: digest, fit, mapping in a row inside
 * it, the requests read from and the results stored into fine-grained host memory (nhdfit_stats.small_finds counts them);
 * a lone pod skips the table image altogether (every block derives the pod's own assignment masks and sweeps nodes with them).
 * A larger call without bitmap_out, communicator, wide nodes or four-group pods is ONE launch as well (nhdfit_stats.batch_finds):
 * the batch is sorted into tiles as nhdfit_stage_requests sorts it - up to 512 pods are then read by the launch straight from the
 * page-locked block, a larger batch travels by ONE copy command with its work items behind the records -, digest, fit and mapping run
 * tile by tile inside one kernel - a tile's fit blocks wait for its digest, the last of them maps its winners - and scores and mappings
 * arrive in fine-grained host memory behind one polled word.
 * Every other call stages the batch and runs the launches of a step (digest, fused step, drain).  The single-launch forms leave
 * nothing staged: nhdfit_enqueue_step / nhdfit_fetch need a nhdfit_stage_requests of their own. */
int nhdfit_find(nhdfit_ctx* ctx, const nhdfit_req* reqs, uint32_t P, double now,
                const uint64_t* cand, uint64_t* score_out, uint64_t* bitmap_out, nhdfit_mapping* map_out);

/* Mode B: like nhdfit_find, but pod k is matched against the cluster as left by the placements of pods
 * 0..k-1 of the same batch - the scheduler loop commits every winner (SetBusy, SetPhysicalIdsFromMapping,
 * ClaimPodNICResources; nhd/NHDScheduler.py:289-304) before it matches the next pending pod (:425-437).
 * The device mirror itself is not modified: the caller applies the placements to its Node objects and
 * re-uploads those nodes.  node_out[p] = global node index or -1; status_out[p] != 0 where the reference's
 * own commit step would have raised (parity undefined from there on).  Single shard only. */
int nhdfit_find_sequential(nhdfit_ctx* ctx, const nhdfit_req* reqs, uint32_t P, double now, const uint64_t* cand,
                           int64_t* node_out, nhdfit_mapping* map_out, int32_t* status_out);

/* Mode B with the commit step on the device: every winner is committed to the packed node
 * state exactly as SetBusy / SetPhysicalIdsFromMapping / ClaimPodNICResources would (nhd/NHDScheduler.py:289-304,
 * nhd/Node.py:663-841, 644-646) before the next pod of the batch is matched, and the physical ids come back.
 *   apply      != 0: the commits stay in the device mirror (the scheduler's Node objects are updated by the caller
 *              with the reference's own mutators, nothing is re-packed or re-uploaded); 0: the mirror is restored
 *   place_out  optional, P records
 *   n_done     pods decided.  n_done < P only when a commit left a node in a NIC state the dictionary has no signature
 *              id for (status NHDFIT_COMMIT_NEW_SIG on one or more pods of [0, n_done)): intern the state, patch those
 *              nodes (nhdfit_download_nodes / nhdfit_set_dictionary / nhdfit_upload_nodes) and submit the remaining pods
 *              as a new batch - with apply != 0 the mirror already holds the placements made so far.  With the
 *              signature closure interned up front (nhd_amd.pack.Packer.close_signatures) this never happens.
 * The batch is decided on THIS context's nodes, with a communicator attached too (no collective inside): across shards the
 * loop's first-fit order is kept by handing the pods a shard could not place to the next rank, nhdfit_comm_sendrecv. */
int nhdfit_schedule_batch(nhdfit_ctx* ctx, const nhdfit_req* reqs, uint32_t P, double now, const uint64_t* cand, int apply,
                          int64_t* node_out, nhdfit_mapping* map_out, nhdfit_placement* place_out, int32_t* status_out,
                          uint32_t* n_done);

/* The commit step for ONE placement the caller obtained from nhdfit_find (the scheduler's pod-at-a-time loop):
 * updates the mirror in place and returns the physical ids.  busy_time = the node's Node.busy_time after SetBusy(). */
int nhdfit_commit(nhdfit_ctx* ctx, uint32_t node, const nhdfit_req* req, const nhdfit_mapping* map, double busy_time,
                  nhdfit_placement* place_out);

/* Benchmark / pipelined form: requests are staged once, then each step only enqueues kernels
 * (request digest -> fit_score -> [all-reduce] -> winner mapping) on the context's stream. */
int nhdfit_stage_requests(nhdfit_ctx* ctx, const nhdfit_req* reqs, uint32_t P);
int nhdfit_enqueue_step(nhdfit_ctx* ctx, double now);
int nhdfit_sync(nhdfit_ctx* ctx);
int nhdfit_fetch(nhdfit_ctx* ctx, uint64_t* score_out, uint64_t* bitmap_out, nhdfit_mapping* map_out);

/* Multi-GPU: rank 0 creates the id, every rank (one process per GPU) joins. */
int nhdfit_comm_unique_id(void* id128);
int nhdfit_comm_init(nhdfit_ctx* ctx, int nranks, int rank, const void* id128);
int nhdfit_comm_destroy(nhdfit_ctx* ctx);
/* Rank-to-rank traffic of mode B across shards (ABI 9): the scheduler's loop - nhd/NHDScheduler.py:425-437, FindNode then commit,
 * pod after pod - takes the FIRST feasible node of the cluster (nhd/Matcher.py:393-421), so with the node axis sharded a pod
 * reaches shard s only if the shards before it turned it down at its turn: the pods a shard could not place travel to the next
 * rank (nhd_amd/sharding.py).  One call = one step of that pipeline: `send_bytes` from `send_buf` go to rank `dst` while
 * `recv_bytes` from rank `src` arrive in `recv_buf` - ONE grouped ncclSend / ncclRecv on the context's communicator and reduce
 * stream over xGMI, host buffers staged through device memory; blocks until both are done.  Either side may be absent
 * (dst < 0 / src < 0); dst == src == own rank is a self-exchange.  Every rank of the communicator must make the matching
 * call.  Without a communicator (a single GPU) only the self-exchange is possible and is a copy. */
int nhdfit_comm_sendrecv(nhdfit_ctx* ctx, const void* send_buf, size_t send_bytes, int dst, void* recv_buf, size_t recv_bytes, int src);
/* Element-wise sum of `bytes` uint8 over the ranks, in place (ncclAllReduce(ncclUint8, ncclSum)): how the per-pod results of
 * mode B across shards meet - every pod is placed by at most one rank, all others hold zeros.  A no-op without a communicator. */
int nhdfit_comm_allreduce_sum_u8(nhdfit_ctx* ctx, void* buf, size_t bytes);
/* rank and size of the context's communicator (0 of 1 without one) */
int nhdfit_comm_rank(nhdfit_ctx* ctx, int* rank, int* nranks);

/* One process, several GPUs (the reference calls FindNode from its one scheduler thread, nhd/NHDScheduler.py:43,277):
 * a group owns one context per device; the caller shards the node axis contiguously over them (nhdfit_group_ctx(g, k) +
 * nhdfit_reserve_nodes(global_base) / nhdfit_upload_nodes / nhdfit_set_dictionary per shard, as for a single context).
 * nhdfit_group_find replicates the requests, runs digest + fit on every device, max-reduces the P packed scores with ONE
 * all-reduce over xGMI (ncclCommInitAll communicators, grouped ncclAllReduce(ncclUint64, ncclMax)), lets the owner of
 * each winner map it and returns scores, mappings and owners.  cand: NULL or one [chunks] mask (or NULL) per shard. */
typedef struct nhdfit_group nhdfit_group;
int  nhdfit_group_create(const int* devices, int n, nhdfit_group** out);
void nhdfit_group_destroy(nhdfit_group* g);
int  nhdfit_group_size(nhdfit_group* g);
nhdfit_ctx* nhdfit_group_ctx(nhdfit_group* g, int k);
const char* nhdfit_group_last_error(nhdfit_group* g);
int  nhdfit_group_find(nhdfit_group* g, const nhdfit_req* reqs, uint32_t P, double now, const uint64_t* const* cand,
                       uint64_t* score_out, nhdfit_mapping* map_out, int32_t* owner_out);

/* Skip the feasibility-bitmap store / the winner-mapping kernel (both on by default). */
int nhdfit_set_outputs(nhdfit_ctx* ctx, int want_bitmap, int want_map);

int nhdfit_get_stats(nhdfit_ctx* ctx, nhdfit_stats* out);
int nhdfit_reset_stats(nhdfit_ctx* ctx);

/* ---- request digest straight from the wire format (host code, no GPU needed) -------------------------
 * The pod's Triad libconfig text -> nhdfit_req, replacing TriadCfgParser(text, False).CfgToTopology(False)
 * (nhd/TriadCfgParser.py:337-380, called from nhd/NHDScheduler.py:262-270) followed by the getters FindNode
 * applies to the resulting CfgTopology (nhd/CfgTopology.py:199-232).  NHDFIT_RF_INITIAL_FILTER / `groups` (the pod's NHD_GROUP
 * annotation, nhd/K8SMgr.py:152-165) are not part of the text: the caller fills them in (`flags` comes back holding
 * NHDFIT_RF_NIC_SPLIT or nothing).
 * Returns 0, or one of the codes below with a message in err[errlen]. */
#define NHDFIT_WIRE_NONE   1   /* the reference's CfgToTopology returns None for this text (pod not scheduled) */
#define NHDFIT_WIRE_RAISE  2   /* the reference would raise (malformed text, value of the wrong type)           */
#define NHDFIT_WIRE_LIMIT  3   /* more proc groups than the record holds (4 / 8) or 255 cores in a group        */
int nhdfit_digest_triad_config(const char* text, size_t len, nhdfit_req* out, char* err, size_t errlen);
/* the same text into a nhdfit_big_req: a pod with up to NHDFIT_BIG_MAX_GROUPS processing groups (what NHDFIT_WIRE_LIMIT above turns away) */
int nhdfit_digest_triad_config_big(const char* text, size_t len, nhdfit_big_req* out, char* err, size_t errlen);
/* n texts in one call: codes[i] as above, out[i] zeroed unless codes[i] == 0; returns how many codes are non-zero */
int nhdfit_digest_triad_configs(const char* const* texts, const size_t* lens, uint32_t n, nhdfit_req* out, int32_t* codes);

#ifdef __cplusplus
}
#endif
#endif /* NHDFIT_H */
