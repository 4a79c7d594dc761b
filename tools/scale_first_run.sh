#!/bin/bash
# The FIRST run on a multi-GPU MI355X node (nothing in this repository has executed RCCL with more than one rank: the build
# pool has one GPU per box).  Boring by design - each step stops at the first failure and says what to look at:
#   1. the library loads, ABI 9, the box shows N devices;
#   2. one rank per GPU, a communicator, the ring's rank-to-rank exchange (nhdfit_comm_sendrecv) and the uint8 sum on 2 ranks;
#   3. mode A over 2 shards: the all-reduce(max) of the packed scores gives the single-GPU winners (small cluster, oracle-checked);
#   4. mode B over 2 shards (nhd_amd.sharding.schedule_batch_sharded over RcclTransport) against the oracle's loop;
#   5. bench.py --gpus N for N in 2 4 8 (as the driver launches it).
#   bash tools/scale_first_run.sh [max_gpus]
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
export MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
MAXG=${1:-8}
python - <<'PY' || exit 1
from nhd_amd import _lib
lib = _lib.load()
n = lib.nhdfit_device_count()
print("step 1: libnhdfit ABI", lib.nhdfit_abi_version(), "devices", n)
assert lib.nhdfit_abi_version() == 9 and n >= 2, "needs ABI 9 and at least two GPUs"
PY
run() { local n=$1; shift; timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 500)) "$@"; }
cat > /tmp/nhd_scale_step.py <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.environ["NHD_ROOT"])
import torch.distributed as dist
from nhd_amd import pack, sharding
from nhd_amd.engine import Engine, winner_index
from workload import planes, refmodel, synth
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)          # control plane only: the unique id, barriers
n, P, cfg = 4096, 300, 4
spec = synth.make_cluster(cfg, n_nodes=n)
pods, groups = synth.make_pods(cfg, n_pods=P)
tops = [refmodel.make_topology(s) for s in pods]
lo, hi = sharding.shard_bounds(n, world, rank)
sub = spec.shard(lo, hi)
pk = pack.Packer(); table = planes.planes_from_spec(pk, sub); reqs = pk.digest_many(tops, groups); pk.close_signatures()
eng = Engine(int(os.environ["LOCAL_RANK"])); eng.set_dictionary(pk); eng.upload(table, global_base=lo)
uid = [eng.unique_id() if rank == 0 else None]; dist.broadcast_object_list(uid, src=0); eng.comm_init(world, rank, uid[0])
assert eng.comm_rank() == (rank, world)
# step 2: the ring's exchange and the byte sum
a = np.full(1000, rank + 1, np.int32); b = np.zeros_like(a)
eng.comm_sendrecv(a, (rank + 1) % world, b, (rank - 1) % world)
assert (b == ((rank - 1) % world) + 1).all(), "sendrecv"
c = np.full(64, rank + 1, np.uint8); eng.comm_allreduce_sum_u8(c); assert (c == world * (world + 1) // 2).all(), "allreduce"
if rank == 0: print("step 2: rank-to-rank exchange and uint8 sum ok on", world, "ranks", flush=True)
# step 3: mode A over the shards
score, _, maps = eng.find(reqs, spec.clock_now, want_bitmap=False, want_map=True)
if rank == 0:
    from oracle import coracle
    win, _ = coracle.Cluster.from_spec(spec).find(coracle.Cluster.from_spec(spec).pods_from_tops(tops, groups), spec.clock_now, want_feas=False, threads=8)
    got = np.array([winner_index(int(s)) if s else -1 for s in score])
    assert (got == win).all(), ("mode A winners differ from the oracle's", np.flatnonzero(got != win)[:5])
    print("step 3: mode A over", world, "shards equals the oracle's winners (", int((win >= 0).sum()), "placed )", flush=True)
# step 4: mode B over the shards
bits = np.zeros(((hi - lo + 63) // 64) * 64, np.uint8); bits[:hi - lo] = (np.asarray(sub.n_gpus) == 0)
nogpu = np.packbits(bits, bitorder="little").view(np.uint64).copy()
node, mp, pl, st = sharding.schedule_batch_sharded(eng, reqs, spec.clock_now, pk, nogpu, sharding.RcclTransport(eng), apply=True, chunk=128)
if rank == 0:
    from oracle import coracle, seq_oracle
    sc = seq_oracle.SeqCluster(coracle.Cluster.from_spec(spec))
    win, omaps, oids, ndef = seq_oracle.schedule_sequence(sc, tops, groups, spec.clock_now)
    assert list(node[:ndef]) == list(win[:ndef]), "mode B decisions differ from the oracle's loop"
    print("step 4: mode B over", world, "shards equals the oracle's loop (", int((node >= 0).sum()), "placed )", flush=True)
dist.barrier(); eng.comm_destroy(); dist.destroy_process_group()
PY
NHD_ROOT=$ROOT run 2 /tmp/nhd_scale_step.py || { echo "steps 2-4 failed on 2 ranks: see the traceback above (RCCL banner: NCCL_DEBUG=INFO)"; exit 1; }
for n in 2 4 8; do
  [ $n -le $MAXG ] || break
  echo "step 5: bench.py --gpus $n"
  run $n bench.py --gpus $n --steps 20 --warmup 5 | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('  n_gpus', d['n_gpus'], 'evals/s %.3g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'strong', [(l.get('config'), l.get('ms_per_step'), (l.get('mode_b') or {}).get('decisions_per_s'), ((l.get('mode_b') or {}).get('parity') or {}).get('identical'), (l.get('mode_b') or {}).get('error')) for l in d.get('strong_scaling', [])])" || { echo "bench.py --gpus $n failed"; exit 1; }
done
echo "scale_first_run: all steps passed"
