"""bench.py's control flow (argument handling, JSON contract, CPU-baseline leg with its parity assert) on CPU: the
Engine is replaced by the host harness, so the numbers mean nothing - the keys and the parity check do."""
import json
import os
import sys

from tests import harness

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Stats:
    fit_ms_total, launches, bytes_last, digest_ms_last, step_ms_last, nsig, lds_bytes, pipes, small_finds = 1.0, 1, 1000, 0.1, 0.2, 1, 1, 2, 0


class DryEngine(harness.HarnessEngine):
    def stage(self, reqs): self._reqs = reqs
    def enqueue(self, now): self._now = now
    def sync(self): pass
    def reset_stats(self): pass
    def set_outputs(self, bitmap=True, mapping=True): pass
    def stats(self): return _Stats()

    def fetch(self, want_bitmap=False, want_map=True):
        return self.find(self._reqs, self._now, want_bitmap=want_bitmap, want_map=want_map)


_SINGLE_SCRIPT = r"""
import runpy, sys
sys.path.insert(0, {root!r})
from tests.test_bench_dryrun import DryEngine
import nhd_amd.engine as eng_mod
eng_mod.Engine = DryEngine
# the other BASELINE shapes of the `other_configs` leg (up to 32 768 nodes x 16 384 pods) are cut down for the host twin: the leg's
# control flow and its in-run parity assert are what is exercised here, not its sizes
from workload import synth
_mc, _mp = synth.make_cluster, synth.make_pods
synth.make_cluster = lambda cfg, n_nodes=None, **kw: _mc(cfg, n_nodes=min(n_nodes, 1024) if n_nodes else n_nodes, **kw)
synth.make_pods = lambda cfg, n_pods=None, **kw: _mp(cfg, n_pods=min(n_pods, 96) if n_pods else n_pods, **kw)
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--nodes-per-gpu", "1024", "--pods", "96", "--cpu-sample-pods", "32", "--no-pmc"]
print("library chatter that must not reach the result stream", file=sys.stderr)
runpy.run_path({bench!r}, run_name="__main__")
"""


def test_bench_json_contract(tmp_path):
    import subprocess
    script = tmp_path / "single.py"
    script.write_text(_SINGLE_SCRIPT.format(root=ROOT, bench=os.path.join(ROOT, "bench.py")))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1                                   # exactly one JSON line
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["vs_baseline"] is None and out["data"] == "synthetic" and "workload" in out["config"]
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
    assert out["mode_b"]["decisions_per_s"] > 0 and out["mode_b"]["placed"] > 0       # decisions under commit semantics
    assert out["end_to_end"]["evals_per_s"] > 0 and out["single_find"]["ms_per_call_median"] > 0
    assert out["score_only"]["evals_per_s"] > 0 and out["deltas"]["mirror_restored"] and out["deltas"]["deltas_per_s"] > 0
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["limiter"] in ("hbm", "lds", "valu", "latency") and out["roofline"]["priced_against"] == "hbm"
    assert "limited_by" in out["roofline"] and "traffic_source" in out["roofline"]
    par = out["mode_b"]["parity"]
    assert par["identical"] and par["pods_checked"] == 96 and out["mode_b"]["commits_that_would_raise"] == 0
    assert out["repeats"]["n"] == 5 and out["cpu_baseline"]["python_restatement"]["value"] > 0
    sl = out["sched_loop"]                                    # row f4 through the drop-in class: three ways to serve a pending list, same decisions and ids
    assert "error" not in sl, sl
    assert sl["identical_decisions_and_ids"] and sl["placed"] > 0 and sl["batched"]["pods_per_s"] > 0 and sl["pod_by_pod_kernel_filter"]["pods_per_s"] > 0
    rf = out["roofline"]                                      # flat scalars (the driver's record keeps scalars only)
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["frac"] == rf["frac_kernel"] and "frac_wall" in rf
    for key in ("hbm_counter_frac", "lds_frac", "valu_issue_frac", "wait_frac", "bank_conflict_share"):
        assert key in rf, key
    big = out["big_pod_find"]                                 # pods beyond the table pass: the leg runs, its in-run parity holds
    assert "error" not in big and big["parity"]["identical"] and big["parity"]["pods"] > 0 and big["calls"] > 0


_RANK_SCRIPT = r"""
import runpy, sys
sys.path.insert(0, {root!r})
from tests.test_bench_dryrun import DryEngine
import nhd_amd.engine as eng_mod

class ShardedDry(DryEngine):
    # the communicator of the host twin: the C-ABI's rank-to-rank entry points (RCCL on the device) over gloo
    def unique_id(self): return b"\0" * 128
    def comm_init(self, nranks, rank, uid):
        from workload.dist import TorchTransport
        self.nranks = nranks; self._tt = TorchTransport()
    def comm_destroy(self): pass
    def comm_rank(self): return self._tt.rank, self._tt.world
    def comm_sendrecv(self, send, dst, recv, src): self._tt.sendrecv(send, dst, recv, src)
    def comm_allreduce_sum_u8(self, buf): self._tt.allreduce_sum_u8(buf)

eng_mod.Engine = ShardedDry
sys.argv = ["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--nodes-per-gpu", "512", "--pods", "40"]
runpy.run_path({bench!r}, run_name="__main__")
"""


def test_bench_two_ranks_control_plane(tmp_path):
    """`--gpus 2` as the driver launches it (one process per rank, RANK / WORLD_SIZE / MASTER_* in the env, gloo
    control plane): rank 0 prints the one JSON line with n_gpus = 2 and both shards' nodes, rank 1 prints nothing."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT.format(root=ROOT, bench=os.path.join(ROOT, "bench.py")))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    lines0 = [ln for ln in outs[0][0].splitlines() if ln.strip()]
    assert len(lines0) == 1 and not outs[1][0].strip()
    out = json.loads(lines0[0])
    assert out["n_gpus"] == 2 and out["config"]["nodes_total"] == 1024 and "cpu_baseline" not in out
    leg = out["strong_scaling"][0]                            # BASELINE config 4's own shape rides along at every N > 1
    assert leg["config"] == 4 and leg["nodes_total"] == 65536 and leg["nodes_per_gpu"] == 32768 and leg["pods"] == 4096 and leg["placed_pods"] > 0
    mb = leg["mode_b"]                                        # ... with BASELINE's own unit: decisions/s under commit semantics over the shards,
    assert "error" not in mb, mb                              # decided again by the independent oracle over the whole cluster
    assert mb["n_gpus"] == 2 and mb["decisions_per_s"] > 0 and mb["placed"] > 0
    assert mb["parity"]["identical"] and mb["parity"]["pods_checked"] == 4096, mb["parity"]


_STRONG_SCRIPT = _RANK_SCRIPT.replace('"--nodes-per-gpu", "512", "--pods", "40"', '"--config", "5", "--total-nodes", "1500", "--pods", "70"')


def test_bench_strong_scaling_shape(tmp_path):
    """BASELINE.json's own multi-GPU shapes are strong scaling (fixed cluster, node axis cut over the ranks):
    `--total-nodes` at two ranks - uneven last shard included."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "strong.py"
    script.write_text(_STRONG_SCRIPT.format(root=ROOT, bench=os.path.join(ROOT, "bench.py")))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    out = json.loads([ln for ln in outs[0][0].splitlines() if ln.strip()][0])
    assert out["scaling"] == "strong" and out["n_gpus"] == 2
    assert out["config"]["nodes_total"] == 1500 and out["config"]["nodes_per_gpu"] == 750


_FOUR_SCRIPT = _RANK_SCRIPT.replace('"--gpus", "2"', '"--gpus", "4"').replace('"--nodes-per-gpu", "512", "--pods", "40"', '"--config", "4", "--total-nodes", "1200", "--pods", "48", "--no-extras"')


def test_bench_four_ranks_strong_shape(tmp_path):
    """`bench.py --gpus 4 --config 4 --total-nodes N` - BASELINE config 4's own multi-GPU shape - at FOUR ranks on the host twin
    (the first 8-GPU box must meet no untested rank count): one JSON line from rank 0, strong scaling,
    the node axis cut four ways, every pod placed somewhere."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "four.py"
    script.write_text(_FOUR_SCRIPT.format(root=ROOT, bench=os.path.join(ROOT, "bench.py")))
    procs = []
    for rank in range(4):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all(not o[0].strip() for o in outs[1:])
    out = json.loads([ln for ln in outs[0][0].splitlines() if ln.strip()][0])
    assert out["scaling"] == "strong" and out["n_gpus"] == 4 and out["config"]["nodes_total"] == 1200 and out["config"]["nodes_per_gpu"] == 300
    assert out["placed_pods"] > 0
