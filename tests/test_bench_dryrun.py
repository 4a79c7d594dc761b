"""bench.py's control flow (argument handling, JSON contract, CPU-baseline leg with its parity assert) on CPU: the
Engine is replaced by the host harness, so the numbers mean nothing - the keys and the parity check do."""
import io
import json
import os
import runpy
import sys
from contextlib import redirect_stdout

from tests import harness

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Stats:
    fit_ms_total, launches, bytes_last, digest_ms_last, step_ms_last, nsig, lds_bytes = 1.0, 1, 1000, 0.1, 0.2, 1, 1


class DryEngine(harness.HarnessEngine):
    def stage(self, reqs): self._reqs = reqs
    def enqueue(self, now): self._now = now
    def sync(self): pass
    def reset_stats(self): pass
    def stats(self): return _Stats()

    def fetch(self, want_bitmap=False, want_map=True):
        return self.find(self._reqs, self._now, want_bitmap=want_bitmap, want_map=want_map)


def test_bench_json_contract(monkeypatch):
    import nhd_amd.engine as eng_mod
    monkeypatch.setattr(eng_mod, "Engine", DryEngine)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "3", "--warmup", "1", "--nodes-per-gpu", "1024",
                                      "--pods", "96", "--cpu-sample-pods", "32"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    buf = io.StringIO()
    with redirect_stdout(buf):
        runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
    lines = [ln for ln in buf.getvalue().splitlines() if ln.strip()]
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
