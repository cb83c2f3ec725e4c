"""GPU: bench.py keeps its output contract — one JSON line on stdout with the metric, the roofline object and the CPU baseline —
checked on the small 'tiny' workload (the default workload is BASELINE.json's cfg3)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_emits_one_json_line_with_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "SGNS training edges/sec" and d["unit"] == "edges/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["value"] > 0 and "workload" in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == "edges/s" and c["sample"]


def test_bench_two_ranks_on_one_gpu():
    """`bench.py --gpus 2` with no torchrun environment starts its two ranks itself; over gloo they share this box's one card and run the
    block schedule for real (RCCL needs one GPU per rank: that run is the driver's).  n_gpus must equal --gpus."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", "tiny", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "cpu_baseline" not in d
    assert "block schedule x2" in d["config"]["parallelism"]
    # N ranks step in the global batch the product trains with — a tenth of the epoch whatever N (fit_distributed's; the schedule keeps the one-GPU embedding only in
    # batches of at most a fifth of it) — so the total work of a step is fixed: "strong"; the round-1..4 step (N x a tenth of the epoch) is behind --weak-batch
    c = d["config"]
    assert d["scaling"] == "strong" and abs(c["batch_fraction_of_epoch"] - 0.1) < 0.01 and c["global_batch_walks"] == 2 * c["walks_per_step_per_gpu"], c
