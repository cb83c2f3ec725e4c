"""CPU: `python bench.py --gpus N` starts N ranks itself (the driver's command shape carries no torchrun environment).
The parent must not touch torch or HIP; rank 0 of the children prints the one JSON line with n_gpus == N.  With
--rendezvous-check the ranks only meet (gloo here: no GPU in this container) — the measuring path is covered by the -m gpu
test test_bench_two_ranks_on_one_gpu."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=e)


def test_gpus_2_spawns_two_ranks_and_reports_n_gpus_2():
    out = _run(["--gpus", "2", "--backend", "gloo", "--workload", "tiny", "--rendezvous-check"])
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rendezvous"] == "ok"
    assert "torch.distributed.run" in out.stderr and "--nproc-per-node 2" in out.stderr


def test_gpus_and_world_size_must_agree():
    out = _run(["--gpus", "2", "--rendezvous-check"], env={"WORLD_SIZE": "3", "RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=3" in out.stderr


def test_bad_arguments_fail_before_any_rank_starts():
    out = _run(["--gpus", "2", "--backend", "nonsense", "--rendezvous-check"])
    assert out.returncode != 0 and "torch.distributed.run" not in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.strip().startswith("{")]


def test_child_failure_is_the_parents_exit_status():
    # ranks that cannot reach the hot path (no GPU in this container, or no libdge.so) exit non-zero: the launcher passes that on
    out = _run(["--gpus", "2", "--backend", "gloo", "--workload", "tiny", "--steps", "1", "--warmup", "0"], env={"CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": ""})
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.strip().startswith("{")]


def test_parent_never_imports_torch():
    code = ("import sys, runpy\n"
            "sys.argv = ['bench.py', '--gpus', '2', '--backend', 'gloo', '--rendezvous-check']\n"
            "import subprocess\n"
            "subprocess.run = lambda *a, **k: type('R', (), {'returncode': 0})()\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    assert not e.code, e.code\n"
            "assert 'torch' not in sys.modules, 'the launcher imported torch'\nprint('clean')\n" % os.path.join(ROOT, "bench.py"))
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        e.pop(k, None)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=ROOT, env=e)
    assert out.returncode == 0 and "clean" in out.stdout, out.stderr[-2000:]


def test_rank_without_a_device_leaves_before_the_rendezvous():
    # one rank, no GPU visible: bench.py must say why and exit non-zero before init_process_group (the other ranks of a real launch would
    # otherwise wait for it), and print no JSON line
    out = _run(["--steps", "1", "--warmup", "0", "--workload", "tiny", "--no-cpu-baseline"], env={"CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": ""}, timeout=300)
    assert out.returncode != 0 and "device(s)" in out.stderr and "nothing measured" in out.stderr, out.stderr[-1500:]
    assert not [l for l in out.stdout.splitlines() if l.strip().startswith("{")]


def test_launcher_refuses_to_start_ranks_under_a_profiler():
    out = _run(["--gpus", "2", "--backend", "gloo", "--rendezvous-check"], env={"LD_PRELOAD_PROBE": "", "ROCP_TOOL_LIBRARIES": "/opt/rocm/lib/librocprofiler-sdk-tool.so"})
    assert out.returncode == 2 and "one rank per rocprofv3 process" in out.stderr and "torch.distributed.run" not in out.stderr
