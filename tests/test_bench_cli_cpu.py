"""bench.py's launch surface, on CPU: `python bench.py --gpus N` must start its own N ranks when no launcher did (the driver's
single-GPU command line has no torchrun in front), and say plainly when the box has fewer than N devices."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=e, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_without_launcher_reports_missing_devices_not_a_launcher_error():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box has two devices: the real run is tests/test_dist_gpu.py::test_bench_two_ranks_over_rccl")
    r = _run("--gpus", "2")
    assert r.returncode != 0
    msg = r.stderr + r.stdout
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    assert f"needs 2 HIP devices, found {have}" in msg, msg[-2000:]
    assert "torch.distributed.run" not in msg and "WORLD_SIZE" not in msg


def test_self_launch_starts_the_ranks_and_rank0_prints_one_json_line():
    """--launch-check runs the self-launch path end to end (re-exec under torch.distributed.run, rendezvous on 127.0.0.1, one
    collective over gloo, one JSON line from rank 0) without any GPU work."""
    r = _run("--gpus", "2", "--launch-check")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j == {"launch_check": True, "world": 2, "n_gpus": 2, "sum_of_ranks_plus_1": 3.0}


def test_launcher_form_still_works():
    """The driver's multi-GPU form: torch.distributed.run in front, --gpus N behind."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "3", "--launch-check"],
                       env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["world"] == 3 and j["sum_of_ranks_plus_1"] == 6.0
