"""bench.py's launcher logic without a GPU: `python bench.py --gpus N` with no launcher in front re-runs itself under
torch.distributed.run exactly as the driver would launch it, and refuses an RCCL launch when fewer devices
than ranks are visible."""
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_plain_gpus_n_relaunches_itself_under_torch_distributed_run(monkeypatch):
    import torch
    import bench
    calls = []
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(subprocess, "run", lambda cmd, env=None: calls.append((cmd, env)) or types.SimpleNamespace(returncode=0))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert bench.self_launch(types.SimpleNamespace(gpus=8, dist_backend="nccl")) == 0
    (cmd, env), = calls
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]          # the same command line, per rank
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and int(env["OMP_NUM_THREADS"]) >= 1


def test_rccl_launch_needs_one_device_per_rank(monkeypatch):
    import torch
    import bench
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: pytest.fail("must not launch"))
    with pytest.raises(SystemExit) as e:
        bench.self_launch(types.SimpleNamespace(gpus=8, dist_backend="nccl"))
    assert "RCCL needs one GPU per rank" in str(e.value)
    # gloo lets ranks share devices (functional check): the launch goes ahead
    calls = []
    monkeypatch.setattr(subprocess, "run", lambda cmd, env=None: calls.append(cmd) or types.SimpleNamespace(returncode=0))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--dist-backend", "gloo"])
    assert bench.self_launch(types.SimpleNamespace(gpus=2, dist_backend="gloo")) == 0 and "--nproc-per-node=2" in calls[0]


def test_an_explicit_steps_is_timed_exactly():
    """`--steps K` (the driver's command line) times exactly K steps; only the flag-less default
    stretches the timed region to 100 ms; an explicit --min-time-ms always applies."""
    import bench
    assert bench.resolve_timed_region(20, None) == (20, 0.0)
    assert bench.resolve_timed_region(None, None) == (30, 100.0)
    assert bench.resolve_timed_region(20, 100.0) == (20, 100.0)
    assert bench.resolve_timed_region(None, 0.0) == (30, 0.0)
