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


def test_numa_binding_plan_without_a_gpu():
    """VERDICT r5 #6a: the rank -> device -> NUMA node -> CPU map is a pure function of (device placement, allowed mask)."""
    from miniengineao_amd import topology as T
    assert T.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11} and T.parse_cpulist("") == set()
    place = {"device": 3, "numa_node": 1, "cpulist": "64-127,192-255"}
    plan = T.plan_binding(place, allowed=set(range(256)))
    assert plan["action"] == "bind" and plan["bind_to"][0] == 64 and len(plan["bind_to"]) == 128 and plan["node_cpus"] == 128
    # a cgroup / taskset mask is respected: only the allowed CPUs of the node; none of them allowed -> the mask is left alone
    assert T.plan_binding(place, allowed={0, 1, 70, 71})["bind_to"] == [70, 71]
    assert T.plan_binding(place, allowed={0, 1})["action"].startswith("leave the mask")
    assert T.plan_binding({"device": 0, "numa_node": -1, "cpulist": ""}, allowed={0, 1})["action"].startswith("leave the mask")
    assert T.plan_binding(place, allowed={64, 65})["action"] == "already inside the node"


def test_dry_run_topology_prints_the_rank_map_and_launches_nothing():
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run-topology"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["ranks"] == 8 and len(line["map"]) == 8 and [m["rank"] for m in line["map"]] == list(range(8))
    if line["visible_devices"] == 0:
        assert all(m["action"] == "no device visible" for m in line["map"])
    else:       # on a GPU box: every rank has a device and a decision
        assert all(m["device"] == m["rank"] % line["visible_devices"] and m["action"] for m in line["map"])
