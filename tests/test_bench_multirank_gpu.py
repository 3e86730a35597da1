"""bench.py's N > 1 path end to end on ONE GPU: two ranks launched the way the driver launches them
(torch.distributed.run), gloo process group so that both ranks may share device 0 (RCCL refuses two ranks on one
GPU).  Everything but the collective transport is the code an 8-GPU node runs: frame sharding, the fences around the
timed region, the MAX over ranks, the per-rank step times, the checksum gather, validation on every rank."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


BENCH_FLAGS = ["--dist-backend", "gloo", "--workload", "4k", "--steps", "3", "--warmup", "1", "--min-time-ms", "0",
               "--no-cpu-baseline", "--skip-latency", "--no-copy-ceiling"]


def run_bench(cmd):
    env = dict(os.environ, OMP_NUM_THREADS="8")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    proc = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly ONE JSON line"
    return json.loads(lines[0])


def check_two_rank_line(line, batch):
    # two ranks on ONE physical device: n_gpus counts devices, `ranks` the processes
    assert line["ranks"] == 2 and line["n_gpus"] == 1 and line["devices_shared"] is True
    assert line["world_seen_by_process_group"] == 2 and line["world_seen_by_rccl"] is None      # gloo group: no RCCL in it
    assert line["config"]["process_group"] == "gloo" and line["config"]["frames_per_step_per_gpu"] == batch
    assert len(line["per_rank_ms_per_step"]) == 2 and line["scaling"] == "weak"
    assert line["steps"] == 3 and line["steps_requested"] == 3
    v = line["validation"]
    assert v["frames_checksummed"] == 2 * batch and v["distinct_checksums"] == 2 * batch     # frame g -> rank g mod 2
    assert v["frames_vs_oracle"] >= 2 * batch and v["mismatching_frames"] == 0
    assert v["pipelined_equals_plain_all_frames"] is True
    assert line["other_workloads"] is None and line["cpu_baseline"] is None and line["best_host_config"] is None   # N = 1 only
    # whole-job value = all ranks' pixels / the slowest rank's time
    px = 3840 * 2160 * batch * line["steps"] * 2
    assert abs(line["value"] - px / (line["ms_per_step"] * 1e-3 * line["steps"]) / 1e6) / line["value"] < 0.01


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_over_gloo():
    """Launched the way the driver launches N > 1: torch.distributed.run in front of bench.py."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "2"] + BENCH_FLAGS
    check_two_rank_line(run_bench(cmd), 2)


@pytest.mark.gpu
def test_plain_invocation_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher in front (the shape of the driver's only known command): the script
    re-runs itself under torch.distributed.run and rank 0's line is its output."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "2"] + BENCH_FLAGS
    check_two_rank_line(run_bench(cmd), 2)


@pytest.mark.gpu
def test_one_frame_per_rank_is_baseline_config_4():
    """BASELINE config 4's shape: independent 4K frames sharded ONE per rank (--batch 1), frame g on rank g."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "1"] + BENCH_FLAGS
    line = run_bench(cmd)
    check_two_rank_line(line, 1)
    assert line["config"]["sharding"] == "frames x2"


@pytest.mark.gpu
def test_rccl_launch_is_refused_without_enough_devices():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU box runs the RCCL launch itself")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True,
                          text=True, cwd=ROOT, timeout=300)
    assert proc.returncode != 0 and "RCCL needs one GPU per rank" in proc.stderr


@pytest.mark.gpu
def test_rccl_two_ranks_when_two_gpus_are_visible():
    """The real thing, wherever a second GPU exists (skipped on the 1-GPU boxes this project has had so far)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    flags = [f for f in BENCH_FLAGS if f not in ("--dist-backend", "gloo")]
    line = run_bench([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "1"] + flags)
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["devices_shared"] is False and line["world_seen_by_rccl"] == 2
    assert line["validation"]["mismatching_frames"] == 0 and line["validation"]["distinct_checksums"] == 2


# ---- round 5: the 8-GPU command path rehearsed on one GPU (readiness, not a curve: VERDICT r4 next #7) ----------------

@pytest.mark.gpu
def test_one_rank_through_the_launcher_with_a_real_rccl_group():
    """`bench.py --gpus 1 --launcher`: the script re-launches itself under torch.distributed.run exactly as for N > 1, the
    rank creates an RCCL communicator on its device (backend "nccl"), and the fences, the MAX over ranks, the per-rank
    times and the checksum gather of the run go through it.  world_seen_by_rccl == 1 alone would also be true with no
    group at all -- process_group_in_use says which it was."""
    flags = [f for f in BENCH_FLAGS if f not in ("--dist-backend", "gloo")]
    line = run_bench([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--launcher", "--batch", "2",
                      "--no-other-workloads", "--no-best-host-config"] + flags)
    assert line["process_group_in_use"] == {"initialized": True, "backend": "nccl", "world": 1}
    assert line["world_seen_by_rccl"] == 1 and line["n_gpus"] == 1 and line["ranks"] == 1
    assert line["config"]["process_group"] == "nccl"
    assert line["validation"]["mismatching_frames"] == 0 and line["validation"]["frames_checksummed"] == 2


@pytest.mark.gpu
def test_without_a_launcher_there_is_no_group():
    line = run_bench([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "1", "--no-other-workloads",
                      "--no-best-host-config"] + [f for f in BENCH_FLAGS if f not in ("--dist-backend", "gloo")])
    assert line["process_group_in_use"] == {"initialized": False, "backend": None, "world": 1}
    assert line["config"]["process_group"] is None


@pytest.mark.gpu
def test_eight_pool_members_one_frame_each_is_baseline_config_4_in_process():
    """BASELINE config 4 through the in-process host (meao_pool_*): eight members, one 4K frame per member per step, every
    member's frame validated against the oracle.  All members sit on device 0 here (their kernels time-share: the value
    is not a scaling number); on an 8-GPU node the same command puts member m on device m."""
    line = run_bench([sys.executable, os.path.join(ROOT, "bench.py"), "--pool", "8", "--batch", "1", "--workload", "4k",
                      "--steps", "3", "--warmup", "1", "--min-time-ms", "0"])
    assert line["pool_members"] == 8 and line["config"]["frames_per_step_per_member"] == 1
    assert len(line["per_member_ms"]) == 8 and all(ms > 0 for ms in line["per_member_ms"])
    v = line["validation"]
    assert v["frames_checksummed"] == 8 and v["distinct_checksums"] == 8
    assert v["frames_vs_oracle"] == 8 and v["mismatching_frames"] == 0          # first and last frame of every member = its one frame
    assert len(line["gather_paths"]) == 8
