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


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_over_gloo():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
           "--workload", "4k", "--batch", "2", "--steps", "3", "--warmup", "1", "--min-time-ms", "0",
           "--no-cpu-baseline", "--skip-latency"]
    env = dict(os.environ, OMP_NUM_THREADS="8")
    proc = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    line = json.loads([l for l in proc.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["world_seen_by_rccl"] == 2 and line["config"]["process_group"] == "gloo"
    assert len(line["per_rank_ms_per_step"]) == 2 and line["scaling"] == "weak"
    v = line["validation"]
    assert v["frames_checksummed"] == 4 and v["distinct_checksums"] == 4          # 2 frames per rank, frame g -> rank g mod 2
    assert v["frames_vs_oracle"] >= 4 and v["mismatching_frames"] == 0
    assert v["pipelined_equals_plain_all_frames"] is True
    assert line["other_workloads"] is None and line["cpu_baseline"] is None      # N = 1 only
    # whole-job value = all ranks' pixels / the slowest rank's time
    px = 3840 * 2160 * 2 * line["steps"] * 2
    assert abs(line["value"] - px / (line["ms_per_step"] * 1e-3 * line["steps"]) / 1e6) / line["value"] < 0.01
