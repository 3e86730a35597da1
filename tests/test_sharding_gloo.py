"""Multi-rank path on CPU: world_size 2 (and 3) over gloo, launched the way the driver launches
bench.py.  Frames shard round-robin with no data-path collective; only the barrier, the
max-over-ranks and the checksum gather are collective."""
import json
import os
import subprocess
import sys

import pytest

from miniengineao_amd import synth
from miniengineao_amd.sharding import frame_checksum, frame_seed, frames_for_rank, owner_of_frame
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_is_a_disjoint_cover():
    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 8, 9, 64):
            parts = [frames_for_rank(n, r, world) for r in range(world)]
            flat = sorted(f for p in parts for f in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert frames_for_rank(8, 3, 8) == [3]           # BASELINE config 4: one 4K frame per GPU
    with pytest.raises(ValueError):
        frames_for_rank(4, 2, 2)
    for world in (1, 2, 8):
        for f in range(20):
            assert f in frames_for_rank(20, owner_of_frame(f, world), world)


def test_frame_checksum_is_order_sensitive_and_63_bit():
    import numpy as np
    a = np.arange(4096, dtype=np.uint8).reshape(64, 64)
    b = a.copy(); b[3, 5], b[3, 6] = a[3, 6], a[3, 5]
    assert frame_checksum(a) != frame_checksum(b) and 0 <= frame_checksum(a) < 2 ** 63
    assert frame_checksum(a) == H.checksum(a) & 0x7FFFFFFFFFFFFFFF


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_batch_over_gloo(oracle, world):
    from miniengineao_amd.distributed import free_port
    port = free_port()
    num_frames = 5
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_worker.py"), str(num_frames)]
    proc = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = [l for l in proc.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["world"] == world
    assert res["frames"] == [frames_for_rank(num_frames, r, world) for r in range(world)]
    assert res["slowest"] >= res["local0"]           # max over ranks
    assert res["world_seen"] == world and len(res["per_rank"]) == world
    assert abs(max(res["per_rank"]) - res["slowest"]) < 1e-9 and abs(res["per_rank"][0] - res["local0"]) < 1e-9
    s = H.settings(oracle, 96, 54)
    for r in range(world):
        for k, f in enumerate(res["frames"][r]):
            depth = synth.make("S2", 96, 54, seed=frame_seed(0x1234ABCD, f))
            want = H.checksum(oracle.run(depth, s, result_only=True)["result"]) & 0x7FFFFFFFFFFFFFFF
            assert res["checksums"][r][k] == want, (r, f)
