"""Worker of tests/test_sharding_gloo.py: one process per rank (gloo on CPU).  Mirrors what
bench.py does on N GPUs: shard frames round-robin, process the local frames, fence, take the
max-over-ranks of the timed region, gather per-frame checksums.  The per-frame work here is
the CPU oracle (test stand-in for the device path, which needs a GPU)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from miniengineao_amd import distributed as mdist  # noqa: E402
from miniengineao_amd import synth  # noqa: E402
from miniengineao_amd.sharding import frame_seed, frames_for_rank  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import helpers as H  # noqa: E402


def main():
    num_frames, w, h = int(sys.argv[1]), 96, 54
    rank, world, _ = mdist.init("gloo")
    mine = frames_for_rank(num_frames, rank, world)
    s = H.settings(O, w, h)
    mdist.fence(torch.device("cpu"))
    t0 = time.perf_counter()
    sums = []
    for f in mine:
        depth = synth.make("S2", w, h, seed=frame_seed(0x1234ABCD, f))
        sums.append(H.checksum(O.run(depth, s, result_only=True)["result"]) & 0x7FFFFFFFFFFFFFFF)
    mdist.fence(torch.device("cpu"))
    local = time.perf_counter() - t0 + 0.001 * rank
    slowest = mdist.max_over_ranks(local)
    gathered = mdist.gather_checksums(sums)
    per_rank = mdist.gather_floats(local)
    if rank == 0:
        print(json.dumps({"world": world, "frames": [frames_for_rank(num_frames, r, world) for r in range(world)],
                          "checksums": gathered, "slowest": slowest, "local0": local, "per_rank": per_rank,
                          "world_seen": mdist.world_size()}))
    mdist.shutdown()


if __name__ == "__main__":
    main()
