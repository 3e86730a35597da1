"""One-off large-frame check (not collected by pytest): hostile and clean 4K / 1080p frames through the
pipelined path (split carried downsample, window-first loads, two-level launch), every output vs the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from miniengineao_amd import synth
from oracle import oracle as O
from tests import helpers as H

bad = 0
for (w, h, n) in ((3840, 2160, 3), (1920, 1080, 2), (2048, 1024, 4)):
    s = H.settings(O, w, h)
    dev = torch.device("cuda", 0)
    seqs = [[synth.make("S2", w, h, seed=10 * k + f) for f in range(n)] for k in range(3)]
    seqs[1][n - 1] = H.hostile_frame(w, h, 99, density=0.0005)
    seqs[2][0] = H.hostile_frame(w, h, 98, density=0.0005)
    dd = [[torch.from_numpy(f).to(dev) for f in b] for b in seqs]
    out = [[torch.empty((h, w), dtype=torch.uint8, device=dev) for _ in b] for b in seqs]
    ao = H.component(s, max_batch=n, pipelined=True)
    st = torch.cuda.current_stream(dev).cuda_stream
    for k in range(3):
        if k + 1 < 3:
            ao.prefetch_device([t.data_ptr() for t in dd[k + 1]])
        ao.execute_device([t.data_ptr() for t in dd[k]], [t.data_ptr() for t in out[k]], st)
    torch.cuda.synchronize(dev)
    for k in range(3):
        for f in range(n):
            want = O.run(seqs[k][f], s, result_only=True)["result"]
            ok, diff = H.nan_aware_equal(out[k][f].cpu().numpy(), want)
            if not ok:
                bad += 1
                print("MISMATCH", w, h, k, f, int(diff.sum()))
    ao.close()
print("large frames:", "all equal" if bad == 0 else f"{bad} mismatching")
