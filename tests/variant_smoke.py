"""Parity smoke of ONE build of libmeao_hip.so (MEAO_LIB_PATH selects it): a few sizes through the plain and the
pipelined path, every debug-visible buffer against the oracle.  Run by tests/test_variants_gpu.py once per variant
library under miniengineao_amd/lib/variants/, so that no experimental -D arm rots unseen."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from miniengineao_amd import synth
from oracle import oracle as O
from tests import helpers as H

O.build()
dev = torch.device("cuda", 0)
bad = 0
for (w, h, n, kw) in ((322, 182, 2, {}), (1280, 720, 3, {}), (515, 301, 2, dict(ao_format=1)), (2048, 1152, 4, {})):
    s = H.settings(O, w, h, **kw)
    seqs = [[synth.make("S2", w, h, seed=10 * k + f) for f in range(n)] for k in range(3)]
    seqs[1][n - 1] = H.hostile_frame(w, h, 99, density=0.001)
    want = [[O.run(f, s) for f in b] for b in seqs]
    dd = [[torch.from_numpy(f).to(dev) for f in b] for b in seqs]
    dt = torch.uint8 if s.ao_format == 0 else torch.int16
    out = [[torch.empty((h, w), dtype=dt, device=dev) for _ in b] for b in seqs]
    st = torch.cuda.current_stream(dev).cuda_stream
    for pipelined in (False, True):
        ao = H.component(s, max_batch=n, pipelined=pipelined)
        for k in range(3):
            if pipelined and k + 1 < 3:
                ao.prefetch_device([t.data_ptr() for t in dd[k + 1]])
            ao.execute_device([t.data_ptr() for t in dd[k]], [t.data_ptr() for t in out[k]], st)
            torch.cuda.synchronize(dev)
            for f in range(n):
                ok, diff = H.nan_aware_equal(out[k][f].cpu().numpy().view(want[k][f]["result"].dtype), want[k][f]["result"])
                if not ok:
                    bad += 1
                    print("MISMATCH result", w, h, pipelined, k, f, int(diff.sum()))
                for i in H.valid_debug_ids(s.num_levels, s.hq_levels):
                    ok, diff = H.nan_aware_equal(ao.debug_buffer(i, frame=f), want[k][f][H.NAMES[i]])
                    if not ok:
                        bad += 1
                        print("MISMATCH", H.NAMES[i], w, h, pipelined, k, f, int(diff.sum()))
        ao.close()
print("variant", os.environ.get("MEAO_LIB_PATH", "product"), "ok" if bad == 0 else f"{bad} mismatches")
sys.exit(1 if bad else 0)
