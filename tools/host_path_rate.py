"""PCIe-inclusive rate of the host-pointer path (meao_execute with MEAO_MEM_HOST on both sides):
pageable host arrays -> staged H2D copy -> 6 kernels -> D2H copy.  Reported in DESIGN.md section 6
next to the HBM-resident headline; never the bench `value`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from miniengineao_amd import AmbientOcclusion, synth  # noqa: E402

w, h, n = 3840, 2160, 8
cam = synth.DEFAULT_CAMERA
frames = [synth.make("S2", w, h, seed=7 + f) for f in range(n)]
ao = AmbientOcclusion(w, h, max_batch=n, near_clip=cam.near, far_clip=cam.far, projection00=cam.proj00(w, h))
ao.render_batch(frames)
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    ao.render_batch(frames)
dt = (time.perf_counter() - t0) / (reps * n)
print(f"host-pointer path: {dt * 1e3:.3f} ms per 4K frame = {w * h / dt / 1e6:.0f} Mpixels/s "
      f"({(w * h * 5) / dt / 1e9:.1f} GB/s over PCIe, pageable memory)")
