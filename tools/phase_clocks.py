#!/usr/bin/env python3
"""Where an upsample workgroup's time goes (diagnostic build -DMEAO_X_PHASE_CLOCKS=1):
    python tools/build_variants.py clocks=-DMEAO_X_PHASE_CLOCKS=1
    MEAO_LIB_PATH=$PWD/miniengineao_amd/lib/variants/libmeao_clocks.so python tools/phase_clocks.py [--workload 4k] [--pipeline]
Every wave of every upsample tile stamps s_memrealtime (100 MHz) at its phase boundaries; the table is the mean
time per wave and phase over all upsample launches of the timed steps (all four passes pooled: run with
--only-final to restrict the launches to the full-resolution pass by giving the blend passes a separate read-out)."""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from miniengineao_amd import AmbientOcclusion, _lib
from bench import WORKLOADS, make_frame, default_batch
from miniengineao_amd.sharding import frame_seed

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="4k")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--pipeline", action="store_true")
a = ap.parse_args()
lib = _lib.load()
read = lib.meao_x_phase_clocks            # AttributeError: not a -DMEAO_X_PHASE_CLOCKS=1 build
read.restype, read.argtypes = C.c_int, [C.POINTER(C.c_uint64 * 64)]
w, h, kind, cam, intensity, ao_format, _ = WORKLOADS[a.workload]
B = default_batch(w, h)
dev = torch.device("cuda", 0)
frames = [make_frame(kind, w, h, frame_seed(0x1234ABCD, f)) for f in range(min(B, 4))]
dd = [torch.from_numpy(frames[f % len(frames)]).to(dev) for f in range(B)]
out = [torch.empty((h, w), dtype=torch.uint8 if ao_format == _lib.AO_R8 else torch.int16, device=dev) for _ in range(B)]
ao = AmbientOcclusion(w, h, num_levels=4, ao_format=ao_format, max_batch=B, near_clip=cam.near, far_clip=cam.far,
                      projection00=cam.proj00(w, h), reversed_z=cam.reversed_z, pipelined=a.pipeline)
ao.intensity = intensity
dp, op = [t.data_ptr() for t in dd], [t.data_ptr() for t in out]
st = torch.cuda.current_stream(dev).cuda_stream
def step():
    if a.pipeline:
        ao.prefetch_device(dp)
    ao.execute_device(dp, op, st)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.08:
    for _ in range(8): step()
    torch.cuda.synchronize()
buf = (C.c_uint64 * 64)()
assert read(C.byref(buf)) == 0            # clear
ao.set_profiling(True)
for _ in range(a.steps): step()
torch.cuda.synchronize()
assert read(C.byref(buf)) == 0
ms, n = ao.pass_times_ms()
names = ["0 window load+fill", "1 barrier", "2 H-blur", "3 barrier", "4 V-blur", "5 barrier(+carried loads)", "6 bilateral pass 0", "7 bilateral pass 1"]
res = {"workload": a.workload, "pipeline": a.pipeline, "pass_us": {nm: round(ms[k] * 1e3, 1) for k, nm in enumerate(_lib.PASS_NAMES) if ms[k] > 0}}
names16 = ["0 loop top (decode next, ...)", "1 the tile", "2 end-of-tile barrier"] + ["-"] * 5
names24 = ["0 window load+fill", "1 barrier", "2 iteration 0", "3 iteration 1", "4 iteration 2", "5 iteration 3", "-", "-"]
for base, label in ((0, "full_resolution_pass"), (8, "blend_passes"), (24, "render")):
    tot, tab = 0.0, {}
    for p, nm in enumerate(names16 if base == 16 else names24 if base == 24 else names):
        cnt = buf[32 + base + p]
        if cnt:
            us = buf[base + p] / cnt / 100.0
            tot += us
            tab[nm] = {"us_per_wave": round(us, 3), "waves": int(cnt)}
    tab["sum_us_per_wave"] = round(tot, 3)
    res[label] = tab
print(json.dumps(res, indent=1))
