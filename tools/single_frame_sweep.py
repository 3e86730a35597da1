#!/usr/bin/env python3
"""One frame per call (the reference's own use, AO.cs:329-347): back-to-back latency and per-pass times for every
combination of the launch-structure thresholds (meao_debug_set), to tune them per frame size.
    python tools/single_frame_sweep.py [--workload 4k]"""
import argparse, itertools, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from miniengineao_amd import AmbientOcclusion, _lib as L
from bench import WORKLOADS, make_frame

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="4k")
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--pipelined", action="store_true")
a = ap.parse_args()
w, h, kind, cam, intensity, ao_format, _ = WORKLOADS[a.workload]
dev = torch.device("cuda", 0)
d = torch.from_numpy(make_frame(kind, w, h, 7)).to(dev)
out = torch.empty((h, w), dtype=torch.uint8 if ao_format == L.AO_R8 else torch.int16, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
BIG = 1 << 30
grid = {"render_small": (0, BIG), "final_small": (0, BIG), "ds_small": (0, BIG), "blend": ("separate", "two_level", "three_level")}
rows = []
for rs, fs, ds, bl in itertools.product(*grid.values()):
    ao = AmbientOcclusion(w, h, num_levels=4, ao_format=ao_format, max_batch=1, near_clip=cam.near, far_clip=cam.far,
                          projection00=cam.proj00(w, h), reversed_z=cam.reversed_z, pipelined=a.pipelined)
    ao.intensity = intensity
    ao.debug_set(L.DEBUG_RENDER_SMALL_MAX_TILES, rs)
    ao.debug_set(L.DEBUG_FINAL_SMALL_MAX_TILES, fs)
    ao.debug_set(L.DEBUG_DS_SMALL_MAX_TILES, ds)
    if bl == "separate":
        ao.debug_set(L.DEBUG_FUSE_COARSE_BLEND, 0)
    else:
        ao.debug_set(L.DEBUG_NESTED_MAX_TILES, 0 if bl == "two_level" else BIG)
    def call():
        if a.pipelined:
            ao.prefetch_device([d.data_ptr()])
        ao.execute_device([d.data_ptr()], [out.data_ptr()], st)
    for _ in range(20): call()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(a.iters): call()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / a.iters * 1e6)
    ao.set_profiling(True)
    for _ in range(50): call()
    ms, n = ao.pass_times_ms()
    ao.close()
    rows.append({"render_small": rs > 0, "final_small": fs > 0, "ds_small": ds > 0, "blend": bl, "us_per_frame": round(best, 2),
                 "pass_us": {nm[:12]: round(ms[k] * 1e3, 1) for k, nm in enumerate(L.PASS_NAMES) if ms[k] > 0}})
rows.sort(key=lambda r: r["us_per_frame"])
for r in rows:
    print(json.dumps(r))
