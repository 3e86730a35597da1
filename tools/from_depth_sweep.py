#!/usr/bin/env python3
"""Calls of n frames, back to back, with the render windows filled from the stored depth mips (the round-4 sequence: downsample |
render | blends | final) or from the raw depth frame (MEAO_DEBUG_RENDER_FROM_DEPTH 1: the downsample pass as extra workgroups of the
render launch; 2: both launches on two streams) -- where the call-size threshold of the default (3) belongs, per frame size.
    python tools/from_depth_sweep.py [--workloads 4k,1080p] [--frames 1,2,4,8]      -> JSON lines, alternating arms, 3 rounds"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from miniengineao_amd import AmbientOcclusion, _lib as L
from bench import WORKLOADS, make_frame

ap = argparse.ArgumentParser()
ap.add_argument("--workloads", default="4k,1080p")
ap.add_argument("--frames", default="1,2,4,8")
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev).cuda_stream
for wl in a.workloads.split(","):
    w, h, kind, cam, intensity, ao_format, _ = WORKLOADS[wl]
    for n in [int(x) for x in a.frames.split(",")]:
        d = [torch.from_numpy(make_frame(kind, w, h, 7 + f)).to(dev) for f in range(n)]
        out = [torch.empty((h, w), dtype=torch.uint8 if ao_format == L.AO_R8 else torch.int16, device=dev) for _ in range(n)]
        dp, op = [t.data_ptr() for t in d], [t.data_ptr() for t in out]
        ctx, ref = {}, None
        for mode in (0, 1, 2):
            c = AmbientOcclusion(w, h, num_levels=4, ao_format=ao_format, max_batch=n, near_clip=cam.near, far_clip=cam.far,
                                 projection00=cam.proj00(w, h), reversed_z=cam.reversed_z)
            c.intensity = intensity
            c.debug_set(L.DEBUG_RENDER_FROM_DEPTH, mode)
            for _ in range(20):
                c.execute_device(dp, op, st)
            torch.cuda.synchronize()
            got = torch.stack(out).clone()
            if ref is None:
                ref = got
            assert torch.equal(got, ref), f"mode {mode} differs from mode 0"
            ctx[mode] = c
        best = {m: 1e9 for m in ctx}
        for _ in range(a.rounds):
            for m, c in ctx.items():
                t0 = time.perf_counter()
                for _ in range(a.iters):
                    c.execute_device(dp, op, st)
                torch.cuda.synchronize()
                best[m] = min(best[m], (time.perf_counter() - t0) / a.iters * 1e6)
        row = {"workload": wl, "frames_per_call": n, "us_per_call": {("stored_mips", "raw_depth_one_launch", "raw_depth_two_streams")[m]: round(v, 2) for m, v in best.items()}}
        for m, c in ctx.items():
            c.set_profiling(True)
            for _ in range(30):
                c.execute_device(dp, op, st)
            ms, _ = c.pass_times_ms()
            row.setdefault("pass_us", {})[("stored_mips", "raw_depth_one_launch", "raw_depth_two_streams")[m]] = \
                {nm[:14]: round(ms[k] * 1e3, 1) for k, nm in enumerate(L.PASS_NAMES) if ms[k] > 0}
            c.close()
        print(json.dumps(row), flush=True)
