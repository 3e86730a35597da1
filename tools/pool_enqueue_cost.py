#!/usr/bin/env python3
"""Host cost of feeding G pool members ONE frame each (BASELINE config 4's shape: 8 x 4K frames, one per GPU) from the
pool's single host thread.  All members sit on device 0 here (1-GPU box): the enqueue cost is a host
property, the kernels' durations are not what is measured.

    python tools/pool_enqueue_cost.py [--members 8] [--workload 4k] [--steps 200]

Per launch structure -- direct (4-5 kernel launches per member and step), pipelined (meao_pool_prefetch_batch +
execute: 3-4 launches) -- prints: host microseconds per step spent
inside meao_pool_prefetch_batch + meao_pool_execute_batch WITHOUT synchronising (queues drained first, `steps` steps
enqueued back to back; the HIP queue depth bounds how far the host can run ahead, so the figure is taken over the first
steps only as well), the same per member, and the GPU time of one member's frame for comparison: one host thread can
feed G GPUs at this shape if G x (enqueue cost per member) < the GPU time of one frame."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from miniengineao_amd import AmbientOcclusionPool, _lib
from bench import WORKLOADS, make_frame
from miniengineao_amd.sharding import frame_seed

ap = argparse.ArgumentParser()
ap.add_argument("--members", type=int, default=8)
ap.add_argument("--workload", default="4k")
ap.add_argument("--steps", type=int, default=200)
a = ap.parse_args()
w, h, kind, cam, intensity, ao_format, _ = WORKLOADS[a.workload]
G = a.members
dev = torch.device("cuda", 0)
frames = [torch.from_numpy(make_frame(kind, w, h, frame_seed(0x1234ABCD, g), g)).to(dev) for g in range(G)]
outs = [torch.empty((h, w), dtype=torch.uint8 if ao_format == _lib.AO_R8 else torch.int16, device=dev) for _ in range(G)]
dp, op = [t.data_ptr() for t in frames], [t.data_ptr() for t in outs]
rows = []
for name, kw, prefetch in (("direct", {}, False), ("direct_pipelined", {"pipelined": True}, True)):
    pool = AmbientOcclusionPool(w, h, [0] * G, max_batch=1, ao_format=ao_format, near_clip=cam.near, far_clip=cam.far,
                                projection00=cam.proj00(w, h), reversed_z=cam.reversed_z, intensity=intensity, **kw)

    def step():
        if prefetch:
            pool.prefetch_device(dp)
        pool.execute_device(dp, op)

    for _ in range(20):
        step()
    pool.synchronize()
    # (1) enqueue only: host time inside the pool calls, nothing waited for
    t0 = time.perf_counter()
    marks = []
    for k in range(a.steps):
        step()
        if k in (9, 49):
            marks.append(time.perf_counter() - t0)
    enq = time.perf_counter() - t0
    pool.synchronize()
    total = time.perf_counter() - t0
    # (2) GPU time of one member's frame: one member alone, back to back
    one = AmbientOcclusionPool(w, h, [0], max_batch=1, ao_format=ao_format, near_clip=cam.near, far_clip=cam.far,
                               projection00=cam.proj00(w, h), reversed_z=cam.reversed_z, intensity=intensity, **kw)
    for _ in range(20):
        if prefetch:
            one.prefetch_device(dp[:1])
        one.execute_device(dp[:1], op[:1])
    one.synchronize()
    t1 = time.perf_counter()
    for _ in range(a.steps):
        if prefetch:
            one.prefetch_device(dp[:1])
        one.execute_device(dp[:1], op[:1])
    one.synchronize()
    frame_us = (time.perf_counter() - t1) / a.steps * 1e6
    one.close()
    pool.close()
    row = {"structure": name, "members": G, "workload": a.workload,
           "host_us_per_step_first_10": round(marks[0] / 10 * 1e6, 1), "host_us_per_step_first_50": round(marks[1] / 50 * 1e6, 1),
           "host_us_per_step_all": round(enq / a.steps * 1e6, 1),
           "host_us_per_member_first_10": round(marks[0] / 10 / G * 1e6, 2),
           "wall_us_per_step_incl_gpu": round(total / a.steps * 1e6, 1),
           "gpu_us_per_frame_one_member": round(frame_us, 1),
           "one_thread_feeds_G_gpus": bool(marks[0] / 10 * 1e6 < frame_us)}
    rows.append(row)
    print(json.dumps(row), flush=True)
