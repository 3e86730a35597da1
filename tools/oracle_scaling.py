#!/usr/bin/env python3
"""Thread scaling of the CPU oracle on this host (the `cpu_baseline` of bench.py): Mpixels/s of one 4K frame at several thread
counts, next to what the host says about its CPUs (logical count, affinity mask, cgroup quota)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from miniengineao_amd import synth
from oracle import oracle as O

info = {"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        info[path] = open(path).read().strip()
    except OSError:
        pass
try:
    info["loadavg"] = open("/proc/loadavg").read().strip()
except OSError:
    pass
print(json.dumps(info))
w, h = 3840, 2160
cam = synth.DEFAULT_CAMERA
d = synth.make("S2", w, h, seed=0x1234ABCD)
s = O.Settings(w, h, proj00=cam.proj00(w, h), near_clip=cam.near, far_clip=cam.far, reversed_z=True)
for n in [int(x) for x in (sys.argv[1:] or "1 4 8 16 32 64 128 256".split())]:
    O.run(d, s, nthreads=n, result_only=True)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); O.run(d, s, nthreads=n, result_only=True); ts.append(time.perf_counter() - t)
    print(json.dumps({"threads": n, "Mpix_s": round(w * h / min(ts) / 1e6, 1), "ms": round(min(ts) * 1e3, 1)}), flush=True)
