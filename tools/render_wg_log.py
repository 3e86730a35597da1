#!/usr/bin/env python3
"""Residency of ONE render launch (diagnostic build -DMEAO_X_PHASE_CLOCKS=1): per workgroup the first wave in, the first and
the last wave out and the CU; per CU the time-weighted number of resident workgroups and the hand-over gaps."""
import ctypes as C, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from miniengineao_amd import AmbientOcclusion, _lib
from bench import WORKLOADS, make_frame
w, h, kind, cam, intensity, ao_format, _ = WORKLOADS["4k"]
B = 16
dev = torch.device("cuda", 0)
fr = [make_frame(kind, w, h, 1 + f) for f in range(2)]
dd = [torch.from_numpy(fr[f % 2]).to(dev) for f in range(B)]
out = [torch.empty((h, w), dtype=torch.uint8, device=dev) for _ in range(B)]
ao = AmbientOcclusion(w, h, num_levels=4, ao_format=ao_format, max_batch=B, near_clip=cam.near, far_clip=cam.far,
                      projection00=cam.proj00(w, h), reversed_z=cam.reversed_z)
lib = _lib.load()
st = torch.cuda.current_stream(dev).cuda_stream
dp, op = [t.data_ptr() for t in dd], [t.data_ptr() for t in out]
for _ in range(60):
    ao.execute_device(dp, op, st)
torch.cuda.synchronize()
assert lib.meao_x_wg_log_preset() == 0
ao.execute_device(dp, op, st)
torch.cuda.synchronize()
n = 692 * B
buf = (C.c_uint64 * (4 * n))()
lib.meao_x_wg_log.restype = C.c_int
assert lib.meao_x_wg_log(buf, n) == 0
a = np.array(buf[:], dtype=np.uint64).reshape(n, 4)
ok = a[:, 1] > 0
a = a[ok]
t0, t1, tf = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64), a[:, 2].astype(np.int64)
base = t0.min()
start, end, first_end = (t0 - base) / 100.0, (t1 - base) / 100.0, (tf - base) / 100.0
xcc = (a[:, 3] & 0xF).astype(np.int64)
hw = (a[:, 3] >> 8)
cu, sh, se = ((hw >> 8) & 0xF).astype(np.int64), ((hw >> 12) & 1).astype(np.int64), ((hw >> 13) & 7).astype(np.int64)
span = end.max()
print("workgroups logged", len(a), "launch span us %.1f" % span)
dur = end - start
print("workgroup residency us: median %.2f mean %.2f p10 %.2f p90 %.2f" % (np.median(dur), dur.mean(), np.percentile(dur, 10), np.percentile(dur, 90)))
print("skew inside a workgroup (last wave out - first wave out) us: median %.2f mean %.2f p90 %.2f" % (np.median(end - first_end), (end - first_end).mean(), np.percentile(end - first_end, 90)))
key = xcc * 1000 + se * 100 + sh * 16 + cu
gaps, resident = [], []
for k in np.unique(key):
    m = key == k
    s_, e_ = np.sort(start[m]), np.sort(end[m])
    # time-weighted resident workgroups on this CU over the launch's steady part (10 % .. 90 % of the span)
    lo, hi = 0.1 * span, 0.9 * span
    occ = (np.clip(end[m], lo, hi) - np.clip(start[m], lo, hi)).sum() / (hi - lo)
    resident.append(occ)
    # hand-over: the k-th end is followed by the (k+4)-th start (4 slots per CU)
    for i in range(len(e_) - 4):
        g = s_[i + 4] - e_[i]
        if lo < e_[i] < hi:
            gaps.append(g)
gaps = np.array(gaps)
print("CUs", len(resident), "time-weighted resident workgroups per CU: mean %.2f min %.2f max %.2f (4 slots)" % (np.mean(resident), np.min(resident), np.max(resident)))
print("hand-over (k-th workgroup out -> (k+4)-th workgroup's first wave in) us: median %.2f mean %.2f p10 %.2f p90 %.2f" % (np.median(gaps), gaps.mean(), np.percentile(gaps, 10), np.percentile(gaps, 90)))
