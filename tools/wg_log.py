#!/usr/bin/env python3
"""Residency of a persistent upsample launch (diagnostic build -DMEAO_X_PHASE_CLOCKS=1 -DMEAO_X_UPS_PERSISTENT=2|3):
when every workgroup started and ended and on which XCD / SE / CU it ran."""
import ctypes as C, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from miniengineao_amd import AmbientOcclusion, _lib
from bench import WORKLOADS, make_frame, default_batch
w, h, kind, cam, intensity, ao_format, _ = WORKLOADS["4k"]
B = 16
dev = torch.device("cuda", 0)
fr = make_frame(kind, w, h, 1)
dd = [torch.from_numpy(fr).to(dev) for _ in range(B)]
out = [torch.empty((h, w), dtype=torch.uint8, device=dev) for _ in range(B)]
ao = AmbientOcclusion(w, h, num_levels=4, ao_format=ao_format, max_batch=B, near_clip=cam.near, far_clip=cam.far,
                      projection00=cam.proj00(w, h), reversed_z=cam.reversed_z)
lib = _lib.load()
st = torch.cuda.current_stream(dev).cuda_stream
for _ in range(40):
    ao.execute_device([t.data_ptr() for t in dd], [t.data_ptr() for t in out], st)
torch.cuda.synchronize()
n = 1792
buf = (C.c_uint64 * (4 * n))()
lib.meao_x_wg_log.restype = C.c_int
assert lib.meao_x_wg_log(buf, n) == 0
a = np.array(buf[:], dtype=np.uint64).reshape(n, 4)
t0, t1 = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
base = t0.min()
start, end = (t0 - base) / 100.0, (t1 - base) / 100.0
hw, xcc = a[:, 2], a[:, 3] & 0xF
tiles = (a[:, 3] >> 8).astype(np.int64)
print("tiles per workgroup: min %d median %d max %d total %d" % (tiles.min(), np.median(tiles), tiles.max(), tiles.sum()))
cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
print("workgroups", n, "kernel span us", round(float(end.max()), 1))
print("start us: min %.1f  median %.1f  p90 %.1f  max %.1f" % (start.min(), np.median(start), np.percentile(start, 90), start.max()))
print("late starters (> 20 us):", int((start > 20).sum()))
print("duration us: min %.1f median %.1f max %.1f" % ((end - start).min(), np.median(end - start), (end - start).max()))
per_cu = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
print("distinct (xcc, se, sh, cu):", len(per_cu), "workgroups per CU histogram:", sorted(collections.Counter(per_cu.values()).items()))
early = start <= 20
per_cu_early = collections.Counter(zip(xcc[early].tolist(), se[early].tolist(), sh[early].tolist(), cu[early].tolist()))
print("  of the early starters:", sorted(collections.Counter(per_cu_early.values()).items()))
print("xcc of blockIdx 0..15:", xcc[:16].tolist())
dur = end - start
for x in range(8):
    m = xcc == x
    print("xcc %d: n %d  duration min %.0f median %.0f max %.0f  end max %.0f" % (x, int(m.sum()), dur[m].min(), np.median(dur[m]), dur[m].max(), end[m].max()))
# within a CU
keys = list(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
by = collections.defaultdict(list)
for k, d in zip(keys, dur.tolist()):
    by[k].append(d)
spread = sorted(((max(v) - min(v), k, sorted(round(x) for x in v)) for k, v in by.items()), reverse=True)
print("largest within-CU spreads:", spread[:4])
print("smallest within-CU spreads:", spread[-4:])
cu_end = sorted((max(v), k) for k, v in by.items())
print("CU finish times: min %.0f median %.0f max %.0f" % (cu_end[0][0], cu_end[len(cu_end) // 2][0], cu_end[-1][0]))
# tiles per workgroup (static stride) and blockIdx of slowest
order = np.argsort(-dur)
print("slowest blockIdx:", order[:12].tolist(), "fastest:", order[-12:].tolist())
