#!/usr/bin/env python3
"""Two pipelined contexts on one GPU, each on its own stream: both streams unmasked (time sharing, what bench.py --pool 2 does) against
each stream confined to one half of the CUs (hipExtStreamCreateWithCUMask) -- does a spatial split let one member's memory-bound last
kernel run next to the other member's VALU-bound kernels?
    python tools/cu_split_probe.py [--batch 16]"""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from miniengineao_amd import AmbientOcclusion, _lib
from bench import WORKLOADS, make_frame
from miniengineao_amd.sharding import frame_seed

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="4k")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--steps", type=int, default=24)
a = ap.parse_args()
hip = C.CDLL("libamdhip64.so")
w, h, kind, cam, intensity, ao_format, _ = WORKLOADS[a.workload]
B = a.batch
dev = torch.device("cuda", 0)
frames = [make_frame(kind, w, h, frame_seed(0x1234ABCD, f)) for f in range(4)]

def stream(pattern):
    s = C.c_void_p()
    if pattern is None:
        assert hip.hipStreamCreate(C.byref(s)) == 0
    else:
        assert hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, (C.c_uint32 * 8)(*pattern)) == 0
    return s

class Member:
    def __init__(self, st):
        self.st = st
        self.dd = [torch.from_numpy(frames[f % 4]).to(dev) for f in range(B)]
        self.out = [torch.empty((h, w), dtype=torch.uint8, device=dev) for _ in range(B)]
        self.dp, self.op = [t.data_ptr() for t in self.dd], [t.data_ptr() for t in self.out]
        self.ao = AmbientOcclusion(w, h, num_levels=4, ao_format=ao_format, max_batch=B, near_clip=cam.near, far_clip=cam.far,
                                   projection00=cam.proj00(w, h), reversed_z=cam.reversed_z, pipelined=True)
        self.ao.intensity = intensity
    def step(self):
        self.ao.prefetch_device(self.dp)
        self.ao.execute_device(self.dp, self.op, self.st.value)

LOW, HIGH = [0xffffffff] * 4 + [0] * 4, [0] * 4 + [0xffffffff] * 4
CASES = {"one member, default-created stream": [None], "two members, unmasked streams": [None, None],
         "two members, low / high half of the CUs": [LOW, HIGH], "one member, low half": [LOW]}
for name, pats in CASES.items():
    ms = [Member(stream(p)) for p in pats]
    for _ in range(8):
        for m in ms: m.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        for m in ms: m.step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({"case": name, "frames_per_member_step": B, "Gpix_s": round(w * h * B * a.steps * len(ms) / el / 1e9, 1),
                      "us_per_round": round(el / a.steps * 1e6, 1)}), flush=True)
    for m in ms: m.ao.close()
