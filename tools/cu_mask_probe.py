#!/usr/bin/env python3
"""How the pipelined step's kernels scale with the number of CUs they may use (hipExtStreamCreateWithCUMask): the question behind
"two half-chip members, one in its memory-bound kernel while the other is in its VALU-bound ones".
    python tools/cu_mask_probe.py [--workload 4k]       # per-pass HIP-event times for masks of 256 / 192 / 128 / 64 CUs"""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from miniengineao_amd import AmbientOcclusion, _lib
from bench import WORKLOADS, make_frame, default_batch
from miniengineao_amd.sharding import frame_seed

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="4k")
ap.add_argument("--steps", type=int, default=12)
a = ap.parse_args()
hip = C.CDLL("libamdhip64.so")
w, h, kind, cam, intensity, ao_format, _ = WORKLOADS[a.workload]
B = default_batch(w, h)
dev = torch.device("cuda", 0)
frames = [make_frame(kind, w, h, frame_seed(0x1234ABCD, f)) for f in range(min(B, 4))]
dd = [torch.from_numpy(frames[f % len(frames)]).to(dev) for f in range(B)]
out = [torch.empty((h, w), dtype=torch.uint8 if ao_format == _lib.AO_R8 else torch.int16, device=dev) for _ in range(B)]
dp, op = [t.data_ptr() for t in dd], [t.data_ptr() for t in out]

def masked_stream(pattern):
    words = (C.c_uint32 * 8)(*pattern)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return s

# bit i of the mask = CU i in the runtime's enumeration (interleaved over the XCDs / shader engines)
PATTERNS = {
    "256 (all)": [0xffffffff] * 8,
    "192 (3 of 4 bits)": [0x77777777] * 8,
    "128 (every other bit)": [0x55555555] * 8,
    "128 (low half)": [0xffffffff] * 4 + [0] * 4,
    "64 (every fourth bit)": [0x11111111] * 8,
}
for name, pat in PATTERNS.items():
    st = masked_stream(pat)
    for pipe in (False, True):
        ao = AmbientOcclusion(w, h, num_levels=4, ao_format=ao_format, max_batch=B, near_clip=cam.near, far_clip=cam.far,
                              projection00=cam.proj00(w, h), reversed_z=cam.reversed_z, pipelined=pipe)
        ao.intensity = intensity
        def step():
            if pipe:
                ao.prefetch_device(dp)
            ao.execute_device(dp, op, st.value)
        for _ in range(6): step()
        hip.hipStreamSynchronize(st)
        ao.set_profiling(True)
        t0 = time.perf_counter()
        for _ in range(a.steps): step()
        hip.hipStreamSynchronize(st)
        el = time.perf_counter() - t0
        ms, n = ao.pass_times_ms()
        res = {nm: round(ms[k] * 1e3, 1) for k, nm in enumerate(_lib.PASS_NAMES) if ms[k] > 0}
        res.update(cus=name, pipelined=pipe, step_us=round(el / a.steps * 1e6, 1))
        print(json.dumps(res), flush=True)
        ao.close()
