"""What the event records of meao_set_profiling cost a batched step, by MEAO_DEBUG_PROFILE_PASS_MASK: all eight, the dominant
kernel's pair, none, the dominant kernel's and render's (alternating arms; prints one JSON object)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miniengineao_amd import AmbientOcclusion, _lib as L
from bench import WORKLOADS, make_frame
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev).cuda_stream
w, h, kind, cam, intensity, ao_format, _ = WORKLOADS["4k"]
B = 16
d = [torch.from_numpy(make_frame(kind, w, h, 7 + f)).to(dev) for f in range(B)]
out = [torch.empty((h, w), dtype=torch.uint8, device=dev) for _ in range(B)]
dp, op = [t.data_ptr() for t in d], [t.data_ptr() for t in out]
c = AmbientOcclusion(w, h, num_levels=4, ao_format=ao_format, max_batch=B, near_clip=cam.near, far_clip=cam.far,
                     projection00=cam.proj00(w, h), reversed_z=cam.reversed_z, pipelined=True)
c.intensity = intensity
def step():
    c.prefetch_device(dp); c.execute_device(dp, op, st)
for _ in range(600): step()
torch.cuda.synchronize()
U0 = L.PASS_NAMES.index("upsample_L1_to_L0")
arms = {"all_events": (1, 0), "dominant_only": (1, 1 << U0), "none": (0, 0), "dominant_and_render": (1, (1 << U0) | 2)}
res = {k: [] for k in arms}
for rnd in range(4):
    for name, (prof, mask) in arms.items():
        c.debug_set(L.DEBUG_PROFILE_PASS_MASK, mask)
        c.set_profiling(prof)
        for _ in range(20): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300): step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 300 * 1e3
        pm = c.pass_times_ms()[0] if prof else None
        res[name].append((round(ms, 4), None if pm is None else round(pm[U0], 5)))
        c.set_profiling(0)
print(json.dumps(res))
