"""Per-pass kernel times of the plain launch sequence (HIP events inside the library), for A/B runs of
kernel variants:   MEAO_LIB_PATH=<variant .so> python tools/bench_passes.py [--workload 4k] [--steps 20]
Prints one JSON line {pass: us per launch, ..., "step_us": wall per step, "ok": results match the oracle}."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from miniengineao_amd import AmbientOcclusion, _lib, synth
from bench import WORKLOADS, make_frame
from miniengineao_amd.sharding import frame_seed

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="4k")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--batch", type=int, default=None)
ap.add_argument("--pipeline", action="store_true")
ap.add_argument("--check", action="store_true")
ap.add_argument("--tag", default="")
ap.add_argument("--debug-set", action="append", default=[], metavar="KEY=VALUE",
                help="meao_debug_set before the run, e.g. BLEND_TALL_MIN_TILES=0 (repeatable)")
a = ap.parse_args()
w, h, kind, cam, intensity, ao_format, _ = WORKLOADS[a.workload]
B = a.batch or max(1, (3840 * 2160 * 16) // (w * h))
dev = torch.device("cuda", 0)
frames = [make_frame(kind, w, h, frame_seed(0x1234ABCD, f)) for f in range(B)]
dd = [torch.from_numpy(f).to(dev) for f in frames]
out = [torch.empty((h, w), dtype=torch.uint8 if ao_format == _lib.AO_R8 else torch.int16, device=dev) for _ in range(B)]
ao = AmbientOcclusion(w, h, num_levels=4, ao_format=ao_format, max_batch=B, near_clip=cam.near, far_clip=cam.far,
                      projection00=cam.proj00(w, h), reversed_z=cam.reversed_z, pipelined=a.pipeline)
ao.intensity = intensity
for kv in a.debug_set:
    key, value = kv.split("=")
    ao.debug_set(getattr(_lib, "DEBUG_" + key), int(value))
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
for _ in range(3): step()
ao.set_profiling(True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): step()
torch.cuda.synchronize(); el = time.perf_counter() - t0
ms, n = ao.pass_times_ms()
res = {nm: round(ms[k] * 1e3, 1) for k, nm in enumerate(_lib.PASS_NAMES) if ms[k] > 0}
res["step_us"] = round(el / a.steps * 1e6, 1)
res["Gpix_s"] = round(w * h * B * a.steps / el / 1e9, 1)
if a.check:
    from oracle import oracle as O
    s = O.Settings(w, h, proj00=cam.proj00(w, h), near_clip=cam.near, far_clip=cam.far, reversed_z=cam.reversed_z,
                   intensity=intensity, ao_format=ao_format)
    ok = True
    for f in sorted({0, B - 1}):
        want = O.run(frames[f], s, nthreads=O.host_cores(), result_only=True)["result"]
        ok = ok and bool(np.array_equal(out[f].cpu().numpy().view(want.dtype), want))
    res["ok"] = ok
res["tag"] = (a.tag or os.path.basename(os.environ.get("MEAO_LIB_PATH", "product"))) + \
    "".join("+" + kv for kv in a.debug_set)
print(json.dumps(res), flush=True)
