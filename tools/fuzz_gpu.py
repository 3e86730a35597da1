"""Long randomized GPU-vs-oracle sweep (not part of the default suite; run on demand):
    python tools/fuzz_gpu.py [count] [seed0]
Emphasises the vector fast paths (widths that are multiples of 4 / 8 / 64) and tile-edge cases."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from miniengineao_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import helpers as H  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
bad = 0
for k in range(count):
    rng = np.random.default_rng(seed0 + k)
    mode = k % 4
    if mode == 0:
        w, h = int(rng.integers(1, 200)) * 4, int(rng.integers(1, 400))
    elif mode == 1:
        w, h = int(rng.integers(1, 14)) * 64 + int(rng.choice([0, 4, 8, 60])), int(rng.integers(1, 10)) * 32 + int(rng.integers(0, 3))
    elif mode == 2:
        w, h = int(rng.integers(1, 900)), int(rng.integers(1, 500))
    else:
        w, h = int(rng.integers(30, 90)) * 8, int(rng.integers(20, 60)) * 8
    reversed_z = bool(rng.integers(0, 2))
    cam = synth.Camera(near=float(rng.uniform(0.01, 0.6)), far=float(rng.uniform(10, 2000)),
                       fov_y_deg=float(rng.uniform(12, 90)), reversed_z=reversed_z)
    depth_format = int(rng.integers(0, 4))
    s = H.settings(O, w, h, cam=cam, ao_format=int(rng.integers(0, 2)), f16_rounding=int(rng.integers(0, 2)),
                   num_levels=int(rng.integers(1, 5)), noise_filter_tolerance=float(rng.uniform(-8, 0)),
                   blur_tolerance=float(rng.uniform(-8, -1)), upsample_tolerance=float(rng.uniform(-12, -1)),
                   thickness_modifier=float(rng.uniform(1, 10)), intensity=float(rng.uniform(0, 2)),
                   depth_format=depth_format)
    if k % 2:
        s.hq_levels, s.sample_set = int(rng.integers(0, s.num_levels + 1)), int(rng.integers(0, 2))
        s.single_pass_stereo = bool(rng.integers(0, 2))
    raw = synth.occluder_field(w, h, seed=k, n_rects=20, n_discs=20, cam=cam)
    if k % 3 == 0:
        x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
        raw[y0:y0 + 50, x0:x0 + 70] = 0.0 if reversed_z else 1.0
    hostile = depth_format == 0 and k % 5 == 0
    if hostile:      # NaN / inf / negative / > 1 / denormal raw depths sprinkled in (IEEE-division bodies)
        bad_px = H.hostile_frame(w, h, seed0 + k, cam=cam, density=0.003)
        mask = rng.random((h, w)) < 0.004
        raw = np.where(mask, bad_px, raw).astype(np.float32)
    depth = O.encode_depth(raw, depth_format) if not hostile else raw
    want = O.run(depth, s, nthreads=8)
    same = (lambda a, b: H.nan_aware_equal(a, b)[0]) if hostile else np.array_equal
    from miniengineao_amd import AmbientOcclusion
    ao = AmbientOcclusion(w, h, num_levels=s.num_levels, ao_format=s.ao_format, f16_rounding=s.f16_rounding,
                          depth_format=depth_format, near_clip=s.near_clip, far_clip=s.far_clip,
                          projection00=s.proj00, reversed_z=reversed_z, max_batch=2, hq_levels=s.hq_levels,
                          sample_set=s.sample_set, single_pass_stereo=s.single_pass_stereo,
                          pipelined=bool(k % 2))
    ao.noiseFilterTolerance, ao.blurTolerance, ao.upsampleTolerance = s.noise_filter_tolerance, s.blur_tolerance, s.upsample_tolerance
    ao.thicknessModifier, ao.intensity = s.thickness_modifier, s.intensity
    if k % 7 == 3:          # a seventh of the cases: the L2 -> L1 pass on 64 x 64 tiles whatever the size (large R8 batches take them by default)
        from miniengineao_amd import _lib
        ao.debug_set(_lib.DEBUG_BLEND_TALL_MIN_TILES, 1)
        ao.debug_set(_lib.DEBUG_NESTED_MAX_TILES, 0)
    if k % 5 == 2:          # a fifth: the full-resolution pass on 64 x 64 tiles whatever the size (calls with many tiles take them by default)
        from miniengineao_amd import _lib
        ao.debug_set(_lib.DEBUG_FINAL_SMALL_MAX_TILES, 0)
    outs = ao.render_batch([depth, depth])
    ok = same(outs[0], want["result"]) and same(outs[1], want["result"])
    for i in H.valid_debug_ids(s.num_levels, s.hq_levels):
        ok = ok and same(ao.debug_buffer(i, frame=1), want[H.NAMES[i]])
    if k % 3 == 1:
        # the pipelined path: announce the same frames, run twice; the second call consumes the downsample
        # that rode inside the first call's last kernel
        import torch
        raw_bytes = np.ascontiguousarray(depth).view(np.uint8)
        d_in = [torch.from_numpy(raw_bytes.copy()).cuda() for _ in range(2)]
        d_out = [torch.zeros((h, w), dtype=torch.uint8 if s.ao_format == 0 else torch.int16, device="cuda") for _ in range(2)]
        pin, pout = [t.data_ptr() for t in d_in], [t.data_ptr() for t in d_out]
        ao.prefetch_device(pin)
        ao.execute_device(pin, pout)
        ao.execute_device(pin, pout)
        ao.synchronize()
        for t in d_out:
            ok = ok and same(t.cpu().numpy().view(want["result"].dtype), want["result"])
        for i in H.valid_debug_ids(s.num_levels, s.hq_levels):
            ok = ok and same(ao.debug_buffer(i, frame=1), want[H.NAMES[i]])
    ao.close()
    if not ok:
        bad += 1
        print("MISMATCH case", k, "seed", seed0 + k, (w, h), s)
print(f"fuzz: {count} cases, {bad} mismatching")
sys.exit(1 if bad else 0)
