"""Generates the committed golden fixtures from the CPU oracle.

The reference (keijiro/MiniEngineAO) ships no golden vectors and cannot be executed in this
environment, so these fixtures are NOT reference outputs: they freeze the oracle's answers
(pinned by tests/test_oracle_kat.py) so that (a) any later change of the oracle is caught and
(b) the GPU tests can compare against data that travels to the GPU box.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz, golden_checksums.json
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from miniengineao_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import helpers as H  # noqa: E402

# name -> (w, h, generator, seed, camera, settings overrides)
SMALL = {
    "s2_67x45_r8": (67, 45, "S2", 11, synth.DEFAULT_CAMERA, {}),
    "s2_130x70_f16_rtne": (130, 70, "S2", 12, synth.DEFAULT_CAMERA, dict(ao_format=1, f16_rounding=1)),
    "s3_96x54_sponza_cam": (96, 54, "S3", 0, synth.SPONZA_CAMERA, dict(intensity=1.1)),
    "s1_64x64_levels2": (64, 64, "S1", 0, synth.DEFAULT_CAMERA, dict(num_levels=2, thickness_modifier=2.0)),
}
# full-size frames: only 64-bit checksums of the result are stored
LARGE = {
    # BASELINE config 1: 1080p radial gradient, single scale (num_levels = 1), CPU plumbing case
    "s1_1080p_single_scale": (1920, 1080, "S1", 0, synth.DEFAULT_CAMERA, dict(num_levels=1)),
    "s3_1080p": (1920, 1080, "S3", 0, synth.SPONZA_CAMERA, dict(intensity=1.1)),
    "s2_1080p": (1920, 1080, "S2", 0x1234ABCD, synth.DEFAULT_CAMERA, {}),
    "s2_4k": (3840, 2160, "S2", 0x1234ABCD, synth.DEFAULT_CAMERA, {}),
    # the SURVEY 8f#4 variants at full size: Render.main on all levels + premin, 68 samples, stereo pair
    "s2_4k_hq4_exhaustive_stereo": (3840, 2160, "S2", 0x1234ABCD, synth.DEFAULT_CAMERA,
                                    dict(hq_levels=4, sample_set=1, single_pass_stereo=True)),
    "s3_1080p_hq2": (1920, 1080, "S3", 0, synth.SPONZA_CAMERA, dict(intensity=1.1, hq_levels=2)),
}


def make_depth(kind, w, h, seed, cam):
    if kind == "S3":
        return synth.atrium(w, h, cam)
    if kind == "S1":
        return synth.radial_gradient(w, h, cam)
    return synth.occluder_field(w, h, seed, cam=cam)


def main():
    O.build()
    sums = {}
    for name, (w, h, kind, seed, cam, over) in SMALL.items():
        depth = make_depth(kind, w, h, seed, cam)
        s = H.settings(O, w, h, cam=cam, **over)
        out = O.run(depth, s)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), depth=depth, **out)
        sums[name] = {k: H.checksum(v) for k, v in out.items()}
        sums[name]["depth"] = H.checksum(depth)
    for name, (w, h, kind, seed, cam, over) in LARGE.items():
        depth = make_depth(kind, w, h, seed, cam)
        s = H.settings(O, w, h, cam=cam, **over)
        out = O.run(depth, s, nthreads=O.host_cores(), result_only=True)
        sums[name] = {"depth": H.checksum(depth), "result": H.checksum(out["result"]),
                      "mean_ao": round(float(out["result"].mean()) / 255.0, 6)}
    with open(os.path.join(HERE, "golden_checksums.json"), "w") as f:
        json.dump(sums, f, indent=1, sort_keys=True)
    print("wrote", len(SMALL), "npz fixtures and golden_checksums.json")


if __name__ == "__main__":
    main()
