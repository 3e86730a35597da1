#!/usr/bin/env python3
"""bench.py -- throughput of the multi-scale SSAO hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 4k|1080p|8k] [--batch B]

A step = one pass of the full pipeline (downsample -> render x4 levels -> upsample x4) over
one batch of B independent synthetic frames, depth already resident in HBM.  Prints ONE JSON
line: Mpixels/s (whole job, all ranks), the roofline of the dominant kernel measured live
with HIP events on the launch stream, and the CPU oracle timed on the host cores (rank 0,
N=1 only) as a reported baseline.

N>1: launched by torch.distributed.run, one rank per GPU; frames are sharded across ranks
(weak scaling: B frames per rank), there is no data-path collective -- RCCL is used only for
the barrier and the max-over-ranks of the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch  # before libmeao_hip.so: both then share torch's libamdhip64 (see _lib.py)

from miniengineao_amd import AmbientOcclusion, _lib, synth
from miniengineao_amd import distributed as mdist
from miniengineao_amd.sharding import frame_checksum, frame_seed, frames_for_rank

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md chip table)
HBM_COPY_CEILING_GBPS = 6290.0

WORKLOADS = {
    # name: (W, H, generator, camera, intensity, ao_format, description)
    "4k": (3840, 2160, "S2", synth.DEFAULT_CAMERA, 1.0, _lib.AO_R8,
           "4K (3840x2160) S2 synthetic depth, full 4-level pipeline, R8 AO"),
    "1080p": (1920, 1080, "S3", synth.SPONZA_CAMERA, 1.1, _lib.AO_R8,
              "1080p S3 Sponza-like atrium (substitute for captured Sponza depth), full 4-level pipeline, R8 AO"),
    "8k": (7680, 4320, "S2", synth.DEFAULT_CAMERA, 1.0, _lib.AO_F16,
           "8K (7680x4320) S2 synthetic depth, full 4-level pipeline, fp16 AO storage"),
}


_ATRIUM = {}


def make_frame(kind: str, w: int, h: int, seed: int) -> np.ndarray:
    if kind == "S3":                      # analytic scene, no seed: every frame of the batch is a separate copy
        if (w, h) not in _ATRIUM:
            _ATRIUM[(w, h)] = synth.atrium(w, h)
        return _ATRIUM[(w, h)]
    return synth.make(kind, w, h, seed=seed)


def cpu_baseline(w, h, cam, intensity, ao_format, depth, budget_s=20.0):
    """The oracle (a port of the reference passes), row-parallel on all host cores, timed on a
    bounded sample of the same workload: whole frames of this workload, median of <=8 after
    one warm-up, stopping early once ~budget_s of CPU time is spent."""
    from oracle import oracle as O   # test infrastructure: used here only as the reported baseline
    cores = os.cpu_count() or 1
    s = O.Settings(w, h, proj00=cam.proj00(w, h), near_clip=cam.near, far_clip=cam.far,
                   reversed_z=cam.reversed_z, intensity=intensity, ao_format=ao_format)
    times = []
    t_begin = time.perf_counter()
    O.run(depth, s, nthreads=cores, result_only=True)       # warm-up
    for _ in range(8):
        t0 = time.perf_counter()
        O.run(depth, s, nthreads=cores, result_only=True)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_begin > budget_s:
            break
    med = float(np.median(times))
    t0 = time.perf_counter()
    O.run(depth, s, nthreads=1, result_only=True)           # the same frame on one core (scalar C)
    single = time.perf_counter() - t0
    return {"value": round(w * h / med / 1e6, 3), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "single_core_value": round(w * h / single / 1e6, 3),
            "sample": f"{len(times)} timed full {w}x{h} frame(s) of the bench workload after 1 warm-up, "
                      f"median; C oracle (oracle/meao_oracle.c) row-parallel on {cores} threads",
            "seconds_per_frame": round(med, 4)}


def pmc_traffic(workload: str, kernel: str, frames_per_launch: int):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 --pmc passes
    (profiles/pmc_traffic.json: FETCH_SIZE / WRITE_SIZE collected in separate runs by
    tests/run_pmc.sh, corrected as MI355X_MICROARCH.md prescribes; per frame, scaled here by the
    frames per launch).  None when no measurement exists for this workload/kernel."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        table = json.load(open(path))
        row = table[workload][kernel]
        out = {"bytes": int(row["bytes_per_frame"] * frames_per_launch), "source": "profiles/pmc_traffic.json",
               "note": row.get("note", "")}
        if "valu_wave_insts_per_frame" in row:      # SQ_INSTS_VALU of the same PMC passes
            out["valu_wave_insts"] = int(row["valu_wave_insts_per_frame"] * frames_per_launch)
        return out
    except (OSError, KeyError, ValueError):
        return None


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="4k")
    ap.add_argument("--batch", type=int, default=None,
                    help="independent frames per step per GPU (1..64); default: 133 Mpixels per step "
                         "(64 frames at 1080p, 16 at 4K, 4 at 8K)")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="batches in flight: N contexts on N HIP streams, step k on stream k mod N "
                         "(lets the HBM-bound downsample of one batch overlap the VALU-bound passes of another)")
    ap.add_argument("--composite", action="store_true",
                    help="also time the next-tier composite kernel (AO x RGBA16F frame, Blit.shader pass 2); "
                         "reported separately, never part of `value`")
    ap.add_argument("--shaded", action="store_true",
                    help="next tier, reported separately: 'depth in -> shaded frame out' with the composite of each "
                         "step riding inside the next step's render kernel (meao_composite_enqueue), next to the "
                         "same work as separate composite launches")
    ap.add_argument("--ao-format", choices=["r8", "f16"], default=None,
                    help="override the AO storage of the workload (R8 = reference, F16 = fp16 AO)")
    ap.add_argument("--fast-numerics", action="store_true",
                    help="MEAO_NUMERICS_FAST (raw v_rcp_f32 divides; NOT bit-exact, reported as such)")
    ap.add_argument("--hq-levels", type=int, default=0,
                    help="variant (not the reference's wiring): the coarsest N levels also run Render.main "
                         "(wide) and the upsamples become main_premin*; changes config.workload")
    ap.add_argument("--exhaustive", action="store_true",
                    help="variant: SAMPLE_EXHAUSTIVELY (68 samples instead of 36); changes config.workload")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="do not use meao_prefetch_batch: every step launches its own downsample pass instead of "
                         "carrying the next step's inside its last upsample kernel")
    ap.add_argument("--roctx", action="store_true",
                    help="meao_set_tracing: roctx ranges around every pass (for rocprofv3 --marker-trace)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--min-time-ms", type=float, default=0.0,
                    help="raise --steps so that the timed region lasts at least this long (rank skew matters "
                         "less in a >= 100 ms region); the JSON reports the steps actually timed")
    ap.add_argument("--validate-frames", type=int, default=2,
                    help="frames per rank checked bit-for-bit against the CPU oracle after the timed region "
                         "(0 = checksums only)")
    ap.add_argument("--skip-latency", action="store_true",
                    help="skip the single-frame latency loop (keeps profiler traces to the batched launches)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N>1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    mdist.init("nccl", dev)                                      # nccl == RCCL on ROCm

    w, h, kind, cam, intensity, ao_format, desc = WORKLOADS[args.workload]
    if args.ao_format is not None:
        ao_format = _lib.AO_R8 if args.ao_format == "r8" else _lib.AO_F16
        desc = desc.replace("fp16 AO storage", "AO").replace("R8 AO", "AO") + \
            (", R8 storage" if ao_format == _lib.AO_R8 else ", fp16 storage")
    if args.hq_levels:
        desc += f", VARIANT hq_levels={args.hq_levels}"
    if args.exhaustive:
        desc += ", VARIANT 68-sample set"
    batch = args.batch if args.batch is not None else max(1, (3840 * 2160 * 16) // (w * h))
    B = max(1, min(batch, _lib.MAX_BATCH))
    ao_dtype = torch.uint8 if ao_format == _lib.AO_R8 else torch.int16

    # synthetic frames of this rank, resident in HBM: global frame g goes to rank g mod world
    # (miniengineao_amd.sharding.frames_for_rank), weak scaling: B frames per rank
    my_frames = frames_for_rank(world * B, rank, world)
    frames = [make_frame(kind, w, h, frame_seed(0x1234ABCD, g)) for g in my_frames]
    depth_dev = [torch.from_numpy(f).to(dev) for f in frames]
    nfl = max(1, args.in_flight)
    out_dev = [[torch.empty((h, w), dtype=ao_dtype, device=dev) for _ in range(B)] for _ in range(nfl)]

    ctxs = []
    for _ in range(nfl):
        c = AmbientOcclusion(w, h, device=local_rank, num_levels=4, ao_format=ao_format, max_batch=B,
                             near_clip=cam.near, far_clip=cam.far, projection00=cam.proj00(w, h),
                             reversed_z=cam.reversed_z, hq_levels=args.hq_levels,
                             sample_set=_lib.SAMPLES_EXHAUSTIVE if args.exhaustive else _lib.SAMPLES_CHECKER,
                             numerics=_lib.NUMERICS_FAST if args.fast_numerics else _lib.NUMERICS_STRICT,
                             pipelined=not args.no_pipeline)
        c.intensity = intensity
        if args.roctx:
            c.set_tracing(True)
        ctxs.append(c)
    ao = ctxs[0]
    tstreams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(nfl - 1)]
    streams = [t.cuda_stream for t in tstreams]
    stream = streams[0]
    dptr = [t.data_ptr() for t in depth_dev]
    optrs = [[t.data_ptr() for t in outs] for outs in out_dev]
    optr = optrs[0]
    counter = [0]

    pipelined = not args.no_pipeline
    use_prefetch = [pipelined]

    def step():
        k = counter[0] % nfl
        counter[0] += 1
        if use_prefetch[0]:
            # streaming use: the frames of this context's NEXT step are announced, so this step's last
            # kernel also runs their downsample pass (every step still does one downsample pass of work)
            ctxs[k].prefetch_device(dptr)
        ctxs[k].execute_device(dptr, optrs[k], streams[k])

    def fence():
        mdist.fence(dev)     # synchronize + barrier + synchronize

    # clock ramp: ~80 ms of untimed back-to-back work before the W warm-up steps.  The chip's clocks and
    # power state settle slowly: in the rocprofv3 trace of this command the render kernel goes 260 ->
    # 200 -> 185 -> 178 us over the first 40 ms of continuous load (profiles/README.md), so a run with a
    # small --steps / --warmup would otherwise be timed on the slope.
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.080:
        for _ in range(8):
            step()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    if args.min_time_ms > 0:
        # size the timed region: all ranks agree on the step count (max over ranks of the estimate)
        fence()
        t_est = time.perf_counter()
        for _ in range(3):
            step()
        fence()
        est = mdist.max_over_ranks((time.perf_counter() - t_est) / 3.0, dev)
        args.steps = max(args.steps, int(np.ceil(args.min_time_ms * 1e-3 / max(est, 1e-6))))
    for c in ctxs:
        c.set_profiling(True)       # HIP events around every pass, on the launch stream
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    per_ctx = [c.pass_times_ms() for c in ctxs]
    samples = sum(n for _, n in per_ctx)
    pass_ms = [sum(ms[k] * n for ms, n in per_ctx) / max(samples, 1) for k in range(_lib.NUM_PASSES)]
    for c in ctxs:
        c.set_profiling(False)

    my_elapsed = elapsed
    elapsed = mdist.max_over_ranks(elapsed, dev)
    per_rank_ms = mdist.gather_floats(my_elapsed / args.steps * 1e3, dev)

    # for reference, the same K steps as the plain launch sequence (every step runs its own downsample
    # pass), with per-pass events: this is where the north-star sub-path (render + upsample passes,
    # nothing else inside those kernels) is timed
    plain, plain_pass_ms = None, None
    if pipelined:
        use_prefetch[0] = False
        for _ in range(3):
            step()
        for c in ctxs:
            c.set_profiling(True)
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        plain_elapsed = mdist.max_over_ranks(time.perf_counter() - t1, dev)
        per_ctx_p = [c.pass_times_ms() for c in ctxs]
        samples_p = sum(n for _, n in per_ctx_p)
        plain_pass_ms = [sum(ms[k] * n for ms, n in per_ctx_p) / max(samples_p, 1) for k in range(_lib.NUM_PASSES)]
        for c in ctxs:
            c.set_profiling(False)
        plain = {"value": round(float(w) * h * B * args.steps * world / plain_elapsed / 1e6, 1),
                 "ms_per_step": round(plain_elapsed / args.steps * 1e3, 4),
                 "pass_ms": {n: round(plain_pass_ms[k], 5) for k, n in enumerate(_lib.PASS_NAMES) if plain_pass_ms[k] > 0}}
        use_prefetch[0] = True
    else:
        plain_pass_ms = pass_ms

    # ---- validation, outside every timed region: one checksum per frame of this rank, gathered over
    # RCCL; a few frames per rank compared bit-for-bit with the CPU oracle
    torch.cuda.synchronize(dev)
    my_sums = [frame_checksum(t.cpu().numpy()) for t in out_dev[(counter[0] - 1) % nfl]]
    all_sums = mdist.gather_checksums(my_sums, dev)
    validated, mismatched = 0, 0
    if args.validate_frames > 0 and not args.fast_numerics:
        from oracle import oracle as O   # test infrastructure: the checker, never the thing measured
        s = O.Settings(w, h, proj00=cam.proj00(w, h), near_clip=cam.near, far_clip=cam.far,
                       reversed_z=cam.reversed_z, intensity=intensity, ao_format=ao_format,
                       hq_levels=args.hq_levels,
                       sample_set=O.SAMPLES_EXHAUSTIVE if args.exhaustive else O.SAMPLES_CHECKER)
        pick = sorted(set([0, B - 1][: args.validate_frames] + list(range(min(B, args.validate_frames)))))[: args.validate_frames]
        for f in pick:
            want = O.run(frames[f], s, nthreads=max(1, (os.cpu_count() or 1) // max(world, 1)), result_only=True)["result"]
            got = out_dev[(counter[0] - 1) % nfl][f].cpu().numpy().view(want.dtype)
            validated += 1
            mismatched += int(not np.array_equal(got, want))
    v = mdist.gather_floats(float(validated), dev), mdist.gather_floats(float(mismatched), dev)
    validation = {"frames_checksummed": sum(len(r) for r in all_sums),
                  "distinct_checksums": len({c for r in all_sums for c in r}),
                  "frames_vs_oracle": int(sum(v[0])), "mismatching_frames": int(sum(v[1])),
                  "sharding": "frame g -> rank g mod world (frames_for_rank)",
                  "rank0_first_checksum": all_sums[0][0] if all_sums and all_sums[0] else None}

    total_pixels = float(w) * h * B * args.steps * world
    value = total_pixels / elapsed / 1e6

    # roofline of the dominant kernel: algorithmic bytes per launch / measured launch duration
    alg = ao.algorithmic_bytes()                 # per frame, reference storage formats
    names = list(_lib.PASS_NAMES)
    ren_ups_bytes = sum(alg[1:])                 # render + upsample passes only (north_star's sub-path)
    if pipelined:
        # the downsample pass of the next step runs inside the last upsample kernel: its bytes move there
        u0, d0 = names.index("upsample_L1_to_L0"), names.index("downsample")
        alg[u0] += alg[d0]
        alg[d0] = 0
        names[u0] = "upsample_L1_to_L0+downsample_next"
    u3, u2 = names.index("upsample_L4_to_L3"), names.index("upsample_L3_to_L2")
    if pass_ms[u3] <= 0 < pass_ms[u2]:
        # the library evaluates L4 -> L3 inside the L3 -> L2 launch (upsample_two_level_kernel): its bytes move there
        alg[u2] += alg[u3]
        alg[u3] = 0
        names[u2] = "upsample_L4_to_L3+L3_to_L2"
    u1 = names.index("upsample_L2_to_L1")
    if pass_ms[u3] <= 0 and pass_ms[u2] <= 0 < pass_ms[u1] and alg[u2] > 0:
        # small calls: all three blend passes run inside the L2 -> L1 launch (upsample_three_level_kernel)
        alg[u1] += alg[u2] + alg[u3]
        alg[u2] = alg[u3] = 0
        names[u1] = "upsample_L4_to_L3+L3_to_L2+L2_to_L1"
    dominant = int(np.argmax(pass_ms))
    passes = []
    for k in range(_lib.NUM_PASSES):
        if pass_ms[k] <= 0:
            continue
        gbps = alg[k] * B / (pass_ms[k] * 1e-3) / 1e9
        passes.append({"kernel": names[k], "ms": round(pass_ms[k], 5),
                       "algorithmic_MB": round(alg[k] * B / 1e6, 3), "GBps": round(gbps, 1),
                       "frac": round(gbps / HBM_PEAK_GBPS, 4)})
    dom_gbps = alg[dominant] * B / (pass_ms[dominant] * 1e-3) / 1e9
    traffic = pmc_traffic(args.workload, names[dominant], B)
    kernel_ms = float(sum(pass_ms))
    whole_gbps = sum(alg) * B / (kernel_ms * 1e-3) / 1e9
    # north_star's sub-path: the render + upsample passes alone (plain launch sequence: nothing else rides in those kernels)
    ren_ups_ms = sum(plain_pass_ms[1:])
    ren_ups_gbps = ren_ups_bytes * B / (ren_ups_ms * 1e-3) / 1e9
    if traffic and "valu_wave_insts" in traffic:
        # what actually limits the kernel: VALU wave-instructions issued per SIMD (1024 SIMDs) over the
        # measured launch time; tools/ubench_valu.hip: 3.0 (fma/mul/add) .. 4.4 (med3/cmp/cvt) .. 8.4 (rcp)
        # cycles per instruction at the 2.4 GHz peak clock
        traffic["valu_cycles_per_wave_inst_per_simd"] = round(
            pass_ms[dominant] * 1e-3 * 2.4e9 / (traffic["valu_wave_insts"] / 1024.0), 2)
    roofline = {"bound": "hbm", "limiter": "valu" if dominant != 0 else "hbm",
                "kernel": names[dominant], "achieved": round(dom_gbps, 1),
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(dom_gbps / HBM_PEAK_GBPS, 4),
                "traffic": traffic, "launch_ms": round(pass_ms[dominant], 5), "event_samples": samples,
                "whole_frame": {"GBps": round(whole_gbps, 1), "frac": round(whole_gbps / HBM_PEAK_GBPS, 4)},
                # north_star's sub-path (target: frac >= 0.60), timed per pass in the plain launch sequence
                "render_plus_upsample": {"GBps": round(ren_ups_gbps, 1), "frac": round(ren_ups_gbps / HBM_PEAK_GBPS, 4),
                                         "ms_per_launch": round(ren_ups_ms, 5),
                                         "algorithmic_MB": round(ren_ups_bytes * B / 1e6, 2),
                                         "timed_in": "plain_launch_sequence (same process, per-pass HIP events)"},
                # the same fraction in bytes that actually moved (PMC traffic of the dominant kernel / its time)
                "real_traffic_frac": None if not traffic else round(
                    traffic["bytes"] / (pass_ms[dominant] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "vs_copy_ceiling_frac": round(dom_gbps / HBM_COPY_CEILING_GBPS, 4), "passes": passes}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not (args.hq_levels or args.exhaustive):
        cpu = cpu_baseline(w, h, cam, intensity, ao_format, frames[0])

    # single-frame latency (one frame per launch sequence), for context
    composite = None
    if args.composite:
        # next-tier consumer of the AO texture: color.rgba *= ao, 17 bytes per texel, HBM-bound
        colors = [torch.ones((h, w, 4), dtype=torch.float16, device=dev) for _ in range(B)]
        reps = 10
        for f in range(B):
            ao.composite_device(_lib.COMPOSITE_MULTIPLY, optr[f], colors[f].data_ptr(), 0, stream)
        fence()
        tc = time.perf_counter()
        for _ in range(reps):
            for f in range(B):
                ao.composite_device(_lib.COMPOSITE_MULTIPLY, optr[f], colors[f].data_ptr(), 0, stream)
        torch.cuda.synchronize(dev)
        per_frame_ms = (time.perf_counter() - tc) * 1e3 / (reps * B)    # back-to-back launches, device-synchronised wall time
        cbytes = w * h * (16 + (1 if ao_format == _lib.AO_R8 else 2))
        cg = cbytes / (per_frame_ms * 1e-3) / 1e9
        composite = {"kernel": "composite_multiply", "ms_per_frame": round(per_frame_ms, 5),
                     "algorithmic_MB_per_frame": round(cbytes / 1e6, 2), "GBps": round(cg, 1),
                     "frac": round(cg / HBM_PEAK_GBPS, 4), "bound": "hbm"}

    shaded = None
    if args.shaded:
        # depth in -> shaded frame out: every step = AO of B frames + composite (color.rgba *= ao) of those frames
        colors = [torch.ones((h, w, 4), dtype=torch.float16, device=dev) for _ in range(B)]
        cptr = [t.data_ptr() for t in colors]

        def shaded_steps(pipelined_composite):
            for _ in range(args.steps):
                if pipelined:
                    ao.prefetch_device(dptr)
                ao.execute_device(dptr, optr, stream)          # carries the composite enqueued by the previous step
                if pipelined_composite:
                    ao.composite_enqueue_device(_lib.COMPOSITE_MULTIPLY, optr, cptr)
                else:
                    for f in range(B):
                        ao.composite_device(_lib.COMPOSITE_MULTIPLY, optr[f], cptr[f], 0, stream)
            ao.composite_flush(stream)
        shaded = {}
        for name, flag in (("separate_composite_launches", False), ("composite_inside_next_render", True)):
            shaded_steps(flag)                                  # warm-up
            fence()
            ts = time.perf_counter()
            shaded_steps(flag)
            fence()
            dt = mdist.max_over_ranks(time.perf_counter() - ts, dev)
            shaded[name] = {"Mpixels_per_s": round(float(w) * h * B * args.steps * world / dt / 1e6, 1),
                            "ms_per_step": round(dt / args.steps * 1e3, 4)}
        shaded["note"] = "next tier, not part of `value`: AO + Blit.shader pass 2 on an RGBA16F frame per step"

    # single-frame use (one frame per call, the real-time case): one launch per pass (DIRECT) vs one
    # hipGraphLaunch per call (MEAO_LAUNCH_GRAPH); back-to-back calls, and call + wait per frame
    latency_ms, single = None, None
    if not args.skip_latency:
        def one_frame_ctx(**kw):
            c = AmbientOcclusion(w, h, device=local_rank, num_levels=4, ao_format=ao_format, max_batch=1,
                                 near_clip=cam.near, far_clip=cam.far, projection00=cam.proj00(w, h),
                                 reversed_z=cam.reversed_z, hq_levels=args.hq_levels,
                                 sample_set=_lib.SAMPLES_EXHAUSTIVE if args.exhaustive else _lib.SAMPLES_CHECKER,
                                 numerics=_lib.NUMERICS_FAST if args.fast_numerics else _lib.NUMERICS_STRICT, **kw)
            c.intensity = intensity
            return c
        lat_iters = 50
        single = {}
        # "pipelined": a stream of single frames whose next depth buffer is known one call ahead
        # (meao_prefetch_batch with n = 1: each call's last kernel carries the next frame's downsample pass)
        variants = (("direct", {}), ("graph", {"launch_mode": _lib.LAUNCH_GRAPH}), ("direct_pipelined", {"pipelined": True}))
        for name, kw in variants:
            c = one_frame_ctx(**kw)
            for sync_each in (False, True):
                for _ in range(3):
                    if name == "direct_pipelined":
                        c.prefetch_device(dptr[:1])
                    c.execute_device(dptr[:1], optr[:1], stream)
                fence()
                t0 = time.perf_counter()
                for _ in range(lat_iters):
                    if name == "direct_pipelined":
                        c.prefetch_device(dptr[:1])
                    c.execute_device(dptr[:1], optr[:1], stream)
                    if sync_each:
                        torch.cuda.synchronize(dev)
                torch.cuda.synchronize(dev)
                single[name + ("_call_and_wait_ms" if sync_each else "_back_to_back_ms")] = \
                    round((time.perf_counter() - t0) / lat_iters * 1e3, 4)
            c.close()
        latency_ms = single["direct_back_to_back_ms"]

    if rank == 0:
        line = {
            "metric": "AO Mpixels/s (full multi-scale SSAO pipeline, depth resident in HBM)",
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": desc, "width": w, "height": h, "frames_per_step_per_gpu": B,
                       "num_levels": 4, "ao_storage": "R8" if ao_format == _lib.AO_R8 else "F16",
                       "numerics": "FAST (raw rcp, not bit-exact)" if args.fast_numerics else "strict (bit-exact vs CPU oracle)", "sharding": f"frames x{world}",
                       "batches_in_flight": nfl,
                       "downsample": "pipelined: each step's last kernel carries the next step's downsample pass "
                                     "(meao_prefetch_batch)" if pipelined else "own pass per step"},
            "roofline": roofline, "cpu_baseline": cpu, "composite_next_tier": composite,
            "depth_in_to_shaded_frame_out": shaded,
            "per_rank_ms_per_step": [round(x, 4) for x in per_rank_ms],
            "world_seen_by_rccl": mdist.world_size(), "validation": validation,
            "single_frame_latency_ms": None if latency_ms is None else round(latency_ms, 4),
            "single_frame": single,
            "plain_launch_sequence": plain,
            "sum_kernel_ms_per_step": round(kernel_ms, 4),
        }
        print(json.dumps(line), flush=True)
    for c in ctxs:
        c.close()
    mdist.shutdown()
    return 0


if __name__ == "__main__":
    sys.exit(main())
