"""Turn the rocprofv3 --pmc passes of tests/run_pmc.sh into profiles/pmc_traffic.json.

    python tests/make_pmc_traffic.py gpurun_out/pmc_<tag> <workload> <frames_per_launch>

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  Correction per MI355X_MICROARCH.md
(HBM section): on gfx950 FETCH_SIZE tallies a 128-byte request of a wide (16 B/lane) coalesced
stream as 64 bytes, so kernels whose reads are 16 B/lane streams are doubled; this is calibrated
on downsample_kernel, whose input bytes are known exactly (4 B x W x H per frame, read once).
Kernels with narrower lanes are left raw and say so.  WRITE_SIZE is used as reported (it matches
the known output bytes of every kernel here to <0.5 %).
"""
import collections
import csv
import glob
import json
import os
import re
import sys

# kernel-name fragment -> (bench pass name(s), FETCH_SIZE factor, note)
KERNELS = {
    "downsample_kernel": ("downsample", 2.0, "reads are one 16 B/lane stream: FETCH_SIZE x2 (calibrated: equals 4*W*H bytes/frame)"),
    "render_kernel": ("render", 2.0, "window fill is 16 B/lane: FETCH_SIZE x2"),
    "upsample_kernel<A, false, true": ("upsample_L1_to_L0", 1.0, "8 B/lane (f16 depth) + 16 B/lane + 4 B/lane reads: FETCH_SIZE left raw (uncalibrated width); raw value equals compulsory + apron bytes"),
    "upsample_kernel<A, false, false": ("upsample_blend_passes", 1.0, "mean of the stand-alone main_blendout launches (L2->L1 only when L4->L3 rides inside L3->L2); FETCH_SIZE raw"),
    "upsample_two_level_kernel<A, false": ("upsample_L4_to_L3+L3_to_L2", 1.0, "the fused two-level launch; FETCH_SIZE raw"),
    # the last upsample kernel carrying the next batch's downsample pass (meao_prefetch_batch): the carried
    # 16 B/lane depth stream (4*W*H bytes per frame, known exactly) is tallied at half size like in
    # downsample_kernel, the upsample reads are raw -> add the missing half of the depth stream
    "upsample_final_with_next_downsample_kernel<A, false": ("upsample_L1_to_L0+downsample_next", 1.0,
                                                           "FETCH_SIZE raw + 2*W*H bytes per frame (the half of the carried 16 B/lane depth stream that the counter misses)"),
}
DEPTH_STREAM_HALF = {"4k": 2 * 3840 * 2160, "1080p": 2 * 1920 * 1080, "8k": 2 * 7680 * 4320}


def mean_counter(root, group, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, group, "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                out[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}


def main():
    root, workload, frames = sys.argv[1], sys.argv[2], int(sys.argv[3])
    fetch, write = mean_counter(root, "fetch", "FETCH_SIZE"), mean_counter(root, "write", "WRITE_SIZE")
    valu = mean_counter(root, "sq1", "SQ_INSTS_VALU")      # VALU wave-instructions per dispatch, whole GPU
    table = {}
    ao = "1" if workload == "8k" else "0"          # AOFMT template argument of the workload's kernels (8K: fp16 AO storage)
    for frag, (name, factor, note) in KERNELS.items():
        frag = frag.replace("<A,", f"<{ao},")
        f = [v for k, v in fetch.items() if frag in k]
        w = [v for k, v in write.items() if frag in k]
        if not f or not w:
            continue
        fb, wb = f[0] * 1024 * factor, w[0] * 1024
        if "downsample_next" in name:
            fb += DEPTH_STREAM_HALF[workload] * frames
        table[name] = {"bytes_per_frame": round((fb + wb) / frames), "fetch_bytes_per_frame": round(fb / frames),
                       "write_bytes_per_frame": round(wb / frames), "fetch_size_factor": factor,
                       "frames_per_launch": frames, "note": note}
        v = [x for k, x in valu.items() if frag in k]
        if v:
            table[name]["valu_wave_insts_per_frame"] = round(v[0] / frames)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    try:
        full = json.load(open(path))
    except OSError:
        full = {}
    full[workload] = table
    full.setdefault("_tags", {"4k": full.get("_tag", "(untagged)")} if "4k" in full and workload != "4k" else {})
    full["_tags"][workload] = os.path.basename(os.path.normpath(root)).replace("pmc_", "")
    full["_tag"] = full["_tags"].get("4k", full["_tags"][workload])
    full["_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes (tests/run_pmc.sh); see tests/make_pmc_traffic.py"
    json.dump(full, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(table, indent=1))


if __name__ == "__main__":
    main()
