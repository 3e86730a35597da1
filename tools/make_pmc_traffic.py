"""Turn the rocprofv3 --pmc passes of tools/run_pmc.sh into profiles/pmc_traffic.json.

    python tools/make_pmc_traffic.py gpurun_out/pmc_<tag> <workload> <frames_per_launch>

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  Calibration (round 5, tools/ubench_fetch.hip under the same
counters, profiles/r05_pmc_calibration.json): on gfx950 FETCH_SIZE reports exactly HALF of the bytes a kernel reads, for
EVERY access width -- 1, 2, 4, 8 and 16 bytes per lane, temporal or not: 0.5000 of a 1 GiB stream in all nine cases (and
0.81 of the bytes USED where 160 of every 256 bytes are touched: whole 128-byte lines move).  Rounds 2-4 doubled only the
16 B/lane streams, as MI355X_MICROARCH.md describes the effect, and left the upsample kernels' reads raw: those rows were
under-counted.  Every FETCH_SIZE is doubled now.  WRITE_SIZE is exact (1.000 - 1.004 of a 1 GiB stream; 1.04 for
single-byte stores) and used as reported.  Residual after the correction, on the one kernel whose bytes are known
exactly: downsample_kernel reads 4*W*H bytes per frame -- corrected counter / known = 1.0002.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# kernel-name fragment -> (bench pass name(s), FETCH_SIZE factor, note)
FETCH_FACTOR = 2.0       # profiles/r05_pmc_calibration.json: the counter reports half of the bytes read, at every access width
KERNELS = {
    "downsample_kernel": ("downsample", FETCH_FACTOR, "FETCH_SIZE x2 (calibrated); the pass reads the even rows of the frame: 2*W*H bytes of f32 depth"),
    "render_kernel": ("render", FETCH_FACTOR, "FETCH_SIZE x2 (calibrated)"),
    "upsample_final_kernel<A, false": ("upsample_L1_to_L0", FETCH_FACTOR, "FETCH_SIZE x2 (calibrated); HiResDB is evaluated from the raw depth frame (4*W*H bytes read)"),
    "upsample_kernel<A, false": ("upsample_blend_passes", FETCH_FACTOR, "mean of the stand-alone main_blendout launches (L2->L1 only when L4->L3 rides inside L3->L2); FETCH_SIZE x2 (calibrated)"),
    "upsample_blend_tall_kernel<A, false": ("upsample_blend_passes", FETCH_FACTOR, "the L2->L1 launch with 64 x 64 tiles (large R8 batches); FETCH_SIZE x2 (calibrated)"),
    "upsample_two_level_kernel<A, false": ("upsample_L4_to_L3+L3_to_L2", FETCH_FACTOR, "the fused two-level launch; FETCH_SIZE x2 (calibrated)"),
    "upsample_final_with_next_downsample_kernel<A, false": ("upsample_L1_to_L0+downsample_next", FETCH_FACTOR,
                                                           "FETCH_SIZE x2 (calibrated; rounds 2-4 doubled only the carried depth stream)"),
}


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
    # the device code the counters were collected from (run_pmc.sh writes it on the GPU box; bench.py compares it with what it loads)
    from miniengineao_amd import codehash, _lib
    try:
        sha = open(os.path.join(root, "code_sha256.txt")).read().split()[0]
    except (OSError, IndexError):
        sha = codehash.device_code_sha256(_lib.LIB_PATH)
    full.setdefault("_code_sha256", {})[workload] = sha
    full["_source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes (tools/run_pmc.sh); FETCH_SIZE x2 for every kernel "
                       "(profiles/r05_pmc_calibration.json); see tools/make_pmc_traffic.py")
    json.dump(full, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(table, indent=1))


if __name__ == "__main__":
    main()
