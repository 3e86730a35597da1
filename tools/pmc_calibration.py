#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of the streams of tools/ubench_fetch.hip (byte counts known exactly) -> what fraction of the
moved bytes each counter reports, per access width.  profiles/README.md says how the factors are applied to
profiles/pmc_traffic.json.

    python tools/pmc_calibration.py gpurun_out/pmc_calibration_<tag> [MiB of the buffer, default 1024]  > profiles/r05_pmc_calibration.json
"""
import collections, csv, glob, json, os, re, sys

root = sys.argv[1]
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
BYTES = mib << 20


def counters(group, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, group, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                out[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return out


def known_bytes(kernel):
    if "read_window_kernel" in kernel:
        return BYTES // 256 * 160
    if "tile_rows_kernel" in kernel:        # image-shaped streams: 3840 x 262144 bytes (write), 7680 x 131072 (read)
        return 3840 * 262144
    return BYTES


rows = []
for group, counter, prefix in (("fetch", "FETCH_SIZE", "read_"), ("write", "WRITE_SIZE", "write_")):
    for kernel, values in sorted(counters(group, counter).items()):
        if prefix not in kernel:
            continue
        name = re.sub(r"^void |\(.*$", "", kernel)
        reported = sum(values) / len(values) * 1024.0          # the counters are in KiB per dispatch
        rows.append({"kernel": name, "counter": counter, "launches": len(values), "known_bytes": known_bytes(kernel),
                     "reported_bytes": round(reported), "reported_over_known": round(reported / known_bytes(kernel), 4)})
print(json.dumps({"buffer_MiB": mib, "rows": rows,
                  "how": "tools/ubench_fetch.hip under rocprofv3 --kernel-trace --pmc FETCH_SIZE (and WRITE_SIZE in its own pass); "
                         "every kernel touches its buffer exactly once"}, indent=1))
