"""Per-kernel launch durations from a rocprofv3 --kernel-trace CSV, split into "all launches" and
"the last N launches" (= bench.py's timed region: the process also runs an untimed clock-ramp phase
and the warm-up steps first, which the --stats average includes).

    python tools/rocprof_timed_region.py gpurun_out/prof_<tag>/trace_kernel_trace.csv [N=30]
"""
import collections
import csv
import re
import sys


def main():
    path, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        m = re.search(r"(\w+_kernel<[^>]*>)", r["Kernel_Name"])
        by[m.group(1) if m else r["Kernel_Name"][:48]].append(
            (int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    print(f"{'kernel':44s} {'launches':>8s} {'avg all us':>11s} {'avg last %d us' % n:>15s} {'avg first 10 us':>16s}")
    for k, v in sorted(by.items()):
        v.sort()
        d = [x[1] for x in v]
        groups = [("", d)]
        if k.startswith("upsample_kernel") and ", false," in k.split("<")[1][3:] and not any(
                x.startswith("upsample_two_level_kernel") for x in by):
            # without the fused two-level launch the three blend passes alternate in this one kernel
            groups = [(" " + name, d[j::3]) for j, name in enumerate(("L4->L3", "L3->L2", "L2->L1"))]
        for name, dd in groups:
            us = lambda xs: sum(xs) / max(len(xs), 1) / 1e3    # noqa: E731
            print(f"{k + name:44s} {len(dd):8d} {us(dd):11.1f} {us(dd[-n:]):15.1f} {us(dd[:10]):16.1f}")


if __name__ == "__main__":
    main()
