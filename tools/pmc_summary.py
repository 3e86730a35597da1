"""Per-kernel means of every counter in a tools/run_pmc.sh output directory.

    python tools/pmc_summary.py gpurun_out/pmc_<tag>
"""
import collections
import csv
import glob
import re
import sys


def kernel_key(name):
    m = re.search(r"(\w+_kernel<[^>]*>)", name)
    return m.group(1) if m else name[:48]


def main():
    out = sys.argv[1]
    for f in sorted(glob.glob(out + "/*/*counter_collection.csv")):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = kernel_key(row["Kernel_Name"])
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[(k, row["Counter_Name"])] += 1
        print("==", f)
        for k, d in sorted(agg.items()):
            print(k, {c: round(v / cnt[(k, c)], 1) for c, v in sorted(d.items())})


if __name__ == "__main__":
    main()
