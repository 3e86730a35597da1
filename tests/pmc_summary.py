"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch)."""
import collections
import csv
import glob
import re
import sys


def short(name):
    m = re.search(r"(\w+_kernel<[^>]*>)", name)
    return m.group(1) if m else name[:60]


def main(root):
    for f in sorted(glob.glob(root + "/*/*counter_collection.csv")):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[(k, row["Counter_Name"])] += 1
        print("==", f)
        for k, d in sorted(agg.items()):
            print(f"{k:44s}", {c: round(v / cnt[(k, c)], 1) for c, v in sorted(d.items())}, "dispatches", cnt[(k, next(iter(d)))])


if __name__ == "__main__":
    main(sys.argv[1])
