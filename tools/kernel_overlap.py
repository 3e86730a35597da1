#!/usr/bin/env python3
"""Which kernels ran at the same time, from a rocprofv3 --kernel-trace CSV: evidence for what two pool members on ONE device
overlap (bench.py --pool 2) and a single context cannot.

    python tools/kernel_overlap.py <trace_kernel_trace.csv> [last N dispatches = 400]

Prints, over the last N dispatches: the span, the time with >= 1 / >= 2 kernels resident, and for every pair of kernel
families the time both were resident (by queue, so that "member A's last kernel next to member B's render" is visible)."""
import collections, csv, re, sys


def family(name):
    m = re.search(r"(\w+_kernel)<", name)
    return m.group(1) if m else name.split("(")[0][:40]


def main():
    path, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 400
    rows = sorted(({"k": family(r["Kernel_Name"]), "q": r.get("Queue_Id", "?"), "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])}
                   for r in csv.DictReader(open(path))), key=lambda r: r["s"])[-n:]
    if not rows:
        print("no dispatches")
        return
    t0, t1 = rows[0]["s"], max(r["e"] for r in rows)
    events = sorted([(r["s"], 1, i) for i, r in enumerate(rows)] + [(r["e"], -1, i) for i, r in enumerate(rows)])
    live, last = set(), t0
    busy1 = busy2 = 0
    pair = collections.Counter()
    for t, kind, i in events:
        dt = t - last
        if dt > 0:
            if len(live) >= 1:
                busy1 += dt
            if len(live) >= 2:
                busy2 += dt
                fams = sorted({(rows[j]["k"], rows[j]["q"]) for j in live})
                for a in range(len(fams)):
                    for b in range(a + 1, len(fams)):
                        pair[(fams[a][0], fams[b][0], fams[a][1] != fams[b][1])] += dt
        last = t
        (live.add if kind == 1 else live.discard)(i)
    queues = sorted({r["q"] for r in rows})
    print(f"{len(rows)} dispatches on queues {queues}, span {(t1 - t0) / 1e3:.1f} us")
    print(f"  >= 1 kernel resident: {busy1 / 1e3:9.1f} us ({100.0 * busy1 / (t1 - t0):.1f} % of the span)")
    print(f"  >= 2 kernels resident: {busy2 / 1e3:8.1f} us ({100.0 * busy2 / (t1 - t0):.1f} % of the span)")
    print("  time two kernels were resident together (us), most first:")
    for (a, b, other_queue), dt in pair.most_common(12):
        print(f"    {dt / 1e3:9.1f}  {a}  +  {b}" + ("   [different queues]" if other_queue else "   [same queue]"))
    per = collections.defaultdict(list)
    for r in rows:
        per[r["k"]].append(r["e"] - r["s"])
    print("  mean duration while sharing the device (us):")
    for k, v in sorted(per.items()):
        print(f"    {sum(v) / len(v) / 1e3:9.1f}  x{len(v):4d}  {k}")


if __name__ == "__main__":
    main()
