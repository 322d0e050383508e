#!/usr/bin/env python3
"""One period of a periodic kernel trace as a timeline: python tools/trace_timeline.py <dir with *_kernel_trace.csv> <first_kernel_substring> [which_period]
Prints every kernel of that period: queue, start offset (us), duration (us), name -- to read the critical path of a multi-stream step."""
import csv, glob, os, sys


def main():
    d, first = sys.argv[1], sys.argv[2]
    which = int(sys.argv[3]) if len(sys.argv) > 3 else -3
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
    a, b = starts[which], starts[which + 1]
    t0 = int(rows[a]["Start_Timestamp"])
    qs = {}
    print(f"period {which}: {b - a} kernels, {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
    for r in rows[a:b]:
        q = qs.setdefault(r["Queue_Id"], len(qs))
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print(f"  q{q} {s / 1e3:8.1f} +{(e - s) / 1e3:6.1f}  {'    ' * q}{r['Kernel_Name'].split('(')[0][:70]}")


if __name__ == "__main__":
    main()
