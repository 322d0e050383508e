#!/usr/bin/env python3
"""Per-kernel durations AND the gaps between consecutive kernels from a `rocprofv3 --kernel-trace --output-format csv` run:
    python tools/trace_gaps.py <dir with *_kernel_trace.csv> [first_kernel_substring] [last_n_forwards]
A "forward" starts at every kernel whose name contains first_kernel_substring (default k_lut_ids); prints, per position inside a forward,
the kernel, its average duration and the average gap to the next kernel's start, and the average span of a whole forward."""
import csv, glob, os, sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else "k_lut_ids"
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    fwd, cur = [], None
    for r in rows:
        if first in r["Kernel_Name"]:
            cur = []
            fwd.append(cur)
        if cur is not None:
            cur.append((r["Kernel_Name"].split("(")[0][:60], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    fwd = fwd[:-1][-skip:]                                   # the LAST `skip` forwards (calibration / warm-up forwards come first)
    fwd = [x for x in fwd if len(x) == len(fwd[-1])]
    n = len(fwd[0])
    dur, gap = defaultdict(float), defaultdict(float)
    for x in fwd:
        for i, (name, s, e) in enumerate(x):
            dur[i] += e - s
            if i + 1 < n:
                gap[i] += x[i + 1][1] - e
    span = sum(x[-1][2] - x[0][1] for x in fwd) / len(fwd)
    print(f"{len(fwd)} forwards of {n} kernels; whole span {span / 1e3:.1f} us; sum of durations {sum(dur.values()) / len(fwd) / 1e3:.1f} us, of gaps {sum(gap.values()) / len(fwd) / 1e3:.1f} us")
    for i in range(n):
        print(f"  {i:2d} {fwd[0][i][0]:60s} {dur[i] / len(fwd) / 1e3:7.2f} us   gap after {gap[i] / len(fwd) / 1e3 if i + 1 < n else 0:6.2f} us")


if __name__ == "__main__":
    main()
