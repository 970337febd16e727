#!/usr/bin/env python3
"""Every kernel of the last complete training step in a rocprofv3 kernel trace, in order, with the idle time in
front of it: python tools/step_timeline.py <kernel_trace.csv> [--all]   (a step ends with the last adam_kernel)."""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    # the vanilla step ends with two Adam launches (coarse and fine model share one optimiser call per bucket)
    b = idx[-1]
    a = max(i for i in idx if b - i > 8)
    step = rows[a + 1:b + 1]
    t0 = int(rows[a]["End_Timestamp"])
    print("step wall %.1f us, %d kernels" % ((int(rows[b]["End_Timestamp"]) - t0) / 1e3, len(step)))
    by = collections.OrderedDict()
    prev = t0
    gaps = 0
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("void ", "")[:70]
        if "--all" in sys.argv:
            print("%8.1f gap %9.1f us  %s" % ((s - prev) / 1e3, (e - s) / 1e3, name))
        k = by.setdefault(name, [0, 0.0])
        k[0] += 1
        k[1] += (e - s) / 1e3
        gaps += max(0, s - prev)
        prev = max(prev, e)
    for name, (n, us) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print("%4d x %10.1f us  %s" % (n, us, name))
    print("kernel time %.1f us, idle between kernels %.1f us" % (sum(v[1] for v in by.values()), gaps / 1e3))


if __name__ == "__main__":
    main()
