"""Per-kernel mean duration in the first and in the last third of a rocprofv3 kernel trace (what changed over a run):
python tools/trace_split.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
parts = {"first third": rows[: n // 3], "last third": rows[2 * n // 3:]}
agg = {k: collections.defaultdict(list) for k in parts}
for k, rs in parts.items():
    for r in rs:
        agg[k][r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
names = sorted(set(agg["first third"]) | set(agg["last third"]), key=lambda x: -sum(agg["last third"].get(x, [0])))
print("%-62s %8s %10s %8s %10s" % ("kernel", "calls", "first us", "calls", "last us"))
for x in names[:25]:
    a, b = agg["first third"].get(x, []), agg["last third"].get(x, [])
    print("%-62s %8d %10.1f %8d %10.1f" % (x, len(a), sum(a) / max(1, len(a)), len(b), sum(b) / max(1, len(b))))
