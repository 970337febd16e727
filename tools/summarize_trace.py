"""Per-kernel, per-grid-size averages from a rocprofv3 --kernel-trace CSV.

    python tools/summarize_trace.py gpurun_out/<dir>/<pid>_kernel_trace.csv

The coarse (64-sample) and fine (192-sample) passes launch the same kernels at
different grid sizes; the stats CSV averages them together, this splits them.
"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "nerf::" not in name:
        continue
    agg[(name, int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("%-44s %10s %6s %12s %12s" % ("kernel", "grid_thr", "calls", "avg_us", "min_us"))
for (name, grid), v in sorted(agg.items()):
    print("%-44s %10d %6d %12.1f %12.1f" % (name, grid, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3))
