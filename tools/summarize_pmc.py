"""Per-kernel, per-grid-size means of rocprofv3 --pmc counters.
    python tools/summarize_pmc.py <counter_collection.csv> [...]
"""
import collections
import csv
import sys

for path in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "nerf::" not in name:
            continue
        key = (name, int(r["Grid_Size"]) if "Grid_Size" in r else 0)
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("#", path)
    for key in sorted(agg):
        vals = {c: sum(v) / len(v) for c, v in agg[key].items()}
        print("%-40s grid %9d calls %3d  " % (key[0], key[1], len(next(iter(agg[key].values())))) +
              "  ".join("%s=%.4g" % (c, v) for c, v in sorted(vals.items())))
