"""profiles/roofline_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/collect_profiles.sh.
    python tools/make_traffic_json.py <out_dir of collect_profiles.sh> > profiles/roofline_traffic.json
Per kernel, only the fine-pass launches count (786,432 points): the largest grid and, for the weight-gradient
kernel, whose grid does not depend on the size, the dispatches within 10 % of the largest counter value."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import mlp_source_sha  # noqa: E402  (the sources the passes were collected from: bench.py checks it)

out_dir = sys.argv[1]
KERNELS = ("wgrad_kernel", "vanilla_fwd_kernel", "vanilla_bwd_data_kernel")


def fine_mean(kind, kernel):
    path = glob.glob(f"{out_dir}/{kind}/*/*_counter_collection.csv")[0]
    rows = [(int(r["Grid_Size"]), float(r["Counter_Value"])) for r in csv.DictReader(open(path))
            if ("nerf::" + kernel) in r["Kernel_Name"]]
    grid = max(g for g, _ in rows)
    vals = [v for g, v in rows if g == grid]
    top = max(vals)
    vals = [v for v in vals if v >= 0.9 * top]
    return sum(vals) / len(vals), len(vals)


def fine_counter(kind, kernel, counter):
    """Mean of one counter of a multi-counter pass over the fine-pass launches (the largest values of the largest grid)."""
    path = glob.glob(f"{out_dir}/{kind}/*/*_counter_collection.csv")[0]
    rows = [(int(r["Grid_Size"]), float(r["Counter_Value"])) for r in csv.DictReader(open(path))
            if ("nerf::" + kernel) in r["Kernel_Name"] and r["Counter_Name"] == counter]
    grid = max(g for g, _ in rows)
    vals = [v for g, v in rows if g == grid]
    top = max(vals)
    vals = [v for v in vals if v >= 0.8 * top]
    return sum(vals) / len(vals)


SIMDS, XCDS = 1024, 8   # SQ counters are sums over the chip's SIMDs, GRBM_GUI_ACTIVE over its 8 XCDs

res = {"_comment": "HBM bytes per launch at the fine-pass size (786,432 points = 4096 rays x 192 samples), from rocprofv3 --pmc "
                   "FETCH_SIZE and --pmc WRITE_SIZE passes over bench.py (tools/collect_profiles.sh; tools/make_traffic_json.py "
                   "keeps the fine-pass dispatches only). Counter unit is KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md "
                   "prescribes for 16-byte-per-lane streaming reads on gfx950. mfma_busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / "
                   "(GRBM_GUI_ACTIVE / 8 XCDs) of the same launches: the share of the kernel's clocks during which a SIMD's "
                   "matrix pipe was busy.",
       "round": 3, "points_per_launch": 786432, "source_sha256": mlp_source_sha()}
for k in KERNELS:
    f, nf = fine_mean("fetch", k)
    w, _ = fine_mean("write", k)
    res[k] = {"fetch_size_kib": round(f), "write_size_kib": round(w), "dispatches": nf, "hbm_bytes": int((2 * f + w) * 1024)}
    try:
        busy, active = fine_counter("mfma", k, "SQ_VALU_MFMA_BUSY_CYCLES"), fine_counter("mfma", k, "GRBM_GUI_ACTIVE")
        res[k]["mfma_busy"] = round((busy / SIMDS) / (active / XCDS), 4)
    except (IndexError, ValueError, ZeroDivisionError):
        pass
print(json.dumps(res, indent=1))
