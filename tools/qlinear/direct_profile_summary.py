#!/usr/bin/env python3
"""Summarise gpurun_out/direct_prof (tools/qlinear/direct_profile.sh): per kernel, mean duration and mean counter values per launch."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
base = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "direct_prof")


def short(name):
    if "qdirect_kernel" in name:
        return name[name.index("qdirect_kernel"):].split("(")[0]
    return name.split("(")[0][:70]


out = {}
for f in glob.glob(os.path.join(base, "trace", "*", "*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        out.setdefault(short(r["Name"]), {})["us_avg"] = round(float(r["AverageNs"]) / 1e3, 2)
        out[short(r["Name"])]["calls"] = int(r["Calls"])
for sub in ("tcc", "fetch", "sq"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(base, sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        for c, v in d.items():
            out.setdefault(k, {})[c] = round(sum(v) / len(v), 1)
for k, d in out.items():
    if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
        d["l2_hit_rate"] = round(d["TCC_HIT_sum"] / max(1.0, d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 4)
    if "FETCH_SIZE" in d:
        d["fabric_read_MB_x2_gfx950"] = round(d["FETCH_SIZE"] * 2 * 1024 / 1e6, 1)   # FETCH_SIZE is in KiB; gfx950 reports half (MI355X_MICROARCH.md)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d:
        d["mfma_busy_frac_of_wave_cycles"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1.0, d.get("SQ_WAVE_CYCLES", 1.0) * 4), 4)
print(json.dumps(out, indent=1))
