#!/usr/bin/env python3
"""Condense a tools/profile_pmc_util.sh run (gpurun_out/pmc_<tag>/{sq,tcc,lds}) into profiles/<tag>_pmc_utilization.json:
per fq:: kernel, the median of every counter over its launches (warm-up launches included: the counters are per-launch work,
not time), plus a few derived ratios.

    python tools/summarize_pmc_util.py <tag>
"""
import collections
import csv
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    src = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}")
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in ("sq", "tcc", "lds"):
        files = glob.glob(os.path.join(src, p, "*", "*_counter_collection.csv"))
        if not files:
            print(f"no counter file for pass {p}", file=sys.stderr)
            continue
        for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            if "fq::" in k:
                per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"command": "rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras (three separate passes)",
           "note": "medians per launch on MI355X; SQ_* cycle counters are summed over waves (MI355X_MICROARCH.md). wait_any_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES "
                   "(share of a wave's lifetime parked on s_waitcnt / barrier), active_inst_frac = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES (issuing), "
                   "valu_busy_of_active = SQ_ACTIVE_INST_VALU / SQ_ACTIVE_INST_ANY",
           "kernels": {}}
    for k, cs in sorted(per.items()):
        m = {c: statistics.median(v) for c, v in sorted(cs.items())}
        d = {}
        if m.get("SQ_WAVE_CYCLES"):
            d["wait_any_frac"] = round(m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"], 3)
            d["active_inst_frac"] = round(m.get("SQ_ACTIVE_INST_ANY", 0) / m["SQ_WAVE_CYCLES"], 3)
        if m.get("SQ_ACTIVE_INST_ANY"):
            d["valu_busy_of_active"] = round(m.get("SQ_ACTIVE_INST_VALU", 0) / m["SQ_ACTIVE_INST_ANY"], 3)
        if m.get("SQ_WAVES"):
            d["valu_insts_per_wave"] = round(m.get("SQ_INSTS_VALU", 0) / m["SQ_WAVES"], 1)
        if m.get("TCC_HIT_sum") is not None and (m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0)) > 0:
            d["l2_hit_rate"] = round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 3)
        m["launches"] = max(len(v) for v in cs.values())
        m["derived"] = d
        out["kernels"][k] = m
    dst = os.path.join(ROOT, "profiles", f"{tag}_pmc_utilization.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    for k, m in out["kernels"].items():
        print(k[:90], m["derived"])
    print("wrote", dst)


if __name__ == "__main__":
    main()
