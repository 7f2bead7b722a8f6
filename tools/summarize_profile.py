#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh run (gpurun_out/prof_<tag>) into the committed evidence under profiles/.

    python tools/summarize_profile.py <tag>

Writes
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `bench.py --steps 50 --warmup 5`
  profiles/<tag>_pmc_traffic.json   FETCH_SIZE / WRITE_SIZE per launch (separate --pmc passes), with the
                                    calibration on kernels of known byte count and the gfx950 correction
  profiles/traffic.json             HBM bytes per launch for bench.py's `roofline.traffic`

Correction (MI355X_MICROARCH.md §HBM): on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
streaming read -> doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  Both are in KiB.
The calibration rows (plain copy / read-only / write-only of a 90,177,536-byte buffer in tools/kbench) confirm both.
"""
import collections
import csv
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOWN = 4096 * 11008 * 2


def counters(path, by_grid=False):
    """counter values per kernel; by_grid: key = (kernel name, grid size) so that paired (two-tensor) launches of the
    same kernel are kept apart from single-tensor ones"""
    d = collections.defaultdict(list)
    for f in glob.glob(os.path.join(path, "runc", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            d[(r["Kernel_Name"], int(r["Grid_Size"])) if by_grid else r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d


def ac_flag(name):
    """last template argument of row_reg_kernel (AC: 0 = plain arithmetic, 1 = autocast)"""
    import re
    m = re.search(r"row_reg_kernel<[^>]*,\s*(\d)>", name)
    return int(m.group(1)) if m else None


def short(name):
    return name.replace("void ", "").split("(")[0]


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)

    # 1. kernel-trace stats
    stats = glob.glob(os.path.join(src, "trace", "runc", "*_kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --core-extras  (MI355X, gfx950)"])
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"])[:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])

    # per-dispatch durations of the STE kernel split by safe/unsafe cannot be told from the stats file; use the trace
    trace = glob.glob(os.path.join(src, "trace", "runc", "*_kernel_trace.csv"))[0]
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        if "fq::" in r["Kernel_Name"]:
            dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))

    # 1b. the same trace split by grid size: paired (two-tensor) launches of a kernel are twice as long as single ones
    by_grid = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        if "fq::" in r["Kernel_Name"]:
            by_grid[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(os.path.join(out, f"{tag}_kernel_stats_by_grid.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["# same rocprofv3 kernel trace, fq:: kernels split by grid size (the larger grid of a kernel = the paired weight+input launch)"])
        w.writerow(["Name", "Grid_Size_X", "Calls", "AverageNs", "MinNs", "MaxNs"])
        for (name, grid), v in sorted(by_grid.items()):
            lo_, hi_ = min(v), max(v)
            cut = (lo_ + hi_) / 2
            a, b = [x for x in v if x < cut], [x for x in v if x >= cut]
            if hi_ > 1.5 * lo_ and min(len(a), len(b)) >= 10:   # two populations behind one kernel name and grid (e.g. the paired STE
                # backward with the weight's gradient in place vs copied): report them apart
                w.writerow([name + "  [faster population]", grid, len(a), round(statistics.mean(a), 1), min(a), max(a)])
                w.writerow([name + "  [slower population]", grid, len(b), round(statistics.mean(b), 1), min(b), max(b)])
            else:
                w.writerow([name, grid, len(v), round(statistics.mean(v), 1), lo_, hi_])

    # 2. PMC traffic
    res = {"unit": "bytes per launch", "correction": "FETCH_SIZE KiB x 2 (gfx950 wide-read undercount), WRITE_SIZE KiB x 1",
           "calibration": {}, "kernels": {}}
    kbf, kbw = counters(os.path.join(src, "kb_fetch")), counters(os.path.join(src, "kb_write"))
    for name in kbf:
        if any(k in name for k in ("copy_kernel<4, false>", "copy_kernel<4, true>", "read_kernel<4>", "write_kernel<4>")):
            res["calibration"][short(name)] = {
                "known_read_bytes": 0 if "write_kernel" in name else KNOWN, "known_write_bytes": 0 if "read_kernel" in name else KNOWN,
                "FETCH_SIZE_KiB_raw": statistics.median(kbf[name]), "WRITE_SIZE_KiB_raw": statistics.median(kbw.get(name, [0])),
                "read_bytes_corrected": statistics.median(kbf[name]) * 2 * 1024, "write_bytes": statistics.median(kbw.get(name, [0])) * 1024}
    bf, bw = counters(os.path.join(src, "fetch"), True), counters(os.path.join(src, "write"), True)
    single_grid = {}
    for (name, grid) in bf:   # the smallest grid of a kernel = its single-tensor launches
        single_grid[name] = min(grid, single_grid.get(name, grid))
    traffic = {}

    def split(vals):
        """bimodal counter values -> (low cluster, high cluster); one cluster if the spread is < 1 %"""
        lo, hi = min(vals), max(vals)
        if hi - lo < 0.01 * hi:
            return vals, []
        cut = (lo + hi) / 2
        return [v for v in vals if v < cut], [v for v in vals if v >= cut]

    for (name, grid) in sorted(bf):
        if "fq::" not in name:
            continue
        fv, wv = bf[(name, grid)], bw.get((name, grid), [])
        entry = {"launches": len(fv), "grid_size": grid}
        if grid != single_grid[name]:   # a paired launch (weight + input of a QuantizeLinear)
            if "ste_mask_kernel" in name:
                # two populations share this kernel and grid: the step's backward with the weight's gradient IN PLACE (the W4
                # half moves nothing) and the copying variant (both gradients to fresh tensors)
                f_lo, f_hi = split(fv)
                w_lo, w_hi = split(wv) if wv else ([], [])
                med = statistics.median
                lo_total = med(f_lo) * 2048 + (med(w_lo) * 1024 if w_lo else 0)
                entry.update({"paired_launch": True})
                if f_hi:
                    hi_total = med(f_hi) * 2048 + (med(w_hi if w_hi else w_lo) * 1024 if (w_hi or w_lo) else 0)
                    entry["weight_gradient_in_place"] = {"read_bytes": med(f_lo) * 2048, "write_bytes": med(w_lo) * 1024 if w_lo else 0, "total": lo_total, "launches": len(f_lo)}
                    entry["both_to_fresh_tensors"] = {"read_bytes": med(f_hi) * 2048, "write_bytes": med(w_hi if w_hi else w_lo) * 1024, "total": hi_total, "launches": len(f_hi)}
                    traffic["ste_bwd_pair_w4a8_inplace"] = lo_total
                    traffic["ste_bwd_pair_w4a8"] = hi_total
                else:
                    entry.update({"read_bytes": med(f_lo) * 2048, "write_bytes": med(w_lo) * 1024 if w_lo else 0, "total": lo_total})
                    traffic["ste_bwd_pair_w4a8"] = lo_total
                res["kernels"][short(name) + f" [grid {grid}]"] = entry
                continue
            rd, wr = statistics.median(fv) * 2048, (statistics.median(wv) * 1024 if wv else 0)
            entry.update({"paired_launch": True, "read_bytes": rd, "write_bytes": wr, "total": rd + wr})
            if "row_reg_kernel" in name and ac_flag(name) == 0:
                traffic["sym_fwd_pair_w4a8"] = rd + wr
            res["kernels"][short(name) + f" [grid {grid}]"] = entry
            continue
        f_lo, f_hi = split(fv)
        w_lo, w_hi = split(wv) if wv else ([], [])
        med = statistics.median
        if "ste_rows" in name or "ste_mask" in name:
            # bimodal on the READ side: rows provably unclipped (g only) vs rows that need x / the bit mask
            wr = med(wv) * 1024 if wv else 0
            entry["rows_safe"] = {"read_bytes": med(f_lo) * 2048, "write_bytes": wr, "total": med(f_lo) * 2048 + wr, "launches": len(f_lo)}
            traffic["ste_bwd_w4"] = entry["rows_safe"]["total"]
            if f_hi:
                entry["rows_clippable"] = {"read_bytes": med(f_hi) * 2048, "write_bytes": wr, "total": med(f_hi) * 2048 + wr, "launches": len(f_hi)}
                traffic["ste_bwd_a8"] = entry["rows_clippable"]["total"]
        elif "row_reg_kernel" in name and ac_flag(name) != 0:
            rd = med(fv) * 2048
            wr = med(wv) * 1024 if wv else None
            entry.update({"read_bytes": rd, "write_bytes": wr, "total": rd + (wr or 0), "note": "autocast arithmetic"})
        elif "row_reg_kernel" in name:
            # bimodal on the WRITE side in training mode: the A8 leg also writes the 1-bit STE mask
            rd = med(fv) * 2048
            entry["no_mask_rows"] = {"read_bytes": rd, "write_bytes": med(w_lo) * 1024 if w_lo else None, "launches": len(w_lo)}
            traffic["sym_fwd_w4"] = rd + (med(w_lo) * 1024 if w_lo else 0)
            traffic["sym_fwd_a8"] = traffic["sym_fwd_w4"]
            if w_hi:
                entry["mask_rows"] = {"read_bytes": rd, "write_bytes": med(w_hi) * 1024, "launches": len(w_hi)}
                traffic["sym_fwd_a8"] = rd + med(w_hi) * 1024
        else:
            rd = med(fv) * 2048
            wr = med(wv) * 1024 if wv else None
            entry.update({"read_bytes": rd, "write_bytes": wr, "total": rd + (wr or 0)})
            if "ste_vec_kernel" in name:
                traffic["ste_bwd_a8_xread"] = entry["total"]
        d = dur.get(short(name))
        if d:
            entry["avg_duration_ns_unprofiled_trace"] = statistics.mean(d)
        res["kernels"][short(name) + f" [grid {grid}]"] = entry
    if "sym_fwd_w4" in traffic:
        traffic["sym_fwd_w4_plain"] = traffic["sym_fwd_w4"]
    json.dump(res, open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    import time
    traffic["_source"] = (f"profiles/{tag}_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, FETCH_SIZE x2 per the gfx950 "
                          f"rule) of `bench.py --steps 50 --warmup 5 --no-cpu-baseline --core-extras`, collected {time.strftime('%Y-%m-%d')} on one MI355X; "
                          "NOT measured in the run that prints this line")
    json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))
    for k, v in dur.items():
        print(f"{k[:90]:92s} n={len(v):4d} avg={statistics.mean(v)/1e3:8.2f} us  min={min(v)/1e3:8.2f}")


if __name__ == "__main__":
    main()
