#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh run (gpurun_out/prof_<tag>) into the committed evidence under profiles/.

    python tools/summarize_profile.py <tag>

bench.py prints, in both profiled modes, a `profile_manifest`: every fq:: launch the process made, in order, as
(entry name, number of launches).  rocprofv3's per-dispatch rows (kernel trace / counter collection) are sorted by
dispatch id, filtered to fq:: kernels and cut into one segment per entry by those counts -- no guessing from kernel names
or grid sizes, and two roles that share a kernel and a grid are still told apart.  A count mismatch aborts.

Writes
  profiles/<tag>_kernel_stats.csv            rocprofv3 --kernel-trace --stats summary of `bench.py --core-extras` (per kernel)
  profiles/<tag>_step_kernels.json           per ENTRY of that run: rocprofv3 average duration, FETCH_SIZE / WRITE_SIZE bytes
  profiles/<tag>_model_shapes_profile.json   the same per entry of `bench.py --model-shapes` (+ bench.py's own HIP-event time)
  profiles/<tag>_model_shapes_kernel_stats.csv
  profiles/<tag>_pmc_calibration.json        the counters on kernels of known byte count (tools/kbench ... ceilings)
  profiles/traffic.json, profiles/traffic_model_shapes.json   HBM bytes per launch keyed by entry name, for bench.py's `traffic`

Correction (MI355X_MICROARCH.md, HBM): on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming
read (16 B per lane) -> doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  Both are in KiB.  Other access
widths are uncalibrated by the guide: the 8-byte-per-lane kernels of tools/kbench calibrate them here, and entries whose
kernel uses such accesses carry the note.
"""
import collections
import csv
import glob
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOWN = 4096 * 11008 * 2


def short(name):
    return name.replace("void ", "").split("(")[0]


def rows_of(path, pattern):
    """the newest file of this kind in the pass's directory (gpurun merges a re-run's files next to an earlier run's: one process = one file)"""
    files = glob.glob(os.path.join(path, "*", pattern))
    if not files:
        return []
    return list(csv.DictReader(open(max(files, key=os.path.getmtime))))


def fq_dispatches(path, counters):
    """fq:: dispatches in launch order: [(kernel, value)] -- value = counter value (counters) or duration in ns (trace)"""
    if counters:
        rows = rows_of(path, "*_counter_collection.csv")
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        return [(short(r["Kernel_Name"]), float(r["Counter_Value"])) for r in rows if "fq::" in r["Kernel_Name"]]
    rows = rows_of(path, "*_kernel_trace.csv")
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [(short(r["Kernel_Name"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if "fq::" in r["Kernel_Name"]]


def manifest_of(log):
    """the profile_manifest bench.py printed (stdout of the profiled run)"""
    for line in reversed(open(log).read().splitlines()):
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        if d.get("bench_extras") == "profile_manifest":
            return d["data"], None
        if "profile_manifest" in d:
            return d["profile_manifest"], d
    raise SystemExit(f"no profile_manifest in {log}")


def segments(dispatches, manifest, what):
    total = sum(n for _, n in manifest)
    if total != len(dispatches):
        raise SystemExit(f"{what}: the manifest announces {total} fq:: launches, the profile holds {len(dispatches)}")
    out, pos = collections.OrderedDict(), 0
    for name, n in manifest:
        if n:
            out[name] = dispatches[pos:pos + n]
        pos += n
    return out


def steady(vals):
    """values of an entry's launches without its warm-up launches (bench.time_launches: warm-up, then 2 x iters timed launches;
    the warm-up is at most a third of the entry)"""
    n_warm = len(vals) // 3
    return vals[n_warm:] if len(vals) > 8 else vals


def per_entry(src, prefix, log_name):
    manifest, doc = manifest_of(os.path.join(src, log_name))
    tr = segments(fq_dispatches(os.path.join(src, prefix + "trace"), False), manifest, prefix + "trace")
    fe = segments(fq_dispatches(os.path.join(src, prefix + "fetch"), True), manifest, prefix + "fetch")
    wr = segments(fq_dispatches(os.path.join(src, prefix + "write"), True), manifest, prefix + "write")
    res = collections.OrderedDict()
    for name in tr:
        if name.startswith(("prologue", "timed region")):
            continue
        kernels = sorted(set(k for k, _ in tr[name]))
        d = [v for _, v in steady(tr[name])]
        f = [v for _, v in steady(fe[name])]
        w = [v for _, v in steady(wr[name])]
        rd, wb = statistics.median(f) * 2048.0, statistics.median(w) * 1024.0
        res[name] = {"kernel": kernels[0] if len(kernels) == 1 else kernels, "launches_profiled": len(d),
                     "rocprof_avg_us": round(statistics.mean(d) / 1e3, 2), "rocprof_min_us": round(min(d) / 1e3, 2),
                     "FETCH_SIZE_KiB_raw": statistics.median(f), "WRITE_SIZE_KiB_raw": statistics.median(w),
                     "read_bytes": rd, "write_bytes": wb, "hbm_bytes": rd + wb}
    return res, manifest, doc


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    cmd_step = "python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --core-extras --no-sidecar --sustain-seconds 0"
    cmd_ms = "python3 bench.py --model-shapes --steps 30"

    # 0. rocprofv3 --stats summaries, per kernel name (as rocprofv3 prints them)
    for sub, dst, cmd in (("trace", f"{tag}_kernel_stats.csv", cmd_step), ("ms_trace", f"{tag}_model_shapes_kernel_stats.csv", cmd_ms)):
        files = glob.glob(os.path.join(src, sub, "*", "*_kernel_stats.csv"))
        if not files:
            continue
        rows = list(csv.DictReader(open(max(files, key=os.path.getmtime))))
        with open(os.path.join(out, dst), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow([f"# rocprofv3 --kernel-trace --stats -- {cmd}  (MI355X, gfx950)"])
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                w.writerow([short(r["Name"])[:140], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])

    # 1. calibration: kernels of known byte count
    cal = {"unit": "bytes per launch", "known_bytes_per_stream": KNOWN,
           "rule": "FETCH_SIZE KiB x 2 x 1024 (gfx950 wide-read undercount), WRITE_SIZE KiB x 1024", "kernels": {}}
    kf = collections.defaultdict(list)
    kw = collections.defaultdict(list)
    for r in rows_of(os.path.join(src, "kb_fetch"), "*_counter_collection.csv"):
        kf[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for r in rows_of(os.path.join(src, "kb_write"), "*_counter_collection.csv"):
        kw[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for name in sorted(kf):
        if any(k in name for k in ("copy_kernel<1, true>", "copy_kernel<4, false>", "copy_kernel<4, true>", "read_kernel<4>", "write_kernel<4>", "read8_kernel",
                                   "write8_kernel", "shrink_kernel")):
            reads = 0 if "write" in name else KNOWN
            writes = 0 if "read" in name else (KNOWN // 2 if "shrink_kernel<2" in name else KNOWN // 4 if "shrink_kernel<4" in name else KNOWN)
            rd, wb = statistics.median(kf[name]) * 2048.0, statistics.median(kw.get(name, [0])) * 1024.0
            cal["kernels"][name] = {"known_read_bytes": reads, "known_write_bytes": writes, "read_bytes_counted_x2": rd, "write_bytes_counted": wb,
                                    "read_ratio": round(rd / reads, 4) if reads else None, "write_ratio": round(wb / writes, 4) if writes else None}
    json.dump(cal, open(os.path.join(out, f"{tag}_pmc_calibration.json"), "w"), indent=1)

    stamp = time.strftime("%Y-%m-%d")
    # 2. the step's kernels (bench.py --core-extras)
    step, _, _ = per_entry(src, "", "trace.log")
    json.dump({"command": cmd_step, "collected": stamp, "note": "one entry per role of the timed step; rocprof_avg_us = rocprofv3 --kernel-trace duration, "
               "hbm_bytes = FETCH_SIZE x 2 KiB + WRITE_SIZE KiB of the separate --pmc passes (medians over the entry's launches, warm-up launches dropped)",
               "entries": step}, open(os.path.join(out, f"{tag}_step_kernels.json"), "w"), indent=1)
    traffic = {k: v["hbm_bytes"] for k, v in step.items()}
    traffic["_source"] = (f"profiles/{tag}_step_kernels.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, FETCH_SIZE x2 per the gfx950 rule) of "
                          f"`{cmd_step}`, collected {stamp} on one MI355X; NOT measured in the run that prints this line")
    json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)

    # 3. one layer's launch kinds, export, W1/W2 one-launch, Asym (bench.py --model-shapes)
    ms, _, doc = per_entry(src, "ms_", "ms_trace.log")
    bench_us = {e["kernel"]: e for e in doc["entries"]}
    narrow = {"quantize_kv pair fwd under autocast: fp32 results": "loads 8 B per lane", "quantize_kv pair bwd under autocast: fp32 grads in, bf16 out": "stores 8 B per lane"}
    for name, v in ms.items():
        b = bench_us.get(name)
        if b:
            v["bench_hip_event_us_under_profiler"] = b["us_per_launch"]
            v["bytes_moved_by_design"] = b["bytes_moved_per_launch"]
            v["algorithmic_bytes"] = b["algorithmic_bytes_per_launch"]
            v["hbm_over_moved"] = round(v["hbm_bytes"] / b["bytes_moved_per_launch"], 4)
            v["achieved_GBs_rocprof"] = round(b["bytes_moved_per_launch"] / (v["rocprof_avg_us"] * 1e-6) / 1e9, 1)
        if name in narrow:
            v["note"] = f"{narrow[name]}: see the 8-byte calibration kernels in {tag}_pmc_calibration.json"
    json.dump({"command": cmd_ms, "collected": stamp, "entries": ms}, open(os.path.join(out, f"{tag}_model_shapes_profile.json"), "w"), indent=1)
    tms = {k: v["hbm_bytes"] for k, v in ms.items()}
    tms["_source"] = (f"profiles/{tag}_model_shapes_profile.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, FETCH_SIZE x2) of `{cmd_ms}`, "
                      f"collected {stamp} on one MI355X; NOT measured in the run that prints this line")
    json.dump(tms, open(os.path.join(out, "traffic_model_shapes.json"), "w"), indent=1)

    for title, d in (("step kernels", step), ("model shapes", ms)):
        print(f"== {title}")
        for k, v in d.items():
            extra = f"  hbm/moved {v['hbm_over_moved']:.3f}  bench-under-profiler {v['bench_hip_event_us_under_profiler']:.2f} us" if "hbm_over_moved" in v else ""
            print(f"{k[:84]:86s} {v['rocprof_avg_us']:8.2f} us  hbm {v['hbm_bytes'] / 1e6:8.2f} MB{extra}")


if __name__ == "__main__":
    main()
