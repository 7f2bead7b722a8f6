#!/usr/bin/env python3
"""-> how busy the GPU was over the LAST `n` steps of a tools/api_path_trace.py run, from rocprofv3's kernel trace:
sum of the fq:: kernels' durations / (end of the last - start of the first of them).   python tools/api_path_trace_summary.py <dir> [n]"""
import csv
import glob
import sys


def main(d, n=300):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    assert f, "no kernel trace under " + d
    rows = [r for r in csv.DictReader(open(f[0])) if "fq::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-2 * n:]                  # two launches per step: the pair forward, the pair backward
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
    span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
    names = sorted({r["Kernel_Name"].split("(")[0][:60] for r in rows})
    print(f"{len(rows)} fq:: dispatches over {span / 1e3 / n:.1f} us/step; kernels busy {busy / 1e3 / n:.1f} us/step = {busy / span:.3f} of the span; kernels: {names}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 300)
