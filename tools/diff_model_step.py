#!/usr/bin/env python3
"""Per-kernel difference between two rocprofv3 kernel-stats CSVs of the whole-model step (tools/profile_model_step.sh):
unquantized vs this package.  Shows where the quantized step's extra GPU time goes.

    python tools/diff_model_step.py <tag> [steps_traced=6]   -> profiles/<tag>_model_step_kernel_diff.txt
"""
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path):
    return {r["Name"]: (int(r["TotalDurationNs"]), int(r["Calls"])) for r in csv.DictReader(open(path))}


def short(n):
    n = re.sub(r"std::array<char\*, \d+ul>|at::native::|\(anonymous namespace\)::|c10::", "", n)
    return n[:130]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6   # 2 warm-up + 4 timed iterations of model_step_bench --iters 4
    a = load(os.path.join(ROOT, "gpurun_out", f"{tag}_modelstep_noquant", "ms_kernel_stats.csv"))
    b = load(os.path.join(ROOT, "gpurun_out", f"{tag}_modelstep_ours", "ms_kernel_stats.csv"))
    rows = []
    for n in set(a) | set(b):
        ta, ca = a.get(n, (0, 0))
        tb, cb = b.get(n, (0, 0))
        rows.append((tb - ta, n, ta, ca, tb, cb))
    rows.sort(key=lambda r: -abs(r[0]))
    fq = sum(tb for _, n, _, _, tb, _ in rows if "fq::" in n)
    tot_a, tot_b = sum(v[0] for v in a.values()), sum(v[0] for v in b.values())
    lines = [f"# whole-model step, 2 LLaMA-7B-sized layers, seq 2048, bf16 autocast, W4A8KV4; GPU kernel time per step (us), {steps} steps traced",
             f"# unquantized {tot_a / steps / 1e3:.1f} us   this package {tot_b / steps / 1e3:.1f} us   difference {(tot_b - tot_a) / steps / 1e3:.1f} us",
             f"# of which fq:: kernels {fq / steps / 1e3:.1f} us; the rest is ATen work on the fp32 K/V the reference's op returns under autocast (RoPE, casts)",
             "# delta_us  calls/step a->b   us/step a->b   kernel"]
    for d, n, ta, ca, tb, cb in rows[:40]:
        lines.append(f"{d / steps / 1e3:9.1f}  {ca / steps:5.1f}->{cb / steps:5.1f}  {ta / steps / 1e3:8.1f}->{tb / steps / 1e3:8.1f}  {short(n)}")
    out = os.path.join(ROOT, "profiles", f"{tag}_model_step_kernel_diff.txt")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:4]))


if __name__ == "__main__":
    main()
