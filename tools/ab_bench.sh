#!/bin/bash
# A/B two builds of the library on the SAME box (box-to-box variance is 3-5 %): alternate them, ROUNDS times each, through
# `bench.py --model-shapes`, then print per-entry medians.   tools/ab_bench.sh <libA.so> <libB.so> [rounds] [outdir]
# The loader is pointed at each variant through LLMQAT_AMD_LIB: the product library is never overwritten (round 3's
# version copied the variants over it and restored it only if every run succeeded).
set -eu
A=$(readlink -f "$1"); B=$(readlink -f "$2"); ROUNDS=${3:-2}; OUT=${4:-gpurun_out/ab}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$OUT"
for r in $(seq 1 "$ROUNDS"); do
  for v in A B; do
    src=$A; [ $v = B ] && src=$B
    LLMQAT_AMD_LIB="$src" timeout -k 10 400 python3 "$ROOT/bench.py" --model-shapes --steps 40 > "$OUT/$v$r.json" 2> "$OUT/$v$r.err"
  done
done
python3 - "$OUT" "$ROUNDS" <<'PY'
import json, sys, statistics
out, rounds = sys.argv[1], int(sys.argv[2])
def load(v):
    runs = [json.loads(open(f"{out}/{v}{r}.json").read().strip().splitlines()[-1]) for r in range(1, rounds + 1)]
    names = runs[0]["order"]
    return names, {n: statistics.median(run["entries"][i]["us_per_launch"] for run in runs) for i, n in enumerate(names)}
names, a = load("A"); _, b = load("B")
res = {"A_us": a, "B_us": b}
json.dump(res, open(f"{out}/ab_summary.json", "w"), indent=1)
for n in names:
    print(f"{n[:78]:80s} A {a[n]:8.2f}  B {b[n]:8.2f}  B/A {b[n]/a[n]:.3f}")
PY
