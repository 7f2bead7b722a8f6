#!/bin/bash
# A/B two builds of the library on the SAME box (box-to-box variance is 3-5 %): alternate them, ROUNDS times each, through
# `bench.py --model-shapes`, then print per-entry medians.   tools/ab_bench.sh <libA.so> <libB.so> [rounds] [outdir]
set -eu
A=$1; B=$2; ROUNDS=${3:-2}; OUT=${4:-gpurun_out/ab}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
LIB=$ROOT/llm-qat_amd/libllmqat_fakequant.so
mkdir -p "$OUT"; cp "$LIB" "$OUT/product.so.keep"; cp "$A" "$OUT/A.so"; cp "$B" "$OUT/B.so"; A=$OUT/A.so; B=$OUT/B.so
for r in $(seq 1 "$ROUNDS"); do
  for v in A B; do
    src=$A; [ $v = B ] && src=$B
    cp "$src" "$LIB"
    timeout -k 10 240 python3 "$ROOT/bench.py" --model-shapes --steps 40 > "$OUT/$v$r.json" 2> "$OUT/$v$r.err"
  done
done
cp "$OUT/product.so.keep" "$LIB"; rm -f "$OUT/A.so" "$OUT/B.so" "$OUT/product.so.keep"
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
