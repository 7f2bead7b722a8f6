#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for HBM traffic.
#   tools/profile_bench.sh <tag>      -> gpurun_out/prof_<tag>/{trace,fetch,write,kb_fetch,kb_write}
# Counter passes are separate runs with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --core-extras"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- $BENCH > "$OUT/fetch.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- $BENCH > "$OUT/write.log" 2>&1 || exit 1
# calibration on kernels with a known byte count (plain copy / read-only / write-only in tools/kbench)
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/kb_fetch" -- $ROOT/tools/kbench > "$OUT/kb_fetch.log" 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/kb_write" -- $ROOT/tools/kbench > "$OUT/kb_write.log" 2>&1 || exit 1
echo "profile $TAG done"; du -sh "$OUT"
