#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for HBM traffic.
#   tools/profile_bench.sh <tag>      -> gpurun_out/prof_<tag>/{trace,fetch,write,ms_trace,ms_fetch,ms_write,kb_fetch,kb_write}
# Counter passes are separate runs with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass).  The program
# follows `--` directly (python3 / the kbench binary): no env / bash -c hop behind the profiler.
#   step kernels   : bench.py --core-extras        (one launch shape per role of the timed step)
#   layer kernels  : bench.py --model-shapes       (every launch kind of a layer, export, W1/W2 one-launch, Asym: fixed order and
#                    call count, so the trace splits per entry by counting fq:: dispatches -- tools/summarize_profile.py)
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# (--sustain-seconds 0: the 5 s steady-state leg would add ~86 000 dispatches of the same two kernels to every trace)
BENCH="python3 $ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --core-extras --no-sidecar --sustain-seconds 0"
MS="python3 $ROOT/bench.py --model-shapes --steps 30"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || exit 1
echo "trace done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- $BENCH > "$OUT/fetch.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- $BENCH > "$OUT/write.log" 2>&1 || exit 1
echo "step pmc done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ms_trace" -- $MS > "$OUT/ms_trace.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/ms_fetch" -- $MS > "$OUT/ms_fetch.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/ms_write" -- $MS > "$OUT/ms_write.log" 2>&1 || exit 1
echo "model-shapes done"
# calibration on kernels with a known byte count (plain copy / read-only / write-only, 16 and 8 bytes per lane, in tools/kbench)
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/kb_fetch" -- $ROOT/tools/kbench 4096 11008 ceilings > "$OUT/kb_fetch.log" 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/kb_write" -- $ROOT/tools/kbench 4096 11008 ceilings > "$OUT/kb_write.log" 2>&1 || exit 1
# keep what travels back small: the per-dispatch csv files are what the summarizer reads
find "$OUT" -name "*.db" -delete 2>/dev/null
echo "profile $TAG done"; du -sh "$OUT"
