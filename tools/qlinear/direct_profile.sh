#!/bin/bash
# rocprofv3 passes over tools/qlinear/direct_profile.py (run on the GPU box):  kernel trace + L2 hit/miss + FETCH_SIZE
#   -> gpurun_out/direct_prof/{trace,tcc,fetch,sq}
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/direct_prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="python3 $ROOT/tools/qlinear/direct_profile.py"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $P > "$OUT/trace.log" 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d "$OUT/tcc" -- $P > "$OUT/tcc.log" 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- $P > "$OUT/fetch.log" 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d "$OUT/sq" -- $P > "$OUT/sq.log" 2>&1 || exit 1
find "$OUT" -name "*.db" -delete 2>/dev/null
echo "direct profile done"; du -sh "$OUT"
