#!/bin/bash
# Utilisation counters for the bench kernels (separate --pmc passes, kernel-trace only).
#   tools/profile_pmc_util.sh <tag>  -> gpurun_out/pmc_<tag>/{sq,tcc,lds}
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d "$OUT/sq" -- $BENCH > "$OUT/sq.log" 2>&1 || echo "sq pass failed"
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d "$OUT/tcc" -- $BENCH > "$OUT/tcc.log" 2>&1 || echo "tcc pass failed"
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d "$OUT/lds" -- $BENCH > "$OUT/lds.log" 2>&1 || echo "lds pass failed"
echo "pmc util $TAG done"; du -sh "$OUT"
