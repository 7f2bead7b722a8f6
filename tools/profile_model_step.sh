#!/bin/bash
# rocprofv3 kernel trace of one whole-model step, unquantized vs this package (bf16 autocast): where the extra time goes.
#   tools/profile_model_step.sh <tag>   -> gpurun_out/<tag>_modelstep_{noquant,ours}/ (kernel stats CSVs)
set -e
tag=${1:-r01}
export TMPDIR=/tmp
out=$PWD/gpurun_out
for which in noquant ours; do
  label="no quantization (bf16 linears)"; [ $which = ours ] && label="llm_qat_amd"
  rocprofv3 --kernel-trace --stats -d $out/${tag}_modelstep_$which -o ms --output-format csv -- python3 tools/model_step_bench.py --autocast --iters 4 --only "$label" > $out/${tag}_modelstep_$which.log 2>&1
done
ls $out/${tag}_modelstep_ours
