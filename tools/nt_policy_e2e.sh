#!/bin/bash
# Whole-model step (tools/model_step_bench.py --autocast, this library only) under different non-temporal thresholds:
# does keeping fake-quant OUTPUTS cacheable (plain stores) help the GEMM that reads them next?
#   tools/nt_policy_e2e.sh  -> gpurun_out/nt_policy_e2e.log
out=gpurun_out/nt_policy_e2e.log
: > $out
for cfg in "4 72" "40 72" "72 72" "200 200" "100000 100000"; do
  set -- $cfg
  for rep in 1 2; do
    echo "== NT store >= $1 MiB, NT load >= $2 MiB (run $rep)" >> $out
    LLMQAT_FQ_NT_LOAD_MIN_MB=$2 python3 tools/model_step_bench.py --autocast --iters 10 --only "llm_qat_amd" 2>/dev/null | grep impl >> $out
  done
done
cat $out
