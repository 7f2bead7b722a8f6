#!/usr/bin/env python3
"""The strided-view cases of tests/test_gpu_strided.py that the kernels serve, run once each (forward + backward, Sym and Asym, with and
without autocast) for `rocprofv3 --kernel-trace --stats`: the trace must hold fq:: kernels only -- no ATen copy / contiguous kernel.
    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/strided_trace -- python3 $ROOT/tools/strided_trace.py
The inputs are built BEFORE the marker kernel (a torch.zeros(7).cumsum launch): everything after it in the trace belongs to the quantizers."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import llm_qat_amd  # noqa: E402
from llm_qat_amd import ops  # noqa: E402
from llm_qat_amd.utils_quant import AsymQuantizer, SymQuantizer  # noqa: E402
from test_gpu_strided import VIEWS  # noqa: E402

clip = torch.tensor([-2.0, 2.0])
cases = []
for name, (build, served) in VIEWS.items():
    if not served or name in ("misaligned_slice_2d", "odd_width_slice_2d"):
        continue
    for ac in (False, True):
        for Q in (SymQuantizer, AsymQuantizer):
            x = build(torch.bfloat16).detach().requires_grad_(True)
            g = torch.empty_like(x, dtype=torch.float32 if (ac and Q is SymQuantizer) else x.dtype).normal_()
            cases.append((name, ac, Q, x, g))
torch.cuda.synchronize()
torch.zeros(7, device="cuda").cumsum(0)      # marker
before = ops._views_served
for name, ac, Q, x, g in cases:
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=ac):
        y = Q.apply(x, clip, 8, False)
    y.backward(g)
torch.cuda.synchronize()
print(f"{len(cases)} strided forward + backward pairs, {ops._views_served - before} launches took a view")
