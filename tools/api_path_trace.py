#!/usr/bin/env python3
"""The module path under the profiler: N steps of QuantizeLinear(11008 -> 4096, W4 A8) forward + backward on the metric tensors (the GEMM
replaced by the no-launch stand-in, as in bench.py's api_path), nothing else -- so that a `rocprofv3 --kernel-trace` of this process can
say how busy the GPU was between the first and the last fake-quant kernel (tools/api_path_trace_summary.py):

    rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/api_path_trace.py            # C++ autograd node (default)
    API_TRACE_NODE=python rocprofv3 ... -- python3 tools/api_path_trace.py                          # Python node (cpp_node(False))
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import llm_qat_amd  # noqa: E402
import llm_qat_amd.utils_quant as UQ  # noqa: E402
from llm_qat_amd.utils_quant import QuantizeLinear  # noqa: E402

N = int(os.environ.get("N", "300"))
dev = torch.device("cuda:0")
rows, cols = 4096, 11008
lins = []
for k in range(4):
    lin = QuantizeLinear(cols, rows, w_bits=4, a_bits=8).to(device=dev, dtype=torch.bfloat16)
    with torch.no_grad():
        lin.weight.normal_(0, 0.02)
    lins.append((lin, torch.randn(rows, cols, device=dev).bfloat16().requires_grad_(True)))
go = torch.empty(rows, rows, dtype=torch.bfloat16, device=dev)
assert UQ._cnode is not None, "the stand-in GEMM lives in _fq_node.so"
no_gemm = UQ._cnode.no_gemm_linear
if os.environ.get("API_TRACE_NODE") == "python":
    llm_qat_amd.cpp_node(False)
F.linear = torch.nn.functional.linear = lambda x, w, b=None: no_gemm(x, w)


def step(k):
    m, a = lins[k % 4]
    m.weight.grad = a.grad = None
    m(a).backward(go)


for k in range(20):
    step(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(N):
    step(k)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print(f"node {llm_qat_amd.host_node()}: {dt * 1e6:.1f} us/step wall over {N} steps (after 20 warm-up steps)")
