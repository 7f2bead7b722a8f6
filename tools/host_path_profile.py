#!/usr/bin/env python3
"""cProfile of the module path on the metric tensors (QuantizeLinear(11008 -> 4096, W4 A8), F.linear replaced by a no-launch stand-in):
which Python functions the host time of a forward / backward goes to.  Companion of tools/api_path_probe.py (which gives the totals)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import llm_qat_amd  # noqa: E402
from llm_qat_amd.utils_quant import QuantizeLinear  # noqa: E402

dev = torch.device("cuda:0")
rows, cols, nsets = 4096, 11008, 4


class _NoGemm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return torch.empty(x.shape[:-1] + (w.shape[0],), dtype=x.dtype, device=x.device)

    @staticmethod
    def backward(ctx, go):
        x, w = ctx.saved_tensors
        return torch.empty_like(x), torch.empty_like(w)


lins = []
for k in range(nsets):
    lin = QuantizeLinear(cols, rows, w_bits=4, a_bits=8).to(device=dev, dtype=torch.bfloat16)
    a = torch.randn(rows, cols, device=dev).bfloat16().requires_grad_(True)
    lins.append((lin, a))
go = torch.empty(rows, rows, dtype=torch.bfloat16, device=dev)
F.linear = torch.nn.functional.linear = lambda x, w, b=None: _NoGemm.apply(x, w)
N = int(os.environ.get("N", "300"))


def steps(n, fwd_prof=None, bwd_prof=None):
    for k in range(n):
        m, a = lins[k % nsets]
        m.weight.grad = a.grad = None
        if fwd_prof:
            fwd_prof.enable()
        out = m(a)
        if fwd_prof:
            fwd_prof.disable()
        if bwd_prof:
            bwd_prof.enable()
        out.backward(go)
        if bwd_prof:
            bwd_prof.disable()
        if k % 8 == 7:
            torch.cuda.synchronize()   # keep the launch queue short: the profile is of host work, not of back-pressure


steps(16)
pf, pb = cProfile.Profile(), cProfile.Profile()
steps(N, pf, pb)
torch.cuda.synchronize()
for name, p in (("FORWARD", pf), ("BACKWARD", pb)):
    print(f"==== {name}: cProfile over {N} steps (times include the profiler's own overhead; read the ranking)")
    st = pstats.Stats(p)
    st.sort_stats("tottime").print_stats(22)
print(llm_qat_amd.stats())
