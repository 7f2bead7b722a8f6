#!/usr/bin/env python3
"""Where the module path's time goes on the metric tensors: host time of forward / backward separately, GPU time by events,
and the kernels one step launches (run under `rocprofv3 --kernel-trace --stats` for the list)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import llm_qat_amd  # noqa: E402
from llm_qat_amd.utils_quant import QuantizeLinear  # noqa: E402

dev = torch.device("cuda:0")
rows, cols = 4096, 11008
nsets = 4


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
    with torch.no_grad():
        lin.weight.normal_(0, 0.02)
    a = torch.randn(rows, cols, device=dev).bfloat16().requires_grad_(True)
    lins.append((lin, a))
go = torch.empty(rows, rows, dtype=torch.bfloat16, device=dev)
real = F.linear
import llm_qat_amd.utils_quant as _UQ
no_gemm = _UQ._cnode.no_gemm_linear if _UQ._cnode is not None else _NoGemm.apply   # (the C++ stand-in: its gradients arrive without a Python wrapper, as a GEMM's do)
F.linear = torch.nn.functional.linear = lambda x, w, b=None: no_gemm(x, w)
N = int(os.environ.get("N", "40"))


class _Floor(torch.nn.Module):   # what PyTorch itself costs for a module of this shape: one autograd Function, two leaves
    def __init__(self, lin):
        super().__init__()
        self.weight = lin.weight

    def forward(self, x):
        return F.linear(x, self.weight)


def loop(mods, n):
    tf = tb = 0.0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_all = time.perf_counter()
    e0.record()
    for k in range(n):
        m, a = mods[k % nsets]
        m.weight.grad = a.grad = None
        t0 = time.perf_counter()
        out = m(a)
        t1 = time.perf_counter()
        out.backward(go)
        t2 = time.perf_counter()
        tf += t1 - t0
        tb += t2 - t1
    e1.record()
    torch.cuda.synchronize()
    return tf / n * 1e6, tb / n * 1e6, (time.perf_counter() - t_all) / n * 1e6, e0.elapsed_time(e1) / n * 1e3


floors = [(_Floor(l), a) for l, a in lins]
loop(floors, 8)
print("floor (module + one no-launch Function + engine): host fwd %.1f us, host bwd %.1f us, wall %.1f us/step, GPU %.1f us/step" % loop(floors, N))
print("autograd node of the operand pair:", llm_qat_amd.host_node())
for rnd in range(int(os.environ.get("ROUNDS", "2"))):     # interleaved: the host's speed drifts over a run
    for name, setup in (("default, C++ node", lambda: (llm_qat_amd.conservative(False), llm_qat_amd.cpp_node(True))),
                        ("default, Python node", lambda: (llm_qat_amd.conservative(False), llm_qat_amd.cpp_node(False))),
                        ("conservative", lambda: llm_qat_amd.conservative(True))):
        setup()
        loop(lins, 8)
        print("%-22s host fwd %.1f us, host bwd %.1f us, wall %.1f us/step, GPU %.1f us/step" % ((name,) + loop(lins, N)))
llm_qat_amd.conservative(False)
llm_qat_amd.cpp_node(True)
tf = tb = 0.0
for k in range(8):
    lin, a = lins[k % nsets]
    lin.weight.grad = a.grad = None
    lin(a).backward(go)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t_all = time.perf_counter()
e0.record()
for k in range(N):
    lin, a = lins[k % nsets]
    lin.weight.grad = a.grad = None
    t0 = time.perf_counter()
    out = lin(a)
    t1 = time.perf_counter()
    out.backward(go)
    t2 = time.perf_counter()
    tf += t1 - t0
    tb += t2 - t1
e1.record()
torch.cuda.synchronize()
t_all = time.perf_counter() - t_all
print(f"host forward {tf / N * 1e6:.1f} us, host backward {tb / N * 1e6:.1f} us, wall {t_all / N * 1e6:.1f} us/step, GPU events {e0.elapsed_time(e1) / N * 1e3:.1f} us/step")
print(llm_qat_amd.stats())
# weight.grad: is it the GEMM's own output (handed on by reference) or a clone?
lin, a = lins[0]
print("weight.grad data_ptr unique per step (no clone expected):", lin.weight.grad.data_ptr())
F.linear = torch.nn.functional.linear = real
