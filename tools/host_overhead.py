#!/usr/bin/env python3
"""Host-side cost per quantizer call (Python + ctypes + allocator + autograd), measured on a tensor small enough
that the GPU is never the bottleneck.   python tools/host_overhead.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import llm_qat_amd
from llm_qat_amd.utils_quant import SymQuantizer
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tiny_llama import EagerQuant   # test infrastructure: the reference's eager op chain as autograd Functions
EagerSym = EagerQuant().SymQuantizer

clip = torch.tensor([-2.0, 2.0])
x = torch.randn(64, 256, device="cuda", dtype=torch.bfloat16)
xg = x.clone().requires_grad_(True)
g = torch.ones_like(x)
N = 2000
from llm_qat_amd import ops
res = ops.train_forward("sym", x, 8, False, -2.0, 2.0)
_, side, rows, cols = res
outs = [SymQuantizer.apply(xg, clip, 8, False) for _ in range(64)]


def many_backward():
    torch.autograd.backward(outs, [g] * 64, retain_graph=True)

for name, fn in (("llm_qat_amd fwd (no grad)", lambda: SymQuantizer.apply(x, clip, 8, False)),
                 ("llm_qat_amd fwd (grad)", lambda: SymQuantizer.apply(xg, clip, 8, False)),
                 ("llm_qat_amd fwd+bwd", lambda: SymQuantizer.apply(xg, clip, 8, False).backward(g)),
                 ("ops.train_forward only", lambda: ops.train_forward("sym", x, 8, False, -2.0, 2.0)),
                 ("ops.train_backward only", lambda: ops.train_backward(g, side, rows, cols, -2.0, 2.0)),
                 ("autograd.backward over 64 nodes (/64)", many_backward),
                 ("eager chain fwd (grad)", lambda: EagerSym.apply(xg, clip, 8, False)),
                 ("eager chain fwd+bwd", lambda: EagerSym.apply(xg, clip, 8, False).backward(g))):
    best = float("inf")
    for rep in range(3):   # best of three: the first loop of a process runs ~2x slow (measured), whatever it times
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        best = min(best, t1 - t0)
    per = 1e6 * best / N / (64 if "64 nodes" in name else 1)
    print(f"{name:40s} {per:7.1f} us/call (host)")
