#!/usr/bin/env python3
"""Host-side cost per quantizer call (Python + ctypes + allocator + autograd), measured on a tensor small enough
that the GPU is never the bottleneck.   python tools/host_overhead.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import llm_qat_amd
from llm_qat_amd.utils_quant import SymQuantizer
from oracle.eager_chain import EagerSym

clip = torch.tensor([-2.0, 2.0])
x = torch.randn(64, 256, device="cuda", dtype=torch.bfloat16)
xg = x.clone().requires_grad_(True)
g = torch.ones_like(x)
N = 2000
for name, fn in (("llm_qat_amd fwd (no grad)", lambda: SymQuantizer.apply(x, clip, 8, False)),
                 ("llm_qat_amd fwd (grad)", lambda: SymQuantizer.apply(xg, clip, 8, False)),
                 ("llm_qat_amd fwd+bwd", lambda: SymQuantizer.apply(xg, clip, 8, False).backward(g)),
                 ("eager chain fwd (grad)", lambda: EagerSym.apply(xg, clip, 8, False)),
                 ("eager chain fwd+bwd", lambda: EagerSym.apply(xg, clip, 8, False).backward(g))):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{name:28s} {1e6 * (t1 - t0) / N:7.1f} us/call (host)")
