#!/usr/bin/env python3
"""Host time of the pieces of one QuantizeLinear(11008 -> 4096, W4 A8) step on the metric tensors, each timed alone (perf_counter over
many iterations, a device sync every 32 so that no launch ever waits for queue space): where the module path's host time goes.
Companion of tools/api_path_probe.py (totals in the un-synchronised loop) and tools/host_path_profile.py (cProfile ranking)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import llm_qat_amd  # noqa: E402
from llm_qat_amd import _lib, ops  # noqa: E402
import llm_qat_amd.utils_quant as UQ  # noqa: E402

dev = torch.device("cuda:0")
rows, cols = 4096, 11008
N = int(os.environ.get("N", "640"))


class _NoGemm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return torch.empty(x.shape[:-1] + (w.shape[0],), dtype=x.dtype, device=x.device)

    @staticmethod
    def backward(ctx, go):
        x, w = ctx.saved_tensors
        return torch.empty_like(x), torch.empty_like(w)


class _Floor(torch.nn.Module):
    def __init__(self, w):
        super().__init__()
        self.weight = w

    def forward(self, x):
        return F.linear(x, self.weight)


def timeit(fn, n=N, setup=None):
    tot = 0.0
    for i in range(n):
        if setup:
            setup()
        t0 = time.perf_counter()
        fn()
        tot += time.perf_counter() - t0
        if i % 32 == 31:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return tot / n * 1e6


lin = UQ.QuantizeLinear(cols, rows, w_bits=4, a_bits=8).to(device=dev, dtype=torch.bfloat16)
a = torch.randn(rows, cols, device=dev).bfloat16().requires_grad_(True)
go = torch.empty(rows, rows, dtype=torch.bfloat16, device=dev)
real = F.linear
F.linear = torch.nn.functional.linear = lambda x, w, b=None: _NoGemm.apply(x, w)
w = lin.weight
L = _lib.lib()
res = {}
state = {}


def fwd():
    state["out"] = lin(a)


def bwd():
    state["out"].backward(go)


def clear():
    lin.weight.grad = a.grad = None


for _ in range(8):
    clear(), fwd(), bwd()
res["module forward (lin(a))"] = timeit(fwd, setup=clear)
res["module backward (out.backward(go))"] = timeit(bwd, setup=lambda: (clear(), fwd()))
floor = _Floor(w)
res["floor forward (plain module + no-launch Function)"] = timeit(lambda: state.__setitem__("out", floor(a)), setup=clear)
res["floor backward"] = timeit(bwd, setup=lambda: (clear(), state.__setitem__("out", floor(a))))
res["ops.pair_forward alone"] = timeit(lambda: state.__setitem__("res", ops.pair_forward(w, a, 4, 8, -2.0, 2.0, True, True)))
r = state["res"]
mw, mx = ops._mask_bytes(rows, cols, _lib.DTYPE_BF16), ops._mask_bytes(rows, cols, _lib.DTYPE_BF16)
wq, xq, sw, sx = r[0], r[1], r[2], r[3]
st = torch.cuda.current_stream().cuda_stream
args = (w.data_ptr(), wq.data_ptr(), rows, 4, sw.data_ptr(), sw.data_ptr() + rows * 8, mw, a.data_ptr(), xq.data_ptr(), rows, 8, sx.data_ptr(), sx.data_ptr() + rows * 8, mx,
        cols, _lib.DTYPE_BF16, 1, 0, -2.0, 2.0, st)
fn = L.fq_sym_fwd_pair
res["raw ctypes fq_sym_fwd_pair (21 prebuilt args)"] = timeit(lambda: fn(*args))
res["4 allocations (2 empty_like 90 MB + 2 side buffers)"] = timeit(lambda: (torch.empty_like(w), torch.empty_like(a), torch.empty(rows * 8 + mw, dtype=torch.uint8, device=dev),
                                                                             torch.empty(rows * 8 + mx, dtype=torch.uint8, device=dev)))
res["8 data_ptr() calls"] = timeit(lambda: (w.data_ptr(), a.data_ptr(), wq.data_ptr(), xq.data_ptr(), sw.data_ptr(), sx.data_ptr(), w.data_ptr(), a.data_ptr()))
res["_PairNode.apply on a prebuilt result"] = timeit(lambda: UQ._PairNode.apply(w, a, r, _lib.DTYPE_BF16, False))
res["_NoGemm.apply (the stand-in's own Function)"] = timeit(lambda: _NoGemm.apply(a, w))
res["nn.Module.__call__ of an empty module"] = timeit(lambda m=torch.nn.Identity(): m(a))
res["_state_word + _region + _stream + key tuple"] = timeit(lambda: (UQ._state_word(a), UQ._region(), ops._stream(a), (UQ._SymQuantizerOperand, 8, False, 3)))
gw, gx = torch.empty_like(w), torch.empty_like(a)
res["ops.pair_backward alone (weight in place)"] = timeit(lambda: ops.pair_backward(gw, gx, sw, sx, rows, rows, cols, -2.0, 2.0, inplace_w=True))
res["_inplace_ok"] = timeit(lambda: UQ._inplace_ok(gw))
bargs = (gw.data_ptr(), gw.data_ptr(), rows, sw.data_ptr(), sw.data_ptr() + rows * 8, gx.data_ptr(), xq.data_ptr(), rows, sx.data_ptr(), sx.data_ptr() + rows * 8, cols, -2.0, 2.0,
         _lib.DTYPE_BF16, st)
fb = L.fq_ste_bwd_mask_pair
res["raw ctypes fq_ste_bwd_mask_pair (15 prebuilt args)"] = timeit(lambda: fb(*bargs))
for k, v in res.items():
    print(f"{k:62s} {v:7.2f} us")
F.linear = torch.nn.functional.linear = real

# ---- inside the backward: how much of out.backward(go) is spent inside this library's node, how much around it (engine, GIL hand-over,
# the stand-in's own node, AccumulateGrad)?  The node's backward runs on the engine's device thread: time it from inside.
F.linear = torch.nn.functional.linear = lambda x, w, b=None: _NoGemm.apply(x, w)
acc = {"node": 0.0, "pair_backward": 0.0, "nogemm": 0.0, "n": 0}
orig_node, orig_pb, orig_ng = UQ._PairNode.backward, ops.pair_backward, _NoGemm.backward


def timed(name, f):
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        acc[name] += time.perf_counter() - t0
        return r
    return g


UQ._PairNode.backward = staticmethod(timed("node", orig_node))
ops.pair_backward = timed("pair_backward", orig_pb)
_NoGemm.backward = staticmethod(timed("nogemm", orig_ng))
tot = 0.0
M = 320
for i in range(M):
    clear(), fwd()
    t0 = time.perf_counter()
    bwd()
    tot += time.perf_counter() - t0
    if i % 32 == 31:
        torch.cuda.synchronize()
print(f"out.backward(go) {tot / M * 1e6:.2f} us, of which inside _PairNode.backward {acc['node'] / M * 1e6:.2f} us (ops.pair_backward {acc['pair_backward'] / M * 1e6:.2f} us), "
      f"inside the stand-in's backward {acc['nogemm'] / M * 1e6:.2f} us")
F.linear = torch.nn.functional.linear = real
