#!/usr/bin/env python3
"""Golden vectors for float64 tensors (the reference has no dtype restriction: models/utils_quant.py:37-74, :96-149, :77-87,
:202-242 run on whatever dtype they are given).  Runs ONLY in the build container: imports the real reference on CPU and records
inputs + the outputs the reference itself produced; fixtures are data only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_f64.py   ->  tests/golden/f64.npz
"""
import json
import os
import sys

sys.dont_write_bytecode = True
REF = os.environ.get("LLMQAT_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from models.utils_quant import AsymQuantizer, QuantizeLinear, SymQuantizer  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import asym_parts, idx_to_i32, sym_parts  # noqa: E402  (the same restated op sequences, dtype-agnostic)

arrays, cases = {}, []


def add(name, meta, **arrs):
    cases.append(dict(name=name, **meta))
    for k, v in arrs.items():
        arrays[f"{name}/{k}"] = np.ascontiguousarray(v)


def rows_mixed(gen, shape):
    x = torch.randn(shape, generator=gen, dtype=torch.float64)
    r = x.reshape(-1, shape[-1])
    scales = [1e-9, 1e-5, 0.02, 1.0, 3.0, 1e6]
    for i in range(r.shape[0]):
        r[i] *= scales[i % len(scales)]
    return x


def adversarial(cols=24):
    rows = [torch.zeros(cols), torch.full((cols,), 1e-7), torch.linspace(-2.0, 2.0, cols), torch.linspace(-1, 1, cols) * 1e300,
            torch.linspace(-1, 1, cols) * 1e-300, torch.arange(cols) + 0.5, -(torch.arange(cols) + 0.5), torch.full((cols,), -3.25)]
    x = torch.stack([r.double() for r in rows])
    extra = torch.randn(3, cols, dtype=torch.float64, generator=torch.Generator().manual_seed(5))
    extra[0, 3] = float("nan")
    extra[1, 5] = float("inf")
    extra[2, 7] = float("-inf")
    return torch.cat([x, extra])


def main():
    gen = torch.Generator().manual_seed(64)
    clip = torch.tensor([-2.0, 2.0])
    inputs = [("mixed_5x33", rows_mixed(gen, (5, 33)), False), ("mixed_3x520", rows_mixed(gen, (3, 520)), False), ("adversarial", adversarial(), False),
              ("rank3_2x3x40", rows_mixed(gen, (2, 3, 40)), False), ("rank4_2x2x3x8", rows_mixed(gen, (2, 2, 3, 8)), False), ("layerwise_4x50", rows_mixed(gen, (4, 50)), True)]
    for tag, x, lw in inputs:
        for bits in (4, 8, 16):
            idx, s, y_parts = sym_parts(x, bits, lw)
            y = SymQuantizer.apply(x, clip, bits, lw)
            assert y.dtype == torch.float64 and np.array_equal(y.numpy().view(np.uint64), y_parts.numpy().view(np.uint64))
            add(f"sym_{tag}_b{bits}", dict(kind="sym", bits=bits, layerwise=lw, shape=list(x.shape)), x=x.numpy(), y=y.numpy(), idx=idx_to_i32(idx))
            idx, alpha, beta, y_parts = asym_parts(x, bits, lw)
            y = AsymQuantizer.apply(x, clip, bits, lw)
            assert np.array_equal(y.numpy().view(np.uint64), y_parts.numpy().view(np.uint64))
            add(f"asym_{tag}_b{bits}", dict(kind="asym", bits=bits, layerwise=lw, shape=list(x.shape)), x=x.numpy(), y=y.numpy(), idx=idx_to_i32(idx))
    # STE backward, standard and custom clips (the clip is a float32 tensor, as the reference builds it :198,:245)
    for tag, x, _ in inputs[:3]:
        for lo, hi in ((-2.0, 2.0), (-0.5, 0.75), (-0.3009, 0.3009)):
            xr = x.clone().requires_grad_(True)
            g = torch.randn(x.shape, generator=gen, dtype=torch.float64)
            SymQuantizer.apply(xr, torch.tensor([lo, hi]), 8, False).backward(g)
            add(f"ste_{tag}_{lo}_{hi}", dict(kind="ste", lo=lo, hi=hi, shape=list(x.shape)), x=x.numpy(), g=g.numpy(), gx=xr.grad.numpy())
    # QuantizeLinear in float64: 1-/2-bit weight branches (value of the detach trick) and the W4A8 module output
    for w_bits in (1, 2):
        for lw in (False, True):
            lin = QuantizeLinear(40, 6, w_bits=w_bits, a_bits=32, weight_layerwise=lw).double()
            with torch.no_grad():
                lin.weight.copy_(rows_mixed(gen, (6, 40)))
                lin.weight[1, 3] = 0.0
            w = lin.weight.detach().clone()
            am = torch.mean(abs(w)) if lw else torch.mean(abs(w), dim=1, keepdim=True)
            sc = am if w_bits == 1 else 2 * am
            out = lin(torch.eye(40, dtype=torch.float64))          # F.linear(I, Wq) = Wq^T exactly (products with 0 / 1 only)
            add(f"w12_b{w_bits}_lw{int(lw)}", dict(kind="w12", w_bits=w_bits, layerwise=lw, shape=[6, 40]), w=w.numpy(), scale=sc.reshape(-1).numpy(), wq=out.t().detach().numpy())
    lin = QuantizeLinear(64, 16, w_bits=4, a_bits=8).double()
    x = rows_mixed(gen, (3, 5, 64)).requires_grad_(True)
    out = lin(x)
    out.square().sum().backward()
    add("qlinear_w4a8", dict(kind="qlinear", w_bits=4, a_bits=8), w=lin.weight.detach().numpy(), x=x.detach().numpy(), out=out.detach().numpy(),
        gw=lin.weight.grad.numpy(), gx=x.grad.numpy())
    arrays["manifest"] = np.frombuffer(json.dumps({"meta": {"torch": torch.__version__, "dtype": "float64", "reference": "models/utils_quant.py"},
                                                   "cases": cases}).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "f64.npz"), **arrays)
    print(f"{len(cases)} cases -> tests/golden/f64.npz ({os.path.getsize(os.path.join(HERE, 'f64.npz')) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
