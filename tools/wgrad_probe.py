#!/usr/bin/env python3
"""What does the gradient that reaches a QuantizeLinear weight node look like on this device?  (view? contiguous? who else holds
it?)  Evidence for the run-time guard of the in-place weight gradient (utils_quant._inplace_ok).   python tools/wgrad_probe.py"""
import json
import sys

import torch

dev = "cuda" if torch.cuda.is_available() else "cpu"
out = []


class Probe(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, tag):
        ctx.tag = tag
        return w * 1.0

    @staticmethod
    def backward(ctx, g):
        b = g._base
        out.append({"case": ctx.tag, "dtype": str(g.dtype), "has_base": b is not None, "contiguous": g.is_contiguous(), "storage_offset": g.storage_offset(),
                    "storage_bytes": g.untyped_storage().nbytes(), "tensor_bytes": g.numel() * g.element_size(), "use_count": g._use_count(),
                    "refcount": sys.getrefcount(g), "base_shape": None if b is None else list(b.shape), "shape": list(g.shape), "stride": list(g.stride())})
        return g, None


for dt in (torch.bfloat16, torch.float32):
    for ac in (False, True):
        for xshape in ((64, 512), (2, 32, 512)):
            for mode in ("plain", "hook_stash", "two_consumers"):
                stash = []
                w = torch.randn(256, 512, device=dev, dtype=dt, requires_grad=True)
                x = torch.randn(*xshape, device=dev, dtype=dt, requires_grad=True)
                with torch.autocast(dev, dtype=torch.bfloat16, enabled=ac):
                    wq = Probe.apply(w, f"{dt} autocast={ac} x{list(xshape)} {mode}")
                    if mode == "hook_stash":
                        wq.register_hook(lambda g: stash.append(g) or None)
                    y = torch.nn.functional.linear(x, wq)
                    if mode == "two_consumers":
                        y = y + wq.float().sum().to(y.dtype)
                del wq
                y.float().sum().backward()
print(json.dumps(out, indent=1))
