#!/usr/bin/env python3
"""torch.compile over a QuantizeLinear: graph count / breaks, fullgraph compile, result equality with eager."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch._dynamo as dynamo  # noqa: E402

from llm_qat_amd.utils_quant import QuantizeLinear  # noqa: E402

torch.manual_seed(0)
lin = QuantizeLinear(512, 256, w_bits=4, a_bits=8).cuda().bfloat16()
x = (torch.randn(8, 512, device="cuda") * 1.5).bfloat16().requires_grad_(True)
ref = lin(x)
ref.float().square().mean().backward()
gw, gx = lin.weight.grad.clone(), x.grad.clone()
lin.zero_grad(set_to_none=True)
x.grad = None
exp = dynamo.explain(lin)(x)
print("graphs", exp.graph_count, "breaks", exp.graph_break_count)
for r in exp.break_reasons[:5]:
    print(" -", str(r.reason)[:300])
for gm in exp.graphs:
    print(gm.code)
dynamo.reset()
for backend in ("aot_eager", "inductor"):
    lin.zero_grad(set_to_none=True)
    x.grad = None
    clin = torch.compile(lin, backend=backend, fullgraph=True)
    out = clin(x)
    out.float().square().mean().backward()
    print(backend, "fullgraph ok; equal to eager:", torch.equal(out, ref), torch.equal(lin.weight.grad, gw), torch.equal(x.grad, gx))
    dynamo.reset()
