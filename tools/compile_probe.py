import sys, torch
sys.path.insert(0, "/root/repo")
from llm_qat_amd.utils_quant import QuantizeLinear
torch.manual_seed(0)
lin = QuantizeLinear(512, 256, w_bits=4, a_bits=8).cuda().bfloat16()
x = (torch.randn(8, 512, device="cuda") * 1.5).bfloat16().requires_grad_(True)
ref = lin(x); ref.float().square().mean().backward()
gw, gx = lin.weight.grad.clone(), x.grad.clone()
lin.zero_grad(set_to_none=True); x.grad = None
import torch._dynamo as dynamo
clin = torch.compile(lin, backend="aot_eager")
out = clin(x); out.float().square().mean().backward()
print("compile ok:", torch.equal(out, ref), torch.equal(lin.weight.grad, gw), torch.equal(x.grad, gx))
exp = dynamo.explain(lin)(x)
print("graphs", exp.graph_count, "breaks", exp.graph_break_count)
for r in exp.break_reasons[:3]: print(" -", str(r.reason)[:200])
