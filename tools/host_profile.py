import sys, cProfile, pstats, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_qat_amd.utils_quant import SymQuantizer
from llm_qat_amd import ops
clip = torch.tensor([-2.0, 2.0])
x = torch.randn(64, 256, device="cuda", dtype=torch.bfloat16)
for _ in range(200): SymQuantizer.apply(x, clip, 8, False)
def t(fn, n=3000):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
print("apply no-grad", t(lambda: SymQuantizer.apply(x, clip, 8, False)))
print("ops.sym_quantize", t(lambda: ops.sym_quantize(x, 8, False)))
print("ops._rowwise", t(lambda: ops._rowwise("sym", x, 8, False, False, False)))
print("empty_like", t(lambda: torch.empty_like(x)))
with torch.no_grad():
    print("apply under no_grad()", t(lambda: SymQuantizer.apply(x, clip, 8, False)))
print("apply no-grad (again, last)", t(lambda: SymQuantizer.apply(x, clip, 8, False)))
xg = x.clone().requires_grad_(True)
print("apply grad input", t(lambda: SymQuantizer.apply(xg, clip, 8, False)))
print("apply no-grad (third)", t(lambda: SymQuantizer.apply(x, clip, 8, False)))
pr = cProfile.Profile(); pr.enable()
for _ in range(3000): SymQuantizer.apply(x, clip, 8, False)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
