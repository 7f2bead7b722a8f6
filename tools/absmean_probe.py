#!/usr/bin/env python3
"""Is  w.abs().mean(dim=1, keepdim=True)  (utils_quant.py:205-209 / :219-224 on the GPU: an `abs` kernel that materialises |w|,
then a mean reduction over it)  bit-identical to a reduction that reads w directly,
    torch.linalg.vector_norm(w, 1, dim=1, keepdim=True, dtype=float32) * (1 / n)   rounded once to w.dtype ?
Both accumulate in fp32 inside ATen's gpu_reduce_kernel; the question is whether the summation ORDER is the same (it depends on the
reduce configuration ATen derives from shape / dtype).  Counts mismatching rows over many shapes.   python tools/absmean_probe.py"""
import json

import torch

torch.manual_seed(0)
res = []
shapes = [(4096, 11008), (11008, 4096), (4096, 4096), (5120, 13824), (13824, 5120), (5120, 5120), (256, 256), (688, 256), (256, 688), (37, 1000), (3, 7),
          (1024, 8192), (64, 32768), (2, 65536), (300, 2048), (129, 4100), (4096, 64), (1, 11008)]
for dt in (torch.bfloat16, torch.float16, torch.float32):
    for rows, cols in shapes:
        for scale in (0.02, 1.0):
            w = (torch.randn(rows, cols, device="cuda") * scale).to(dt)
            ref = w.abs().mean(dim=1, keepdim=True)
            s32 = torch.linalg.vector_norm(w, 1, dim=1, keepdim=True, dtype=torch.float32)
            factor = torch.tensor(float(rows), dtype=torch.float32, device="cuda") / torch.tensor(float(rows * cols), dtype=torch.float32, device="cuda")
            cand = (s32 * factor).to(dt)
            cand2 = torch.linalg.vector_norm(w, 1, dim=1, keepdim=True) / cols           # all in w.dtype: expected to differ (double rounding)
            s_abs = w.abs().sum(dim=1, keepdim=True, dtype=torch.float32)                 # control: the materialised |w|, fp32 sum
            cand3 = (s_abs * factor).to(dt)
            res.append({"dtype": str(dt), "shape": [rows, cols], "scale": scale, "rows_differ_norm_f32": int((ref != cand).sum().item()),
                        "rows_differ_norm_native": int((ref != cand2).sum().item()), "rows_differ_abs_sum_f32": int((ref != cand3).sum().item())})
bad = [r for r in res if r["rows_differ_norm_f32"]]
print(json.dumps({"cases": len(res), "cases_with_a_differing_row (vector_norm fp32 path)": len(bad),
                  "cases_with_a_differing_row (abs + fp32 sum control)": sum(1 for r in res if r["rows_differ_abs_sum_f32"]),
                  "cases_with_a_differing_row (native-dtype norm / n)": sum(1 for r in res if r["rows_differ_norm_native"]), "examples": bad[:8]}, indent=1))
