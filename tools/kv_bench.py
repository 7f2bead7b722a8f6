#!/usr/bin/env python3
"""The K / V hooks (modeling_llama_quant.py:320-327) through the Python drop-in: two SymQuantizer.apply calls vs
quantize_kv (one launch each way), forward + backward, plain bf16 and under bf16 autocast (fp32 results, fp32 gradients).
GPU time per K+V pair from HIP events over many iterations (host-side autograd overhead included on both sides).

    python tools/kv_bench.py   -> gpurun_out/kv_bench.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import llm_qat_amd  # noqa: E402
from llm_qat_amd.utils_quant import SymQuantizer, quantize_kv  # noqa: E402


def main():
    clip = torch.tensor([-2.0, 2.0])
    out = []
    for hidden in (4096, 5120):
        sets = []
        for i in range(6):
            g = torch.Generator(device="cuda").manual_seed(i)
            k = torch.randn(1, 2048, hidden, generator=g, device="cuda").bfloat16().requires_grad_(True)
            v = torch.randn(1, 2048, hidden, generator=g, device="cuda").bfloat16().requires_grad_(True)
            sets.append((k, v, torch.randn(1, 2048, hidden, device="cuda"), torch.randn(1, 2048, hidden, device="cuda")))
        for autocast in (False, True):
            for one in (False, True):
                def step(s):
                    k, v, gk, gv = s
                    k.grad = v.grad = None
                    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                        if one:
                            kq, vq = quantize_kv(k, v, clip, clip, 4)
                        else:
                            kq, vq = SymQuantizer.apply(k, clip, 4, False), SymQuantizer.apply(v, clip, 4, False)
                    torch.autograd.backward([kq, vq], [gk.to(kq.dtype) if kq.dtype != gk.dtype else gk, gv.to(vq.dtype) if vq.dtype != gv.dtype else gv])
                # the gradient casts for the non-autocast case are hoisted out of the timed loop
                tsets = sets if autocast else [(k, v, gk.bfloat16(), gv.bfloat16()) for k, v, gk, gv in sets]
                for i in range(10):
                    step(tsets[i % 6])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                n = 200
                for i in range(n):
                    step(tsets[i % 6])
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / n * 1e3
                row = dict(hidden=hidden, autocast=autocast, impl="quantize_kv (one launch each way)" if one else "two SymQuantizer.apply calls", us_per_kv_fwd_bwd=round(us, 1))
                out.append(row)
                print(row, flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "kv_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
