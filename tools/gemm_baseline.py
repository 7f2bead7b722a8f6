#!/usr/bin/env python3
"""Cost-model inputs for SURVEY §8 f4a (fake-quant fused into the GEMM prologue): what the UNFUSED QuantizeLinear forward
costs today on this device, per LLaMA-7B shape --

    pair_us   one fq_sym_fwd_pair launch (weight [out,in] W4 + input [tokens,in] A8, no side outputs: the no-grad forward)
    gemm_us   F.linear(xq, wq) in bf16 (torch -> hipBLASLt / rocBLAS), operands freshly written by the pair launch
    tf        the GEMM's TFLOP/s

A fused quantize-on-load GEMM has to beat pair_us + gemm_us.   -> gpurun_out/gemm_baseline.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from llm_qat_amd import _lib  # noqa: E402

SHAPES = [  # (label, tokens, in, out)   modeling_llama_quant.py:210-230,:262-289
    ("q/k/v/o_proj", 2048, 4096, 4096),
    ("gate/up_proj", 2048, 4096, 11008),
    ("down_proj", 2048, 11008, 4096),
]


def main():
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    code = _lib.DTYPE_BF16
    out = []
    for label, m, k, n in SHAPES:
        nsets = 4
        g = torch.Generator(device=dev).manual_seed(7)
        sets = []
        for _ in range(nsets):
            w = (torch.randn(n, k, generator=g, device=dev) * 0.02).bfloat16()
            x = torch.randn(m, k, generator=g, device=dev).bfloat16()
            sets.append(dict(w=w, x=x, wq=torch.empty_like(w), xq=torch.empty_like(x)))

        def pair(s):
            rc = L.fq_sym_fwd_pair(s["w"].data_ptr(), s["wq"].data_ptr(), n, 4, None, None, 0, s["x"].data_ptr(), s["xq"].data_ptr(), m, 8, None, None, 0,
                                   k, code, 0, 0, -2.0, 2.0, st)
            if rc:
                _lib.check(rc, "pair")

        def gemm(s):
            return F.linear(s["xq"], s["wq"])

        def both(s):
            pair(s)
            return F.linear(s["xq"], s["wq"])

        def t(fn, iters=50):
            for i in range(5):
                fn(sets[i % nsets])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(iters):
                fn(sets[i % nsets])
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters * 1e3

        for s in sets:
            pair(s)
        rounds = [(t(pair), t(gemm), t(both)) for _ in range(5)]
        p, gm, b = (sorted(r[i] for r in rounds)[len(rounds) // 2] for i in range(3))
        flops = 2.0 * m * k * n
        row = dict(shape=label, tokens=m, in_features=k, out_features=n, pair_us=round(p, 2), gemm_us=round(gm, 2), pair_plus_gemm_us=round(b, 2),
                   gemm_tflops=round(flops / gm / 1e6, 1), fq_bytes=(m + n) * k * 4, gemm_flops=flops)
        out.append(row)
        print(row, flush=True)
        del sets
        torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "gemm_baseline.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
