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


def raw_abi():
    """GPU time of the launches themselves (C ABI on preallocated buffers, rotating sets): the autocast K/V path before
    (2 fp32-result forwards with bounds; per tensor an ATen fp32->bf16 cast + the x-re-reading row backward) and after
    (1 paired forward with masks; 1 paired fq_ste_bwd_mask_wide)."""
    from llm_qat_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    code = _lib.DTYPE_BF16
    rows_out = []
    for hidden in (4096, 5120):
        rows, cols = 2048, hidden
        mb = L.fq_ste_mask_bytes(rows, cols, code)
        sets = []
        for i in range(8):
            d = {}
            for t in "kv":
                d["x" + t] = torch.randn(rows, cols, device="cuda").bfloat16()
                d["y" + t] = torch.empty(rows, cols, device="cuda")
                d["g" + t] = torch.randn(rows, cols, device="cuda")
                d["o" + t] = torch.empty(rows, cols, device="cuda", dtype=torch.bfloat16)
                d["s" + t] = torch.empty(rows * 8 + mb, dtype=torch.uint8, device="cuda")
            sets.append(d)

        def chk(rc):
            if rc:
                _lib.check(rc, "kv_bench")

        def fwd_two(d):
            for t in "kv":
                chk(L.fq_sym_fwd_autocast(d["x" + t].data_ptr(), d["y" + t].data_ptr(), rows, cols, 4, code, 1, 1, -2.0, 2.0, d["s" + t].data_ptr(), None, 0, None, 0, st))

        def fwd_pair(d):
            chk(L.fq_sym_fwd_pair(d["xk"].data_ptr(), d["yk"].data_ptr(), rows, 4, d["sk"].data_ptr(), d["sk"].data_ptr() + rows * 8, mb,
                                  d["xv"].data_ptr(), d["yv"].data_ptr(), rows, 4, d["sv"].data_ptr(), d["sv"].data_ptr() + rows * 8, mb,
                                  cols, code, 1, 2, -2.0, 2.0, st))

        def bwd_two(d):  # bounds only: cast, then re-read x
            for t in "kv":
                g16 = d["g" + t].to(torch.bfloat16)
                chk(L.fq_ste_bwd_rows(g16.data_ptr(), d["x" + t].data_ptr(), d["o" + t].data_ptr(), rows, cols, -2.0, 2.0, d["s" + t].data_ptr(), code, st))

        def bwd_pair(d):
            chk(L.fq_ste_bwd_mask_wide(d["gk"].data_ptr(), d["ok"].data_ptr(), rows, d["sk"].data_ptr(), d["sk"].data_ptr() + rows * 8,
                                       d["gv"].data_ptr(), d["ov"].data_ptr(), rows, d["sv"].data_ptr(), d["sv"].data_ptr() + rows * 8,
                                       cols, -2.0, 2.0, code, st))

        def t(fn, n=200):
            for i in range(10):
                fn(sets[i % 8])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(n):
                fn(sets[i % 8])
            e1.record()
            torch.cuda.synchronize()
            return round(e0.elapsed_time(e1) / n * 1e3, 2)

        for d in sets:
            fwd_two(d)
        a, c = t(fwd_two), t(bwd_two)
        for d in sets:
            fwd_pair(d)
        b, e = t(fwd_pair), t(bwd_pair)
        row = dict(hidden=hidden, level="C ABI launches, bf16 autocast, K+V [2048,hidden]", fwd_two_calls_us=a, fwd_one_launch_us=b,
                   bwd_cast_plus_rows_x2_us=c, bwd_one_launch_us=e)
        rows_out.append(row)
        print(row, flush=True)
        del sets
    return rows_out


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
    out += raw_abi()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "kv_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
