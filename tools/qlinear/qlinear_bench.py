#!/usr/bin/env python3
"""SURVEY §8 f4a cost model, measured: QuantizeLinear's no-grad forward on the LLaMA-7B shapes,

    unfused   fq pair launch (W + x -> HBM, 4 B/elem) + F.linear (hipBLASLt) on the quantized operands
    fused     fq_sym_row_scales pre-pass(es) (2 B/elem, read only) + fq_qlinear_fwd applying the fake-quant in its
              global -> LDS staging (W only / x only / both), and the same GEMM with nothing quantized (its raw speed)
    ablation  the fused kernel's staging pipeline without MFMAs, and its LDS reads + MFMAs without staging

All kinds are timed on the same buffers in interleaved rounds (median of rounds).   -> gpurun_out/qlinear_bench.json
"""
import json
import os
import statistics
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from llm_qat_amd import _lib  # noqa: E402
import qlinear as QX  # noqa: E402

SHAPES = [("q/k/v/o_proj", 2048, 4096, 4096), ("gate/up_proj", 2048, 4096, 11008), ("down_proj", 2048, 11008, 4096)]


def main():
    L = _lib.lib()
    LQ = QX.lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    code = _lib.DTYPE_BF16
    rounds, iters = 7, 20
    out = []
    for label, m, k, n in SHAPES:
        nsets = 3
        g = torch.Generator(device=dev).manual_seed(7)
        sets = []
        for _ in range(nsets):
            w = (torch.randn(n, k, generator=g, device=dev) * 0.02).bfloat16()
            x = torch.randn(m, k, generator=g, device=dev).bfloat16()
            sets.append(dict(w=w, x=x, wq=torch.empty_like(w), xq=torch.empty_like(x), o=torch.empty(m, n, dtype=torch.bfloat16, device=dev),
                             ws=torch.empty(n, 2, device=dev), xs=torch.empty(m, 2, device=dev)))

        def chk(rc):
            if rc:
                _lib.check(rc, "qlinear_bench")

        def pair(s):
            chk(L.fq_sym_fwd_pair(s["w"].data_ptr(), s["wq"].data_ptr(), n, 4, None, None, 0, s["x"].data_ptr(), s["xq"].data_ptr(), m, 8, None, None, 0,
                                  k, code, 0, 0, -2.0, 2.0, st))

        def quant_x(s):
            chk(L.fq_sym_fwd(s["x"].data_ptr(), s["xq"].data_ptr(), m, k, 8, code, 0, None, None, 0, st))

        def quant_w(s):
            chk(L.fq_sym_fwd(s["w"].data_ptr(), s["wq"].data_ptr(), n, k, 4, code, 0, None, None, 0, st))

        def scales_w(s):
            chk(L.fq_sym_row_scales(s["w"].data_ptr(), s["ws"].data_ptr(), n, k, 4, code, 0, 0, -2.0, 2.0, None, None, 0, st))

        def scales_x(s):
            chk(L.fq_sym_row_scales(s["x"].data_ptr(), s["xs"].data_ptr(), m, k, 8, code, 0, 0, -2.0, 2.0, None, None, 0, st))

        def fused(qa, qw, abl=0, ac=0, xkey=None):
            def fn(s):
                xk = xkey or ("x" if qa else "xq")
                chk(LQ.fq_qlinear_fwd(s[xk].data_ptr(), s["xs"].data_ptr() if qa else None, (s["w"] if qw else s["wq"]).data_ptr(),
                                     s["ws"].data_ptr() if qw else None, s["o"].data_ptr(), m, k, n, code, ac, None, None, abl, st))
            return fn

        kinds = {
            "unfused: pair launch": pair,
            "unfused: F.linear (hipBLASLt)": lambda s: F.linear(s["xq"], s["wq"]),
            "unfused total: pair + F.linear": lambda s: (pair(s), F.linear(s["xq"], s["wq"])),
            "standalone fq_sym_fwd(x) A8": quant_x,
            "standalone fq_sym_fwd(W) W4": quant_w,
            "prepass: row scales W": scales_w,
            "prepass: row scales x": scales_x,
            "fused kernel, nothing quantized (its raw GEMM)": fused(0, 0),
            "fused kernel, W on load": fused(0, 1),
            "fused kernel, x on load": fused(1, 0),
            "fused kernel, W + x on load": fused(1, 1),
            "fused kernel, W on load, autocast arithmetic": fused(0, 1, ac=1),
            "fused kernel, W + x on load, autocast arithmetic": fused(1, 1, ac=1),
            "ablation: staging only (no MFMA), nothing quantized": fused(0, 0, 1),
            "ablation: staging only (no MFMA), W on load": fused(0, 1, 1),
            "ablation: staging only (no MFMA), W + x on load": fused(1, 1, 1),
            "ablation: LDS reads + MFMA only (no staging)": fused(0, 0, 2),
            "fused W-only path total: quant x + scales W + kernel(W on load)": lambda s: (quant_x(s), scales_w(s), fused(0, 1)(s)),
            "fused both path total: scales W + scales x + kernel(W + x on load)": lambda s: (scales_w(s), scales_x(s), fused(1, 1)(s)),
        }
        for s in sets:
            pair(s), scales_w(s), scales_x(s)
        names = list(kinds)
        res = {kname: [] for kname in names}

        def timed(fn):
            for i in range(3):
                fn(sets[i % nsets])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(iters):
                fn(sets[i % nsets])
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters * 1e3

        for r in range(rounds):
            for kname in (names if r % 2 == 0 else names[::-1]):
                res[kname].append(timed(kinds[kname]))
        flops = 2.0 * m * k * n
        row = dict(shape=label, tokens=m, in_features=k, out_features=n, gemm_flops=flops, us={}, tflops={})
        print(f"== {label}: x[{m},{k}] . W[{n},{k}]^T", flush=True)
        for kname in names:
            med = statistics.median(res[kname])
            row["us"][kname] = round(med, 2)
            if "kernel" in kname or "F.linear" in kname or "MFMA" in kname:
                row["tflops"][kname] = round(flops / med / 1e6, 1)
            print(f"   {kname:72s} {med:8.2f} us" + (f"  ({flops / med / 1e6:7.1f} TFLOP/s)" if kname in row["tflops"] else ""), flush=True)
        out.append(row)
        del sets
        torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "qlinear_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
