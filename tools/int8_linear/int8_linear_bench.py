"""What the integer consumer of the export format buys at LLaMA-7B's projection shapes (seq 2048, bs 1), inference forward of ONE
QuantizeLinear, W4 A8, bf16 weights:

    module            layer(x) under no_grad: the product's default (weight + input fake-quantized in one launch, then F.linear in bf16)
    module_wcached    the same with the weight's fake-quant cached (enable_weight_quant_cache(persistent=True)): x quant + F.linear
    int8              tools/int8_linear.Int8Linear: fq_sym_export(x) -> int8 GEMM over the bins (torch._int_mm) -> scale epilogue

    python tools/int8_linear/int8_linear_bench.py [--json out.json]
"""
import argparse
import importlib.util
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def timed(torch, fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    args = ap.parse_args()
    import torch
    import llm_qat_amd
    from llm_qat_amd.utils_quant import QuantizeLinear
    spec = importlib.util.spec_from_file_location("_fq_int8_linear", os.path.join(HERE, "int8_linear.py"))
    I8 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(I8)
    torch.manual_seed(0)
    rows = []
    for name, k, n in (("down_proj", 11008, 4096), ("gate_proj / up_proj", 4096, 11008), ("q/k/v/o_proj", 4096, 4096)):
        lin = QuantizeLinear(k, n, w_bits=4, a_bits=8).cuda().bfloat16()
        with torch.no_grad():
            lin.weight.copy_(torch.randn(n, k, device="cuda") * 0.02)
        x = torch.randn(1, 2048, k, device="cuda").bfloat16()
        x.view(-1)[:: 997] *= 20.0   # outlier entries, as activations have
        q = I8.Int8Linear(lin)
        with torch.no_grad():
            ref = lin(x)
            out, ex, acc = q(x, return_parts=True)
            t_mod = timed(torch, lambda: lin(x))
            llm_qat_amd.enable_weight_quant_cache(True, persistent=True)
            lin(x)
            t_wc = timed(torch, lambda: lin(x))
            llm_qat_amd.enable_weight_quant_cache(False)
            t_int = timed(torch, lambda: q(x))
            t_exp = timed(torch, lambda: q.export_input(x))
            wt = q.w_bins.t()
            t_mm = timed(torch, lambda: torch._int_mm(ex.bins, wt))
            xq, wq = ex.dequantize(), lin.export_weight(container="int8").dequantize()
            t_gemm = timed(torch, lambda: torch.nn.functional.linear(xq, wq))
        d = (out.double() - ref.double())
        rows.append({"layer": name, "shape": f"x [2048,{k}] @ W [{n},{k}]^T", "us_module": round(t_mod, 1), "us_module_weight_cached": round(t_wc, 1),
                     "us_int8_total": round(t_int, 1), "us_int8_export_x": round(t_exp, 1), "us_int8_gemm": round(t_mm, 1),
                     "us_int8_epilogue_and_host": round(t_int - t_exp - t_mm, 1), "us_bf16_gemm_alone": round(t_gemm, 1),
                     "speedup_vs_module": round(t_mod / t_int, 2), "speedup_vs_module_weight_cached": round(t_wc / t_int, 2),
                     "rel_rms_vs_module": float(d.pow(2).mean().sqrt() / ref.double().pow(2).mean().sqrt()),
                     "max_abs_diff_over_rms": float(d.abs().max() / ref.double().pow(2).mean().sqrt()),
                     "x_bins_saturated": int(ex.overflow.sum()), "w_bins_saturated": q.w_overflow})
        print(json.dumps(rows[-1]), flush=True)
        del lin, q, x, ref, out, ex, acc, xq, wq
        torch.cuda.empty_cache()
    if args.json:
        os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
        with open(args.json, "w") as f:
            json.dump({"what": __doc__, "device": torch.cuda.get_device_name(0), "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
