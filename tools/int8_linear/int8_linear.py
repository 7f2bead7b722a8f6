"""A consumer for the export format -- a measurement, not part of the product package.

`QuantizeLinear.export_weight()` / `ops.sym_export()` emit SymQuantizer's integer bins + per-row {s, t2}.  This file is the other end:
an inference forward of one QuantizeLinear that multiplies the BINS,

    bins_x = fq_sym_export(x)  (int8, one pass over x: read 2 B, write 1 B per element)
    acc    = int8 x int8 -> int32 GEMM of bins_x and the layer's exported weight bins (the library GEMM: torch._int_mm = hipBLASLt)
    out    = bf16(acc * (1/t2_x[m]) * (1/t2_w[n]))           (fq_int8_epilogue.hip, one pass)

so the fake-quantized operands are never written or re-read in bf16 -- the purpose of §8 f4 -- and the GEMM runs on the int8 MFMA path.
What it is NOT: bit-identical to the reference's forward.  The reference multiplies y = bf16(bin / t2) -- every operand rounded to
bf16 once more -- in a bf16 GEMM; this multiplies the integers exactly and scales once, i.e. it is the arithmetic the fake-quant
SIMULATES.  The two differ by about one bf16 rounding of the OUTPUT: measured relative rms 3.3e-3 at LLaMA-7B's shapes (bf16 eps =
3.9e-3) -- the reference's extra rounding of every operand (2^-9 relative each) does not average out against a sum of random-sign
terms, and each side rounds its output once.  tools/int8_linear/int8_linear_bench.py and tests/test_gpu_int8_consumer.py measure it.
Activation bins of 8-bit bf16 rows can reach +-128 (the reference has no clamp): -128 is an int8, +128 is saturated to +127 and counted (`overflow`).

    python tools/int8_linear/int8_linear.py            # build tools/int8_linear/libfq_int8_epilogue.so (hipcc, gfx950)
"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(HERE, "fq_int8_epilogue.hip")
LIB = os.path.join(HERE, "libfq_int8_epilogue.so")
_lib = None


def build(force=False, verbose=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc if os.path.exists(hipcc) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
           "-fvisibility=hidden", "-fPIC", "-shared", "-Wall", "-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB)
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        L.fq_int8_epilogue.argtypes = [vp, vp, vp, vp, i64, i64, vp]
        L.fq_int8_epilogue.restype = ctypes.c_int
        _lib = L
    return _lib


class Int8Linear:
    """inference forward of one QuantizeLinear (3 <= w_bits <= 8, 3 <= a_bits <= 8, SymQuantizer, row-wise) on its exported bins"""

    def __init__(self, layer):
        import torch
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        if not (3 <= layer.w_bits <= 8 and 3 <= layer.a_bits <= 8) or layer.weight_layerwise or layer.act_layerwise or layer._act_kind != "sym":
            raise ValueError("Int8Linear serves SymQuantizer layers with 3..8-bit row-wise weights and activations")
        if layer.weight.dtype != torch.bfloat16 or not layer.weight.is_cuda:
            raise ValueError("Int8Linear serves bf16 layers on the GPU")
        ex = layer.export_weight(container="int8")            # the weight's bins are static at inference: exported once
        self.w_bins, self.w_scales, self.w_overflow = ex.bins.contiguous(), ex.scales.contiguous(), int(ex.overflow.sum())
        self.a_bits, self.out_features, self.in_features = layer.a_bits, layer.out_features, layer.in_features
        if self.out_features % 8 or self.in_features % 8:
            raise ValueError("torch._int_mm wants out_features and in_features to be multiples of 8")

    def export_input(self, x):
        from llm_qat_amd import ops
        m = x.numel() // self.in_features
        return ops.sym_export(x.reshape(m, self.in_features), self.a_bits, False, container="int8", autocast=False)

    def __call__(self, x, return_parts=False):
        import torch
        ex = self.export_input(x)
        m = ex.bins.shape[0]
        if m <= 16:
            raise ValueError("torch._int_mm wants more than 16 rows")
        acc = torch._int_mm(ex.bins, self.w_bins.t())
        out = torch.empty(x.shape[:-1] + (self.out_features,), dtype=torch.bfloat16, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib().fq_int8_epilogue(acc.data_ptr(), ex.scales.data_ptr(), self.w_scales.data_ptr(), out.data_ptr(), m, self.out_features,
                                        torch.cuda.current_stream().cuda_stream)
        if rc:
            raise RuntimeError(f"fq_int8_epilogue failed (code {rc})")
        return (out, ex, acc) if return_parts else out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
