"""The fused quantize-on-load GEMM EXPERIMENT -- not part of the product package.

Round 2 built `fq_qlinear_fwd` (tools/qlinear/fq_qlinear.hip: a bf16 MFMA GEMM whose global -> LDS staging applies the fake-quant),
measured it slower than the product's unfused path on every LLaMA shape, and round 4 moved it here: its own small
library, its tests (tests/test_gpu_qlinear.py) and its bench (tools/qlinear/qlinear_bench.py) keep it reproducible.

    python tools/qlinear/qlinear.py            # build tools/qlinear/libfq_qlinear_exp.so (hipcc, gfx950)
"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(HERE, "fq_qlinear.hip")
SRC_DIRECT = os.path.join(HERE, "fq_qlinear_direct.hip")   # round 5: W direct-to-VGPR as the MFMA operand, x by LDS-DMA
LIB = os.path.join(HERE, "libfq_qlinear_exp.so")
_DEPS = [SRC, SRC_DIRECT] + [os.path.join(ROOT, "llm-qat_amd", "csrc", f) for f in ("fq_device.h", "fq_kernels.h", "fq_launch.h")] + [
    os.path.join(ROOT, "include", "llmqat_fakequant.h")]
_lib = None


def build(force=False, verbose=False):
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in _DEPS):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc if os.path.exists(hipcc) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
           "-fvisibility=hidden", "-fPIC", "-shared", "-Wall", "-Wno-unused-variable", "-Wno-unused-but-set-variable", "-Wno-unused-function", "-o", LIB, SRC, SRC_DIRECT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB)
        vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        L.fq_qlinear_fwd.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i32, i32, vp, vp, i32, vp]
        L.fq_qlinear_fwd.restype = i32
        L.fq_qlinear_last_error.restype = ctypes.c_char_p
        L.fq_qlinear_direct_fwd.argtypes = [vp, vp, vp, vp, i64, i64, i64, i32, i32, i32, i32, vp]
        L.fq_qlinear_direct_fwd.restype = i32
        _lib = L
    return _lib


def check(rc, what):
    if rc:
        raise RuntimeError(f"{what} failed (code {rc}): {lib().fq_qlinear_last_error().decode(errors='replace')}")


def qlinear_forward(x, weight, w_bits, a_bits, quantize_x=True, quantize_w=True, autocast=None, dump=False, ablation=0, x_scales=None, w_scales=None):
    """out = fq(x) @ fq(weight).T, bf16, the fake-quant applied while the GEMM loads its operands.  quantize_x / quantize_w = False
    multiplies that operand as it is.  The scale pre-passes are the PRODUCT's (llm_qat_amd.ops.sym_row_scales) unless given.
    -> out, or (out, staged_x, staged_w) with dump=True (the operand tiles exactly as the MFMAs saw them); None when the shape /
    alignment is not served."""
    import torch
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from llm_qat_amd import ops
    if x.dtype != torch.bfloat16 or weight.dtype != torch.bfloat16 or not (x.is_cuda and weight.is_cuda) or weight.dim() != 2:
        return None
    k = weight.shape[1]
    if x.shape[-1] != k or not (x.is_contiguous() and weight.is_contiguous()) or x.numel() == 0:
        return None
    n, m = weight.shape[0], x.numel() // k
    ac = ops.autocast_active(x) if autocast is None else bool(autocast)
    if quantize_x and x_scales is None:
        x_scales = ops.sym_row_scales(x.reshape(m, k), a_bits, False, autocast=ac)
    if quantize_w and w_scales is None:
        w_scales = ops.sym_row_scales(weight, w_bits, False, autocast=ac)
    out = torch.empty(x.shape[:-1] + (n,), dtype=torch.bfloat16, device=x.device)
    dx = torch.empty_like(x) if dump else None
    dw = torch.empty_like(weight) if dump else None
    with torch.cuda.device(x.device):
        rc = lib().fq_qlinear_fwd(x.data_ptr(), x_scales.data_ptr() if quantize_x else None, weight.data_ptr(), w_scales.data_ptr() if quantize_w else None,
                                  out.data_ptr(), m, k, n, 1, 1 if ac else 0, dx.data_ptr() if dump else None, dw.data_ptr() if dump else None,
                                  int(ablation), torch.cuda.current_stream().cuda_stream)
    if rc == -8:
        return None
    check(rc, "qlinear_forward")
    return (out, dx, dw) if dump else out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
