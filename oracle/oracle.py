"""ctypes front-end of oracle/fq_oracle.c (numpy in, numpy out) -- TEST INFRASTRUCTURE ONLY.

16-bit float tensors travel as raw uint16 bit patterns, fp32 as float32.  See fq_oracle.c
for the reference lines each entry point restates and for the parity status (pinned by
tests/golden/*.npz).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfq_oracle.so")
DTYPES = {"fp32": 0, "bf16": 1, "fp16": 2}
SEM_CPU, SEM_DEVICE = 0, 1
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("fq_oracle.c", "fq_oracle_f64.c", "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libfq_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
        L.fqo_sym_fwd.argtypes = [vp, vp, vp, vp, i64, i64, i32, i32, i32]
        L.fqo_asym_fwd.argtypes = [vp, vp, vp, vp, vp, i64, i64, i32, i32, i32]
        L.fqo_ste_bwd.argtypes = [vp, vp, vp, i64, f32, f32, i32]
        L.fqo_w12_fwd.argtypes = [vp, vp, vp, vp, i64, i64, i32, i32]
        L.fqo_sym_fwd_autocast.argtypes = [vp, vp, vp, vp, i64, i64, i32, i32, i32, i32]
        L.fqo_ste_bwd_wide.argtypes = [vp, vp, vp, i64, f32, f32, i32]
        L.fqo_ste_bwd_wide.restype = ctypes.c_int
        L.fqo_export.argtypes = [vp, vp, vp, vp, i64, i64, i32, i32, i32, i32, i32, i32]
        L.fqo_export.restype = ctypes.c_int
        for f in (L.fqo_sym_fwd, L.fqo_asym_fwd, L.fqo_ste_bwd, L.fqo_w12_fwd, L.fqo_version, L.fqo_sym_fwd_autocast):
            f.restype = ctypes.c_int
        L.fqo64_sym_fwd.argtypes = [vp, vp, vp, vp, i64, i64, i32]
        L.fqo64_asym_fwd.argtypes = [vp, vp, vp, vp, vp, i64, i64, i32, i32]
        L.fqo64_ste_bwd.argtypes = [vp, vp, vp, i64, f32, f32]
        L.fqo64_w12_fwd.argtypes = [vp, vp, vp, vp, i64, i64, i32]
        for f in (L.fqo64_sym_fwd, L.fqo64_asym_fwd, L.fqo64_ste_bwd, L.fqo64_w12_fwd):
            f.restype = ctypes.c_int
        _lib = L
    return _lib


def _check(x, dtype):
    want = np.float64 if dtype == "fp64" else np.float32 if dtype == "fp32" else np.uint16
    if x.dtype != want:
        raise TypeError(f"{dtype} data must be {want}, got {x.dtype}")
    return np.ascontiguousarray(x)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def sym_fwd(x, rows, cols, bits, dtype, sem=SEM_CPU, want_idx=True):
    """-> (y, idx int32 or None, scale float32[rows]); x is any-shape array of rows*cols elements."""
    x = _check(x, dtype)
    assert x.size == rows * cols
    y = np.empty_like(x)
    idx = np.empty(x.shape, np.int32) if want_idx else None
    if dtype == "fp64":   # fq_oracle_f64.c: every op in double (sem does not matter for Sym); the scale comes back as float64
        scale = np.empty(rows, np.float64)
        rc = lib().fqo64_sym_fwd(_p(x), _p(y), _p(idx), _p(scale), rows, cols, bits)
        if rc:
            raise ValueError(f"fqo64_sym_fwd rc={rc}")
        return y, idx, scale
    scale = np.empty(rows, np.float32)
    rc = lib().fqo_sym_fwd(_p(x), _p(y), _p(idx), _p(scale), rows, cols, bits, DTYPES[dtype], sem)
    if rc:
        raise ValueError(f"fqo_sym_fwd rc={rc}")
    return y, idx, scale


def sym_fwd_autocast(x, rows, cols, bits, dtype, wide=True, sem=SEM_DEVICE, want_scale=False):
    """SymQuantizer under CUDA autocast on a 16-bit tensor -> (y float32 if wide else dtype bits, idx[, s float32[rows]]).
    sem: how `max + 1e-6` treats the scalar -- SEM_DEVICE (default: what a real torch.autocast("cuda") run computes) or SEM_CPU
    (the reference's CPU behaviour; the "cpu" cases of tests/golden/autocast.npz)."""
    x = _check(x, dtype)
    y = np.empty(x.shape, np.float32 if wide else np.uint16)
    idx = np.empty(x.shape, np.int32)
    scale = np.empty(rows, np.float32)
    rc = lib().fqo_sym_fwd_autocast(_p(x), _p(y), _p(idx), _p(scale), rows, cols, bits, DTYPES[dtype], 1 if wide else 0, sem)
    if rc:
        raise ValueError(f"fqo_sym_fwd_autocast rc={rc}")
    return (y, idx, scale) if want_scale else (y, idx)


def ste_bwd_wide(g32, x, lo, hi, dtype):
    """STE backward behind the fp32-result forward: fp32 gradient in, gradient in x's (16-bit) dtype out"""
    g32, x = _check(g32, "fp32"), _check(x, dtype)
    assert g32.size == x.size
    gx = np.empty(x.shape, np.uint16)
    rc = lib().fqo_ste_bwd_wide(_p(g32), _p(x), _p(gx), x.size, lo, hi, DTYPES[dtype])
    if rc:
        raise ValueError(f"fqo_ste_bwd_wide rc={rc}")
    return gx


def asym_fwd(x, rows, cols, bits, dtype, sem=SEM_CPU, want_idx=True):
    """-> (y, idx, alpha[rows], beta[rows])"""
    x = _check(x, dtype)
    assert x.size == rows * cols
    y = np.empty_like(x)
    idx = np.empty(x.shape, np.int32) if want_idx else None
    if dtype == "fp64":
        alpha, beta = np.empty(rows, np.float64), np.empty(rows, np.float64)
        rc = lib().fqo64_asym_fwd(_p(x), _p(y), _p(idx), _p(alpha), _p(beta), rows, cols, bits, sem)
        if rc:
            raise ValueError(f"fqo64_asym_fwd rc={rc}")
        return y, idx, alpha, beta
    alpha = np.empty(rows, np.float32)
    beta = np.empty(rows, np.float32)
    rc = lib().fqo_asym_fwd(_p(x), _p(y), _p(idx), _p(alpha), _p(beta), rows, cols, bits, DTYPES[dtype], sem)
    if rc:
        raise ValueError(f"fqo_asym_fwd rc={rc}")
    return y, idx, alpha, beta


CONTAINERS = {"int4": 1, "int8": 2, "int16": 3}


def export(kind, x, rows, cols, bits, container, dtype, sem=None, autocast=False):
    """packed bins (uint8 bytes [rows, row_bytes]), scales float32[rows, 2], overflow int32[rows].
    sem=None: SEM_CPU, or SEM_DEVICE with autocast (as sym_fwd_autocast)."""
    if sem is None:
        sem = SEM_DEVICE if autocast else SEM_CPU
    x = _check(x, dtype)
    assert x.size == rows * cols
    row_bytes = {"int4": (cols + 1) // 2, "int8": cols, "int16": cols * 2}[container]
    bins = np.zeros((rows, row_bytes), np.uint8)
    scales = np.empty((rows, 2), np.float32)
    over = np.empty(rows, np.int32)
    rc = lib().fqo_export(_p(x), _p(bins), _p(scales), _p(over), rows, cols, bits, CONTAINERS[container], DTYPES[dtype], sem,
                          1 if kind == "asym" else 0, 1 if autocast else 0)
    if rc:
        raise ValueError(f"fqo_export rc={rc}")
    return bins, scales, over


def unpack_bins(bins, cols, container, signed):
    """uint8 [rows, row_bytes] -> int32 [rows, cols]"""
    b = np.asarray(bins, np.uint8)
    if container == "int8":
        v = b.astype(np.int32)
        return np.where(v >= 128, v - 256, v) if signed else v
    if container == "int16":
        v = b[:, 0::2].astype(np.int32) | (b[:, 1::2].astype(np.int32) << 8)
        return np.where(v >= 32768, v - 65536, v) if signed else v
    lo, hi = (b & 0xF).astype(np.int32), (b >> 4).astype(np.int32)
    v = np.empty((b.shape[0], b.shape[1] * 2), np.int32)
    v[:, 0::2], v[:, 1::2] = lo, hi
    v = v[:, :cols]
    return np.where(v >= 8, v - 16, v) if signed else v


def ste_bwd(g, x, lo, hi, dtype):
    g, x = _check(g, dtype), _check(x, dtype)
    assert g.size == x.size
    gx = np.empty_like(g)
    if dtype == "fp64":
        rc = lib().fqo64_ste_bwd(_p(g), _p(x), _p(gx), g.size, lo, hi)
        if rc:
            raise ValueError(f"fqo64_ste_bwd rc={rc}")
        return gx
    rc = lib().fqo_ste_bwd(_p(g), _p(x), _p(gx), g.size, lo, hi, DTYPES[dtype])
    if rc:
        raise ValueError(f"fqo_ste_bwd rc={rc}")
    return gx


def w12_fwd(w, rows, cols, w_bits, dtype, scale_in=None):
    """1-/2-bit weight branch of QuantizeLinear -> (q, scale[rows])"""
    w = _check(w, dtype)
    q = np.empty_like(w)
    if dtype == "fp64":
        sc = np.empty(rows, np.float64)
        si = None if scale_in is None else np.ascontiguousarray(scale_in, np.float64)
        rc = lib().fqo64_w12_fwd(_p(w), _p(q), _p(sc), _p(si), rows, cols, w_bits)
        if rc:
            raise ValueError(f"fqo64_w12_fwd rc={rc}")
        return q, sc
    sc = np.empty(rows, np.float32)
    si = None if scale_in is None else np.ascontiguousarray(scale_in, np.float32)
    rc = lib().fqo_w12_fwd(_p(w), _p(q), _p(sc), _p(si), rows, cols, w_bits, DTYPES[dtype])
    if rc:
        raise ValueError(f"fqo_w12_fwd rc={rc}")
    return q, sc


def rows_cols(shape, layerwise):
    """How the reference's granularity rules map a tensor shape to [rows, cols]
    (utils_quant.py:50-70): layerwise -> one row; ndim<=3 -> last dim; 4-D -> (d0*d1, d2*d3)."""
    n = int(np.prod(shape)) if len(shape) else 1
    if layerwise:
        return 1, n
    if len(shape) <= 3:
        cols = shape[-1] if len(shape) else 1
        return (n // cols if cols else 0), cols
    if len(shape) == 4:
        return shape[0] * shape[1], shape[2] * shape[3]
    raise ValueError("ndim >= 5")  # utils_quant.py:70
