"""ctypes binding of the C ABI in include/llmqat_fakequant.h.

There is no fallback: if the HIP library is missing or fails to load, every op raises.
"""
import ctypes
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
# LLMQAT_AMD_LIB points the loader at another build of the library (A/B runs of kernel variants: tools/ab_bench.sh) -- the product
# file is never overwritten; fq_build_info() / LIB_PATH say which one is loaded
LIB_PATH = os.environ.get("LLMQAT_AMD_LIB") or os.path.join(HERE, "libllmqat_fakequant.so")
ABI_VERSION = 5

DTYPE_F32, DTYPE_BF16, DTYPE_F16, DTYPE_F64 = 0, 1, 2, 3
SEM_CPU_EAGER, SEM_DEVICE_EAGER = 0, 1

# every symbol include/llmqat_fakequant.h declares (tests check the .so exports them all)
EXPORTS = (
    "fq_version", "fq_build_info", "fq_last_error", "fq_rowwise_workspace_bytes",
    "fq_sym_fwd", "fq_asym_fwd", "fq_sym_fwd_debug", "fq_asym_fwd_debug",
    "fq_ste_bwd", "fq_ste_bwd_rows",
    "fq_ste_mask_bytes", "fq_sym_fwd_train", "fq_asym_fwd_train", "fq_ste_bwd_mask",
    "fq_w12_fwd", "fq_sym_fwd_autocast", "fq_sym_fwd_pair", "fq_ste_bwd_mask_pair", "fq_ste_bwd_mask_wide",
    "fq_export_bins_bytes", "fq_sym_export", "fq_asym_export", "fq_sym_row_scales", "fq_sym_fwd_multi", "fq_ste_bwd_mask_multi", "fq_w12_fwd_rows",
    "fq_rowwise_fwd_v", "fq_sym_fwd_multi_v", "fq_ste_bwd_mask_multi_v", "fq_ste_bwd_v",
)
MAX_TENSORS = 4  # tensors per multi-tensor launch


class FwdTensor(ctypes.Structure):  # fq_fwd_tensor
    _fields_ = [("x", ctypes.c_void_p), ("y", ctypes.c_void_p), ("rows", ctypes.c_int64), ("bits", ctypes.c_int), ("row_bounds", ctypes.c_void_p),
                ("mask", ctypes.c_void_p), ("mask_bytes", ctypes.c_size_t)]


class BwdTensor(ctypes.Structure):  # fq_bwd_tensor
    _fields_ = [("g", ctypes.c_void_p), ("gx", ctypes.c_void_p), ("rows", ctypes.c_int64), ("row_bounds", ctypes.c_void_p), ("mask", ctypes.c_void_p)]



class RowsView(ctypes.Structure):  # fq_rows_view: rows that do not follow one another in memory (strides in elements; n_inner = 0: contiguous)
    _fields_ = [("n_inner", ctypes.c_int64), ("stride_outer", ctypes.c_int64), ("stride_inner", ctypes.c_int64)]


class FwdTensorV(ctypes.Structure):  # fq_fwd_tensor_v
    _fields_ = FwdTensor._fields_ + [("xv", RowsView), ("yv", RowsView)]


class BwdTensorV(ctypes.Structure):  # fq_bwd_tensor_v
    _fields_ = BwdTensor._fields_ + [("gv", RowsView), ("gxv", RowsView)]


BINS_NONE, BINS_INT4, BINS_INT8, BINS_INT16 = 0, 1, 2, 3
ERR_UNSUPPORTED = -8

_lock = threading.Lock()
_lib = None


class FakeQuantLibraryError(RuntimeError):
    pass


def _bind(L):
    vp, i64, i32, f32, sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
    L.fq_version.argtypes = []
    L.fq_version.restype = i32
    L.fq_build_info.argtypes = []
    L.fq_build_info.restype = ctypes.c_char_p
    L.fq_last_error.argtypes = []
    L.fq_last_error.restype = ctypes.c_char_p
    L.fq_rowwise_workspace_bytes.argtypes = [i64, i64, i32]
    L.fq_rowwise_workspace_bytes.restype = sz
    for name in ("fq_sym_fwd", "fq_asym_fwd"):
        f = getattr(L, name)
        f.argtypes = [vp, vp, i64, i64, i32, i32, i32, vp, vp, sz, vp]
        f.restype = i32
    for name in ("fq_sym_fwd_debug", "fq_asym_fwd_debug"):
        f = getattr(L, name)
        f.argtypes = [vp, vp, vp, vp, i64, i64, i32, i32, i32, vp, sz, vp]
        f.restype = i32
    L.fq_ste_bwd.argtypes = [vp, vp, vp, i64, f32, f32, i32, vp]
    L.fq_ste_bwd.restype = i32
    L.fq_ste_bwd_rows.argtypes = [vp, vp, vp, i64, i64, f32, f32, vp, i32, vp]
    L.fq_ste_bwd_rows.restype = i32
    L.fq_ste_mask_bytes.argtypes = [i64, i64, i32]
    L.fq_ste_mask_bytes.restype = sz
    for name in ("fq_sym_fwd_train", "fq_asym_fwd_train"):
        f = getattr(L, name)
        f.argtypes = [vp, vp, i64, i64, i32, i32, i32, f32, f32, vp, vp, sz, vp]
        f.restype = i32
    L.fq_ste_bwd_mask.argtypes = [vp, vp, i64, i64, f32, f32, vp, vp, sz, i32, vp]
    L.fq_ste_bwd_mask.restype = i32
    L.fq_w12_fwd.argtypes = [vp, vp, vp, i64, i64, i32, i32, i32, vp]
    L.fq_w12_fwd.restype = i32
    L.fq_sym_fwd_autocast.argtypes = [vp, vp, i64, i64, i32, i32, i32, i32, f32, f32, vp, vp, sz, vp, sz, vp]
    L.fq_sym_fwd_autocast.restype = i32
    L.fq_sym_fwd_pair.argtypes = [vp, vp, i64, i32, vp, vp, sz, vp, vp, i64, i32, vp, vp, sz, i64, i32, i32, i32, f32, f32, vp]
    L.fq_sym_fwd_pair.restype = i32
    L.fq_ste_bwd_mask_pair.argtypes = [vp, vp, i64, vp, vp, vp, vp, i64, vp, vp, i64, f32, f32, i32, vp]
    L.fq_ste_bwd_mask_pair.restype = i32
    L.fq_ste_bwd_mask_wide.argtypes = [vp, vp, i64, vp, vp, vp, vp, i64, vp, vp, i64, f32, f32, i32, vp]
    L.fq_ste_bwd_mask_wide.restype = i32
    L.fq_export_bins_bytes.argtypes = [i64, i64, i32]
    L.fq_export_bins_bytes.restype = sz
    L.fq_sym_export.argtypes = [vp, vp, vp, vp, i64, i64, i32, i32, i32, i32, i32, vp]
    L.fq_sym_export.restype = i32
    L.fq_asym_export.argtypes = [vp, vp, vp, vp, i64, i64, i32, i32, i32, i32, vp]
    L.fq_asym_export.restype = i32
    L.fq_sym_row_scales.argtypes = [vp, vp, i64, i64, i32, i32, i32, i32, f32, f32, vp, vp, sz, vp]
    L.fq_sym_row_scales.restype = i32
    L.fq_w12_fwd_rows.argtypes = [vp, vp, vp, i64, i64, i32, i32, vp]
    L.fq_w12_fwd_rows.restype = i32
    L.fq_sym_fwd_multi.argtypes = [i32, ctypes.POINTER(FwdTensor), i64, i32, i32, i32, f32, f32, vp]
    L.fq_sym_fwd_multi.restype = i32
    L.fq_ste_bwd_mask_multi.argtypes = [i32, ctypes.POINTER(BwdTensor), i64, f32, f32, i32, i32, vp]
    L.fq_ste_bwd_mask_multi.restype = i32
    rv = ctypes.POINTER(RowsView)
    L.fq_rowwise_fwd_v.argtypes = [i32, vp, rv, vp, rv, i64, i64, i32, i32, i32, f32, f32, vp, vp, sz, vp]
    L.fq_rowwise_fwd_v.restype = i32
    L.fq_sym_fwd_multi_v.argtypes = [i32, ctypes.POINTER(FwdTensorV), i64, i32, i32, i32, f32, f32, vp]
    L.fq_sym_fwd_multi_v.restype = i32
    L.fq_ste_bwd_mask_multi_v.argtypes = [i32, ctypes.POINTER(BwdTensorV), i64, f32, f32, i32, i32, vp]
    L.fq_ste_bwd_mask_multi_v.restype = i32
    L.fq_ste_bwd_v.argtypes = [vp, rv, vp, rv, vp, rv, i64, i64, f32, f32, vp, i32, vp]
    L.fq_ste_bwd_v.restype = i32
    return L


def lib():
    """The loaded library; raises FakeQuantLibraryError (never falls back) if unavailable."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise FakeQuantLibraryError(
                        f"{LIB_PATH} not found: build it with `python llm-qat_amd/build.py` "
                        "(or __graft_entry__.build()). There is no CPU/eager fallback.")
                try:
                    L = ctypes.CDLL(LIB_PATH)
                except OSError as e:  # e.g. libamdhip64 missing
                    raise FakeQuantLibraryError(f"cannot load {LIB_PATH}: {e}") from e
                _bind(L)
                v = L.fq_version()
                if v != ABI_VERSION:
                    raise FakeQuantLibraryError(f"ABI mismatch: library {v}, binding {ABI_VERSION}")
                _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().fq_last_error().decode(errors="replace")
        if rc in (-2,):  # FQ_ERR_BITS
            raise ValueError(f"{what}: {msg}")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")
