"""Functional front-end over the C ABI: torch tensors in, torch tensors out.

Everything here is host-side plumbing (shape -> [rows, cols], dtype codes, stream handle,
workspace); the arithmetic happens in the HIP kernels.  CPU tensors are rejected: this
package is the MI355X path and has no CPU implementation.
"""
import os

import torch

from . import _lib

_DTYPES = {torch.float32: _lib.DTYPE_F32, torch.bfloat16: _lib.DTYPE_BF16, torch.float16: _lib.DTYPE_F16, torch.float64: _lib.DTYPE_F64}
_SEM_NAMES = {"cpu_eager": _lib.SEM_CPU_EAGER, "device_eager": _lib.SEM_DEVICE_EAGER}
# Default since round 5: a CUDA tensor gets what the reference computes ON A GPU ("device_eager": ATen's GPU kernels keep an added Python
# scalar in fp32 and turn `.div(python scalar)` into a multiply by the fp32 reciprocal) -- the drop-in runs on the GPU, so that is the
# arithmetic it replaces; pinned to the reference's own code by tests/golden/device_scalars.npz and to live ATen on the MI355X
# (tests/test_gpu_device_scalars.py).  "cpu_eager" (the reference run on CPU tensors, the arithmetic of the round-1 fixtures) stays a switch;
# CPU tensors served through allow_cpu_tensors() run ATen's own CPU kernels and are not affected by either.
_semantics = _SEM_NAMES[os.environ.get("LLMQAT_AMD_SEMANTICS", "device_eager")]


def set_semantics(name):
    """'device_eager' (default: bit-equal to the reference's eager ops run on a GPU, outside autocast) or 'cpu_eager' (bit-equal to the
    reference run on CPU).  They differ only for bf16 rows whose |max| is below ~3e-4, fp16 rows with |max| in [2^-13, 2^-12) and for
    fp32 AsymQuantizer (DESIGN.md "Numerics"); under autocast the arithmetic is the device's either way."""
    global _semantics
    _semantics = _SEM_NAMES[name]


def get_semantics():
    return "device_eager" if _semantics == _lib.SEM_DEVICE_EAGER else "cpu_eager"


# Under torch.autocast("cuda") the reference runs on the device by definition, where ATen keeps the `+ 1e-6` scalar in fp32: the
# autocast arithmetic is always launched with device-eager scalars (the C ABI takes `sem` there too: the parity tests drive both).
_SEM_AUTOCAST = _lib.SEM_DEVICE_EAGER


def rows_cols(shape, layerwise):
    """Granularity rules of utils_quant.py:50-70 -> the [rows, cols] view the kernels take."""
    n = 1
    for d in shape:
        n *= d
    if layerwise:
        return 1, n
    nd = len(shape)
    if nd <= 3:
        cols = shape[-1] if nd else 1
        return (n // cols if cols else 0), cols
    if nd == 4:
        return shape[0] * shape[1], shape[2] * shape[3]
    raise ValueError(f"fake-quant expects at most 4 dimensions, got {nd}")  # utils_quant.py:70


def bits_arg(num_bits):
    """`num_bits` as the Python int the kernels take.  The reference computes `2 ** (num_bits - 1) - 1` in Python and then
    `int / Tensor` (= reciprocal * int, models/utils_quant.py:71), so an integral float means the same thing; a NON-integral float would
    mean a fractional number of levels and a TENSOR would turn that line into a true Tensor / Tensor division (different roundings) --
    neither is what any caller of the reference passes, and neither is served: refused loudly rather than silently truncated."""
    if isinstance(num_bits, torch.Tensor):
        raise TypeError("num_bits must be a Python int (a tensor changes the reference's arithmetic to a true division: not served)")
    b = int(num_bits)
    if b != num_bits:
        raise ValueError(f"num_bits must be integral, got {num_bits!r}")
    return b


def _prep(x, what):
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"{what}: expected a torch.Tensor, got {type(x).__name__}")
    if x.device.type != "cuda":
        raise RuntimeError(f"{what}: tensor is on '{x.device}'. llm_qat_amd runs on MI355X only and has no CPU "
                           "fallback; move the tensor to the GPU (or use the reference implementation on CPU).")
    code = _DTYPES.get(x.dtype)
    if code is None:
        raise NotImplementedError(f"{what}: dtype {x.dtype} is not supported (float32, bfloat16, float16, float64 are)")
    return code


def _on_device(x, call):
    """run `call(stream)` with x's device current (the common case -- it already is -- costs one C call)"""
    idx = x.device.index
    cur = _cur_dev()
    if idx is None or idx == cur:
        return call(_raw_stream(cur))
    with torch.cuda.device(idx):
        return call(_raw_stream(idx))


class _DeviceOf:
    """Make x's device current for the launch if it is not already."""

    def __init__(self, x):
        self.idx = x.device.index
        self.prev = None

    def __enter__(self):
        cur = _cur_dev()
        if self.idx is not None and self.idx != cur:
            self.prev = cur
            torch.cuda.set_device(self.idx)

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)


# The raw current-stream handle / current device index straight from the C bindings (what Inductor's generated code uses):
# 0.2 us per call instead of the 2-3 us `torch.cuda.current_stream().cuda_stream` spends building a Stream object.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_dev = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device
if _raw_stream is None:
    def _raw_stream(idx):
        return torch.cuda.current_stream(idx).cuda_stream


def _stream(x):
    """the current stream of x's device, as the `hipStream_t` the C ABI takes"""
    idx = x.device.index
    return _raw_stream(_cur_dev() if idx is None else idx)


_ws_bytes_memo = {}


def _ws_bytes(rows, cols, code):
    k = (rows, cols, code)
    v = _ws_bytes_memo.get(k)
    if v is None:
        v = _ws_bytes_memo[k] = _lib.lib().fq_rowwise_workspace_bytes(rows, cols, code)
    return v


def _empty_input(what, x, layerwise):
    """A tensor without elements, as the reference treats it (models/utils_quant.py:50-68 / :110-122; asked of the live reference):
    a reduction over NOTHING raises what torch raises there -- RuntimeError for the layerwise (whole-tensor) max, and for the 4-D
    branch's `view(d0, d1, -1)` when d0 * d1 == 0 (the -1 is then ambiguous); IndexError for an empty reduction dimension (the last
    one; d2 * d3 == 0 in the 4-D branch) -- while zero ROWS of a non-empty last dimension simply give an empty result."""
    if layerwise:
        raise RuntimeError(f"{what}: max(): Expected reduction dim to be specified for input.numel() == 0 (layerwise reduction of an empty tensor)")
    if x.dim() == 4:
        if x.shape[0] * x.shape[1] == 0:
            raise RuntimeError(f"{what}: cannot reshape tensor of 0 elements into shape [{x.shape[0]}, {x.shape[1]}, -1] because the unspecified "
                               "dimension size -1 can be any value and is ambiguous")
        raise IndexError(f"{what}: max(): Expected reduction dim 2 to have non-zero size.")
    if x.shape[-1] == 0:
        raise IndexError(f"{what}: max(): Expected reduction dim {x.dim() - 1} to have non-zero size.")


# ---- rows that do not follow one another in memory ("last dim contiguous, rows strided": slices, chunk(), transpose(0, 1) of a 3-D tensor)
# are served IN the kernels (the C ABI's fq_rows_view, ABI 5): no .contiguous() in front of the launch, no copy_ behind it, and the result
# keeps the layout the reference's elementwise ops give it (torch.empty_like's rule: the input's strides when it is dense, else contiguous).
# Anything else that is not contiguous -- a strided LAST dimension, 4-D views, layerwise -- keeps the copy path.
def rows_view(t, layerwise=False):
    """-> None: contiguous (no view needed) | (n_inner, stride_outer, stride_inner) in elements | False: needs a copy"""
    if t.is_contiguous():
        return None
    nd = t.dim()
    if layerwise or nd < 2 or nd > 3 or t.shape[-1] < 2 or t.stride(-1) != 1:
        return False
    if nd == 2:
        return (t.shape[0], 0, t.stride(0))
    return (t.shape[1], t.stride(0), t.stride(1))


def check_4d(x, layerwise):
    """the reference flattens a 4-D input with `input.view(d0, d1, -1)` (models/utils_quant.py:63 / :127): a layout for which that view does
    not exist raises there -- and here, with torch's own exception, instead of being served through a copy"""
    if x.dim() == 4 and not layerwise and not x.is_contiguous():
        x.view(x.shape[0], x.shape[1], -1)


def _rv(v):
    return _lib.RowsView(*v) if v else _lib.RowsView(0, 0, 0)


def _strided_out(x, dtype=None):
    """the result tensor for a strided input and its view: empty_like keeps a dense input's strides and makes everything else contiguous,
    exactly what the reference's elementwise ops (TensorIterator) do"""
    y = torch.empty_like(x) if dtype is None else torch.empty_like(x, dtype=dtype)
    return y, rows_view(y)


_views_served = 0   # launches that took a view instead of a copy (tests read it)


def _rowwise(kind, x, num_bits, layerwise, want_bounds, debug):
    what = f"{kind}_quantize"
    code = _prep(x, what)
    rows, cols = rows_cols(tuple(x.shape), layerwise)
    if x.numel() == 0:
        _empty_input(what, x, layerwise)
        return torch.empty_like(x), None, None, None
    check_4d(x, layerwise)
    L = _lib.lib()
    xv = rows_view(x, layerwise)
    if xv and not debug and code != _lib.DTYPE_F64:   # strided rows: served in the kernel
        y, yv = _strided_out(x)
        if yv is not False:
            bounds = torch.empty((rows, 2), dtype=torch.float32, device=x.device) if want_bounds else None
            with _DeviceOf(x):
                rc = L.fq_rowwise_fwd_v(1 if kind == "asym" else 0, x.data_ptr(), _rv(xv), y.data_ptr(), _rv(yv), rows, cols, int(num_bits), code, _semantics,
                                        0.0, 0.0, bounds.data_ptr() if want_bounds else None, None, 0, _stream(x))
            if rc != _lib.ERR_UNSUPPORTED:
                _lib.check(rc, what)
                global _views_served
                _views_served += 1
                return y, bounds, None, None
    xc = x if x.is_contiguous() else x.contiguous()
    y = torch.empty_like(xc)
    ws_bytes = _ws_bytes(rows, cols, code)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes else None
    ws_ptr = ws.data_ptr() if ws is not None else None
    bounds = idx = scale = None
    if code == _lib.DTYPE_F64:
        want_bounds = False   # float64: a correctness path without training-mode side buffers (the backward re-reads x)
    with _DeviceOf(x):
        if debug:
            idx = torch.empty(xc.shape, dtype=torch.int32, device=x.device)
            scale = torch.empty((rows,) if kind == "sym" else (rows, 2), dtype=torch.float32, device=x.device)
            fn = L.fq_sym_fwd_debug if kind == "sym" else L.fq_asym_fwd_debug
            rc = fn(xc.data_ptr(), y.data_ptr(), idx.data_ptr(), scale.data_ptr(), rows, cols, int(num_bits), code,
                    _semantics, ws_ptr, ws_bytes, _stream(x))
        else:
            if want_bounds:
                bounds = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
            fn = L.fq_sym_fwd if kind == "sym" else L.fq_asym_fwd
            rc = fn(xc.data_ptr(), y.data_ptr(), rows, cols, int(num_bits), code, _semantics,
                    bounds.data_ptr() if bounds is not None else None, ws_ptr, ws_bytes, _stream(x))
    _lib.check(rc, what)
    if xc is not x:  # keep the input's strides, as the reference's elementwise ops do
        out = torch.empty_like(x)
        out.copy_(y)
        y = out
        if idx is not None:
            idx = idx.reshape(x.shape)
    return y, bounds, idx, scale


# ---- lean path used by the autograd Functions: one side allocation (row bounds + STE mask), memoised sizes ----
_mask_bytes_memo = {}


def _mask_bytes(rows, cols, code):
    k = (rows, cols, code)
    v = _mask_bytes_memo.get(k)
    if v is None:
        v = _mask_bytes_memo[k] = _lib.lib().fq_ste_mask_bytes(rows, cols, code)
    return v


def _aligned(g):
    """contiguous and 16-byte aligned (a gradient can be an offset view into a larger buffer)"""
    g = g if g.is_contiguous() else g.contiguous()
    return g.clone() if g.data_ptr() & 15 else g


def train_forward(kind, x, num_bits, layerwise, lo, hi):
    """-> (y, side, rows, cols) or None.  `side` is one uint8 buffer: float[rows][2] bounds followed by the mask."""
    if x.device.type != "cuda":
        _prep(x, f"{kind}_quantize")
    code = _DTYPES.get(x.dtype)
    if code is None:
        _prep(x, f"{kind}_quantize")
    if x.numel() == 0:
        return None
    xv = None
    if not x.is_contiguous():
        check_4d(x, layerwise)
        xv = rows_view(x, layerwise)
        if not xv:
            return None
    rows, cols = rows_cols(tuple(x.shape), layerwise)
    mbytes = _mask_bytes(rows, cols, code)
    if not mbytes:
        return None
    L = _lib.lib()
    if xv:   # strided rows: served in the kernel (the side buffer is indexed by row number and stays dense)
        y, yv = _strided_out(x)
        if yv is False:
            return None
        side = torch.empty(rows * 8 + mbytes, dtype=torch.uint8, device=x.device)
        sp = side.data_ptr()
        with _DeviceOf(x):
            rc = L.fq_rowwise_fwd_v(1 if kind == "asym" else 0, x.data_ptr(), _rv(xv), y.data_ptr(), _rv(yv), rows, cols, int(num_bits), code, _semantics,
                                    lo, hi, sp, sp + rows * 8, mbytes, _stream(x))
        if rc:
            if rc == _lib.ERR_UNSUPPORTED:
                return None
            _lib.check(rc, f"{kind}_quantize_train")
        global _views_served
        _views_served += 1
        return y, side, rows, cols
    y = torch.empty_like(x)
    side = torch.empty(rows * 8 + mbytes, dtype=torch.uint8, device=x.device)
    sp = side.data_ptr()
    fn = L.fq_sym_fwd_train if kind == "sym" else L.fq_asym_fwd_train
    dev = x.device.index
    cur = _cur_dev()
    if dev is not None and dev != cur:
        with torch.cuda.device(dev):
            rc = fn(x.data_ptr(), y.data_ptr(), rows, cols, int(num_bits), code, _semantics, lo, hi, sp, sp + rows * 8, mbytes, _raw_stream(dev))
    else:
        rc = fn(x.data_ptr(), y.data_ptr(), rows, cols, int(num_bits), code, _semantics, lo, hi, sp, sp + rows * 8, mbytes, _raw_stream(cur))
    if rc:
        if rc == _lib.ERR_UNSUPPORTED:
            return None
        _lib.check(rc, f"{kind}_quantize_train")
    return y, side, rows, cols


def _mask_backward_v(gs, sides, rows, cols, lo, hi, code, wide, inplace=None, out_dtype=None):
    """fq_ste_bwd_mask_multi_v over 1..2 gradients of which at least one has strided rows -> [gx] or None (not served: copy path)"""
    n = len(gs)
    arr = (_lib.BwdTensorV * n)()
    outs = []
    for i, (g, sd, r) in enumerate(zip(gs, sides, rows)):
        gv = rows_view(g)
        if gv is False or g.data_ptr() & 15:
            return None
        if inplace and inplace[i]:
            o, ov = g, gv
        else:
            o, ov = _strided_out(g, out_dtype)
            if ov is False:
                return None
        outs.append(o)
        sp = sd.data_ptr()
        arr[i] = _lib.BwdTensorV(g.data_ptr(), o.data_ptr(), r, sp, sp + r * 8, _rv(gv), _rv(ov))
    with _DeviceOf(gs[0]):
        rc = _lib.lib().fq_ste_bwd_mask_multi_v(n, arr, cols, float(lo), float(hi), code, 1 if wide else 0, _stream(gs[0]))
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "ste_backward_mask[strided]")
    global _views_served
    _views_served += 1
    return outs


def train_backward(grad_output, side, rows, cols, lo, hi, inplace=False):
    """inplace: mask the gradient where it stands and return it (fq_ste_bwd_mask with gx == g): rows that cannot clip are
    not touched, so a weight's gradient costs a launch and no traffic.  Only for callers that own grad_output exclusively."""
    code = _DTYPES.get(grad_output.dtype)
    if code is None or grad_output.device.type != "cuda":
        _prep(grad_output, "ste_backward")
    if not grad_output.is_contiguous() and grad_output.numel():   # strided rows: masked in the kernel, no .contiguous() copy
        res = _mask_backward_v([grad_output], [side], [rows], cols, lo, hi, code, False, [inplace])
        if res is not None:
            return res[0]
    g = _aligned(grad_output)
    gx = g if inplace else torch.empty_like(g)
    sp = side.data_ptr()
    dev = g.device.index
    L = _lib.lib()
    cur = _cur_dev()
    if dev is not None and dev != cur:
        with torch.cuda.device(dev):
            rc = L.fq_ste_bwd_mask(g.data_ptr(), gx.data_ptr(), rows, cols, lo, hi, sp, sp + rows * 8, side.numel() - rows * 8, code, _raw_stream(dev))
    else:
        rc = L.fq_ste_bwd_mask(g.data_ptr(), gx.data_ptr(), rows, cols, lo, hi, sp, sp + rows * 8, side.numel() - rows * 8, code, _raw_stream(cur))
    if rc:
        _lib.check(rc, "ste_backward_mask")
    return gx


def train_backward_wide(grad_output, side, rows, cols, lo, hi, out_dtype):
    """STE backward behind a fp32-result (autocast) forward: fp32 grad_output -> masked gradient in the input's dtype
    (the autograd engine's cast folded in; fq_ste_bwd_mask_wide).  side = bounds + mask of THAT forward."""
    code = _DTYPES.get(out_dtype)
    if code is None or grad_output.device.type != "cuda":
        raise TypeError(f"ste_backward[wide]: unsupported input dtype {out_dtype} / device {grad_output.device}")
    if grad_output.dtype == torch.float32 and not grad_output.is_contiguous() and grad_output.numel():
        res = _mask_backward_v([grad_output], [side], [rows], cols, lo, hi, code, True, None, out_dtype)
        if res is not None:
            return res[0]
    g = _aligned(grad_output if grad_output.dtype == torch.float32 else grad_output.float())
    gx = torch.empty(g.shape, dtype=out_dtype, device=g.device)
    sp = side.data_ptr()
    with _DeviceOf(g):
        rc = _lib.lib().fq_ste_bwd_mask_wide(g.data_ptr(), gx.data_ptr(), rows, sp, sp + rows * 8, None, None, 0, None, None,
                                             cols, float(lo), float(hi), code, _stream(g))
    _lib.check(rc, "ste_backward_mask_wide")
    return gx


def autocast_active(x):
    """True when the reference's op chain would run its fp32-promoted arithmetic on x: a 16-bit CUDA tensor inside
    torch.autocast("cuda") (`reciprocal`, i.e. the `int / Tensor` of utils_quant.py:71, is on autocast's fp32 list)."""
    return x.dtype in (torch.bfloat16, torch.float16) and x.is_cuda and torch.is_autocast_enabled("cuda")


def autocast_narrow_ok(x):
    """QuantizeLinear may ask for the autocast result "rounded once to the operand dtype" only when that dtype IS the
    autocast dtype: F.linear's autocast cast rounds the reference's fp32 result to torch.get_autocast_dtype("cuda"), so
    for an fp16 tensor inside autocast(bf16) (or the reverse) the narrow path would round twice.  Otherwise the callers
    return the fp32 result and let F.linear do the single rounding, exactly as the reference does."""
    return torch.get_autocast_dtype("cuda") == x.dtype


def sym_forward_autocast(x, num_bits, layerwise, wide, lo=-2.0, hi=2.0, train=None):
    """SymQuantizer.forward with autocast arithmetic (fq_sym_fwd_autocast).
    train: None (no side outputs) | "bounds" | "mask".  -> (y, side or bounds or None, rows, cols, got)
    where got is the side information actually produced ("mask", "bounds" or None)."""
    code = _prep(x, "sym_quantize[autocast]")
    rows, cols = rows_cols(tuple(x.shape), layerwise)
    if x.numel() == 0:
        _empty_input("sym_quantize[autocast]", x, layerwise)
        return torch.empty(x.shape, dtype=torch.float32 if wide else x.dtype, device=x.device), None, rows, cols, None
    check_4d(x, layerwise)
    L = _lib.lib()
    xv = rows_view(x, layerwise)
    if xv:   # strided rows: served in the kernel (register-kernel shapes; anything else takes the copy path below)
        mbytes = _mask_bytes(rows, cols, code)
        y, yv = _strided_out(x, torch.float32 if wide else x.dtype)
        if mbytes and yv is not False:
            side = got = None
            bp = mp = None
            if train == "mask":
                side = torch.empty(rows * 8 + mbytes, dtype=torch.uint8, device=x.device)
                bp, mp, got = side.data_ptr(), side.data_ptr() + rows * 8, "mask"
            elif train == "bounds":
                side = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
                bp, got = side.data_ptr(), "bounds"
            arr = (_lib.FwdTensorV * 1)(_lib.FwdTensorV(x.data_ptr(), y.data_ptr(), rows, int(num_bits), bp, mp, mbytes if mp else 0, _rv(xv), _rv(yv)))
            with _DeviceOf(x):
                rc = L.fq_sym_fwd_multi_v(1, arr, cols, code, _SEM_AUTOCAST, 2 if wide else 1, float(lo), float(hi), _stream(x))
            if rc != _lib.ERR_UNSUPPORTED:
                _lib.check(rc, "sym_quantize[autocast]")
                global _views_served
                _views_served += 1
                return y, side, rows, cols, got
    xc = x if x.is_contiguous() else x.contiguous()
    y = torch.empty(xc.shape, dtype=torch.float32 if wide else x.dtype, device=x.device)
    side, got = None, None
    ws_bytes = _ws_bytes(rows, cols, code)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes else None
    ws_ptr = ws.data_ptr() if ws is not None else None
    with _DeviceOf(x):
        st = _stream(x)
        rc = _lib.ERR_UNSUPPORTED
        if train == "mask" and xc is x:  # (a wide result's mask has its own layout: backward = train_backward_wide)
            mbytes = _mask_bytes(rows, cols, code)
            if mbytes:
                side = torch.empty(rows * 8 + mbytes, dtype=torch.uint8, device=x.device)
                sp = side.data_ptr()
                rc = L.fq_sym_fwd_autocast(xc.data_ptr(), y.data_ptr(), rows, cols, int(num_bits), code, _SEM_AUTOCAST, int(wide), float(lo), float(hi),
                                           sp, sp + rows * 8, mbytes, ws_ptr, ws_bytes, st)
                got = "mask"
        if rc == _lib.ERR_UNSUPPORTED:
            side, got = None, None
            bptr = None
            if train in ("mask", "bounds"):
                side = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
                bptr, got = side.data_ptr(), "bounds"
            rc = L.fq_sym_fwd_autocast(xc.data_ptr(), y.data_ptr(), rows, cols, int(num_bits), code, _SEM_AUTOCAST, int(wide), float(lo), float(hi),
                                       bptr, None, 0, ws_ptr, ws_bytes, st)
    _lib.check(rc, "sym_quantize[autocast]")
    if xc is not x:
        out = torch.empty_like(x, dtype=y.dtype)  # keeps the input's strides, as the reference's elementwise ops do
        out.copy_(y)
        y = out
    return y, side, rows, cols, got


def pair_forward(w, x, w_bits, a_bits, lo, hi, need_w, need_x, wide=False):
    """Two tensors with the same row length in ONE launch (fq_sym_fwd_pair): QuantizeLinear's weight [out, in] (per
    output channel) + input [..., in] (per token), or the attention block's K + V (per token).
    -> (wq, xq, side_w, side_x, rows_w, rows_x, cols) or None when the pair is not served (different dtypes / devices,
    misaligned, non-contiguous, rows too long): use two calls then.
    side_* (row bounds + STE mask, as in train_forward) is produced only for the operands that need a gradient.
    wide: under autocast, return the reference's fp32 results (else those rounded once to the operand dtype)."""
    if w.dtype != x.dtype or w.device != x.device or not w.is_cuda:
        return None
    wv = xv = None
    if not (w.is_contiguous() and x.is_contiguous()):   # strided rows of either operand: served in the kernel where the view allows it
        wv, xv = rows_view(w), rows_view(x)
        if wv is False or xv is False:
            return None
    code = _DTYPES.get(w.dtype)
    if code is None or not (1 <= w.dim() <= 3 and 1 <= x.dim() <= 3) or x.shape[-1] != w.shape[-1] or x.numel() == 0 or w.numel() == 0:
        return None
    cols = w.shape[-1]
    rows_w, rows_x = w.numel() // cols, x.numel() // cols
    mw, mx = _mask_bytes(rows_w, cols, code), _mask_bytes(rows_x, cols, code)
    if not mw or not mx:
        return None
    ac = autocast_active(w)
    wide = bool(wide and ac)
    if ac and not wide and not autocast_narrow_ok(w):
        return None  # autocast dtype != operand dtype: the two-call path returns fp32 results, F.linear rounds once
    L = _lib.lib()
    side_w = torch.empty(rows_w * 8 + mw, dtype=torch.uint8, device=w.device) if need_w else None
    side_x = torch.empty(rows_x * 8 + mx, dtype=torch.uint8, device=w.device) if need_x else None
    pw, px = (side_w.data_ptr() if need_w else None), (side_x.data_ptr() if need_x else None)
    if wv or xv:
        (wq, wqv), (xq, xqv) = _strided_out(w, torch.float32 if wide else None), _strided_out(x, torch.float32 if wide else None)
        if wqv is False or xqv is False:
            return None
        arr = (_lib.FwdTensorV * 2)(
            _lib.FwdTensorV(w.data_ptr(), wq.data_ptr(), rows_w, w_bits, pw, pw + rows_w * 8 if need_w else None, mw if need_w else 0, _rv(wv), _rv(wqv)),
            _lib.FwdTensorV(x.data_ptr(), xq.data_ptr(), rows_x, a_bits, px, px + rows_x * 8 if need_x else None, mx if need_x else 0, _rv(xv), _rv(xqv)))
        rc = _on_device(w, lambda st: L.fq_sym_fwd_multi_v(2, arr, cols, code, _SEM_AUTOCAST if ac else _semantics, (2 if wide else 1) if ac else 0, lo, hi, st))
        if rc:
            if rc == _lib.ERR_UNSUPPORTED:
                return None
            _lib.check(rc, "quantize_pair[strided]")
        global _views_served
        _views_served += 1
        return wq, xq, side_w, side_x, rows_w, rows_x, cols
    if wide:
        wq, xq = torch.empty(w.shape, dtype=torch.float32, device=w.device), torch.empty(x.shape, dtype=torch.float32, device=w.device)
    else:
        wq, xq = torch.empty_like(w), torch.empty_like(x)
    rc = _on_device(w, lambda st: L.fq_sym_fwd_pair(
        w.data_ptr(), wq.data_ptr(), rows_w, w_bits, pw, pw + rows_w * 8 if need_w else None, mw if need_w else 0,
        x.data_ptr(), xq.data_ptr(), rows_x, a_bits, px, px + rows_x * 8 if need_x else None, mx if need_x else 0,
        cols, code, _SEM_AUTOCAST if ac else _semantics, (2 if wide else 1) if ac else 0, lo, hi, st))
    if rc:
        if rc == _lib.ERR_UNSUPPORTED:
            return None
        _lib.check(rc, "quantize_pair")
    return wq, xq, side_w, side_x, rows_w, rows_x, cols


# ---- the module's hot path: everything that depends only on (shapes, dtype, device) is decided once per module and shape (`pair_plan`);
# the call itself -- allocate, launch, build the node -- is the C++ node's (csrc/fq_autograd_node.cpp::pair_forward).  Without the node,
# pair_forward / pair_backward above serve (rounds 4-5 had lean Python twins of them here; the C++ node replaced those).
def pair_plan(w, x):
    """-> (code, cols, rows_w, rows_x, mask bytes w, mask bytes x, device index) for a contiguous CUDA weight [out, in] and input [..., in] of
    one dtype that the two-tensor launch serves, else None"""
    if w.dtype != x.dtype or w.device != x.device or not w.is_cuda or w.dim() != 2 or not 1 <= x.dim() <= 3:
        return None
    code = _DTYPES.get(w.dtype)
    cols = w.shape[1]
    if code is None or code == _lib.DTYPE_F64 or x.shape[-1] != cols or not x.numel() or not w.numel():
        return None
    rows_w, rows_x = w.shape[0], x.numel() // cols
    mw, mx = _mask_bytes(rows_w, cols, code), _mask_bytes(rows_x, cols, code)
    if not mw or not mx:
        return None
    return code, cols, rows_w, rows_x, mw, mx, w.device.index


def weight_forward(w, w_bits, lo, hi, need):
    """A QuantizeLinear's weight ALONE (its input was fake-quantized by a sibling projection), shaped like its half of pair_forward:
    -> (wq, side or None, rows, cols), or None where the pair would not be served either (the caller then takes the ordinary node).
    Under autocast: the reference's arithmetic rounded once to the operand dtype (the narrow form), as in the pair launch."""
    if not (w.is_cuda and w.is_contiguous() and w.dim() == 2 and w.numel()):
        return None
    if autocast_active(w):
        if not autocast_narrow_ok(w):
            return None
        y, side, rows, cols, got = sym_forward_autocast(w, w_bits, False, wide=False, lo=lo, hi=hi, train="mask" if need else None)
        if need and got != "mask":
            return None
        return y, (side if need else None), rows, cols
    if need:
        return train_forward("sym", w, w_bits, False, lo, hi)
    return sym_quantize(w, w_bits), None, w.shape[0], w.shape[1]


def pair_backward(gw, gx, side_w, side_x, rows_w, rows_x, cols, lo, hi, inplace_w=False):
    """STE backward of both operands in one launch; either gradient may be None (then only the other is computed).
    inplace_w: the first tensor's gradient (a weight's) is masked where it stands and returned itself (see train_backward)."""
    if gw is None or gx is None:
        if gx is None:
            return train_backward(gw, side_w, rows_w, cols, lo, hi, inplace=inplace_w), None
        return None, train_backward(gx, side_x, rows_x, cols, lo, hi)
    code = _DTYPES.get(gw.dtype)
    if not (gw.is_contiguous() and gx.is_contiguous()):   # strided rows: masked in the kernel
        res = _mask_backward_v([gw, gx], [side_w, side_x], [rows_w, rows_x], cols, lo, hi, code, False, [inplace_w, False])
        if res is not None:
            return res[0], res[1]
    gw, gx = _aligned(gw), _aligned(gx)
    ow, ox = (gw if inplace_w else torch.empty_like(gw)), torch.empty_like(gx)
    pw, px = side_w.data_ptr(), side_x.data_ptr()
    L = _lib.lib()
    rc = _on_device(gw, lambda st: L.fq_ste_bwd_mask_pair(gw.data_ptr(), ow.data_ptr(), rows_w, pw, pw + rows_w * 8,
                                                          gx.data_ptr(), ox.data_ptr(), rows_x, px, px + rows_x * 8, cols, lo, hi, code, st))
    if rc:
        if rc == _lib.ERR_UNSUPPORTED:
            return train_backward(gw, side_w, rows_w, cols, lo, hi, inplace=inplace_w), train_backward(gx, side_x, rows_x, cols, lo, hi)
        _lib.check(rc, "quantize_linear_pair_backward")
    return ow, ox


def pair_backward_wide(gw, gx, side_w, side_x, rows_w, rows_x, cols, lo, hi, out_dtype):
    """pair_backward behind a wide (fp32-result) pair_forward: fp32 gradients in, gradients in the operands' dtype out."""
    if gw is None or gx is None:
        if gx is None:
            return train_backward_wide(gw, side_w, rows_w, cols, lo, hi, out_dtype), None
        return None, train_backward_wide(gx, side_x, rows_x, cols, lo, hi, out_dtype)
    code = _DTYPES.get(out_dtype)
    gw = _aligned(gw if gw.dtype == torch.float32 else gw.float())
    gx = _aligned(gx if gx.dtype == torch.float32 else gx.float())
    ow, ox = torch.empty(gw.shape, dtype=out_dtype, device=gw.device), torch.empty(gx.shape, dtype=out_dtype, device=gx.device)
    pw, px = side_w.data_ptr(), side_x.data_ptr()
    with _DeviceOf(gw):
        rc = _lib.lib().fq_ste_bwd_mask_wide(gw.data_ptr(), ow.data_ptr(), rows_w, pw, pw + rows_w * 8,
                                             gx.data_ptr(), ox.data_ptr(), rows_x, px, px + rows_x * 8, cols, float(lo), float(hi), code, _stream(gw))
    _lib.check(rc, "quantize_pair_backward_wide")
    return ow, ox


def multi_forward(tensors, bits, need, lo, hi):
    """2..4 tensors with the same row length, dtype and device in ONE launch (fq_sym_fwd_multi): a QuantizeLinear's weight
    and input plus the weights of the sibling projections that share that input (q/k/v, gate/up).
    tensors / bits / need: parallel lists (need[i]: record bounds + STE mask for tensor i's backward).
    -> ([y_i], [side_i or None], [rows_i], cols) or None when the group is not served."""
    n = len(tensors)
    t0 = tensors[0]
    if not 2 <= n <= _lib.MAX_TENSORS or not t0.is_cuda:
        return None
    code = _DTYPES.get(t0.dtype)
    cols = t0.shape[-1]
    if code is None:
        return None
    rows, mbytes = [], []
    for t in tensors:
        if t.dtype != t0.dtype or t.device != t0.device or not t.is_contiguous() or not 1 <= t.dim() <= 3 or t.shape[-1] != cols or t.numel() == 0:
            return None
        r = t.numel() // cols
        mb = _mask_bytes(r, cols, code)
        if not mb:
            return None
        rows.append(r)
        mbytes.append(mb)
    ac = autocast_active(t0)
    if ac and not autocast_narrow_ok(t0):
        return None
    ys = [torch.empty_like(t) for t in tensors]
    sides = [torch.empty(r * 8 + mb, dtype=torch.uint8, device=t0.device) if nd else None for r, mb, nd in zip(rows, mbytes, need)]
    arr = (_lib.FwdTensor * n)()
    for i, (t, y, sd) in enumerate(zip(tensors, ys, sides)):
        sp = sd.data_ptr() if sd is not None else None
        arr[i] = _lib.FwdTensor(t.data_ptr(), y.data_ptr(), rows[i], int(bits[i]), sp, sp + rows[i] * 8 if sp else None, mbytes[i] if sp else 0)
    with _DeviceOf(t0):
        rc = _lib.lib().fq_sym_fwd_multi(n, arr, cols, code, _SEM_AUTOCAST if ac else _semantics, 1 if ac else 0, float(lo), float(hi), _stream(t0))
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "quantize_multi")
    return ys, sides, rows, cols


def multi_backward(grads, sides, rows, cols, lo, hi, inplace=None):
    """STE backward of the tensors of a multi_forward in one launch; grads[i] may be None (that tensor is skipped).
    inplace[i]: tensor i's gradient is masked where it stands and returned itself (weights; see train_backward)."""
    live = [i for i, g in enumerate(grads) if g is not None]
    out = [None] * len(grads)
    inplace = inplace or [False] * len(grads)
    if not live:
        return out
    if len(live) == 1:
        i = live[0]
        out[i] = train_backward(grads[i], sides[i], rows[i], cols, lo, hi, inplace=inplace[i])
        return out
    code = _DTYPES.get(grads[live[0]].dtype)
    gs = {i: _aligned(grads[i]) for i in live}
    arr = (_lib.BwdTensor * len(live))()
    for j, i in enumerate(live):
        out[i] = gs[i] if inplace[i] else torch.empty_like(gs[i])
        sp = sides[i].data_ptr()
        arr[j] = _lib.BwdTensor(gs[i].data_ptr(), out[i].data_ptr(), rows[i], sp, sp + rows[i] * 8)
    g0 = gs[live[0]]
    with _DeviceOf(g0):
        rc = _lib.lib().fq_ste_bwd_mask_multi(len(live), arr, cols, float(lo), float(hi), code, 0, _stream(g0))
    if rc == _lib.ERR_UNSUPPORTED:
        for i in live:
            out[i] = train_backward(gs[i], sides[i], rows[i], cols, lo, hi, inplace=inplace[i])
        return out
    _lib.check(rc, "quantize_multi_backward")
    return out


def quantize_train(kind, x, num_bits, layerwise, lo, hi):
    """Training-mode forward (fq_*_fwd_train): -> (y, row_bounds, mask) or None if this shape/alignment is
    not served by the STE-mask path (the caller then uses the general forward + x-based backward).
    Convenience form of train_forward() with the side buffer split into its two views."""
    res = train_forward(kind, x, num_bits, layerwise, float(lo), float(hi))
    if res is None:
        return None
    y, side, rows, _ = res
    return y, side[: rows * 8].view(torch.float32).view(rows, 2), side[rows * 8:]


def ste_backward_mask(grad_output, lo, hi, row_bounds, mask, rows, cols, inplace=False):
    """STE backward from the (row_bounds, mask) a quantize_train call recorded -- x is not needed."""
    code = _prep(grad_output, "ste_backward_mask")
    g = _aligned(grad_output)  # a contiguous gradient can still be an offset view into a flat buffer
    gx = g if inplace else torch.empty_like(g)
    if g.numel() == 0:
        return gx
    L = _lib.lib()
    with _DeviceOf(g):
        rc = L.fq_ste_bwd_mask(g.data_ptr(), gx.data_ptr(), rows, cols, float(lo), float(hi), row_bounds.data_ptr(),
                               mask.data_ptr(), mask.numel(), code, _stream(g))
    _lib.check(rc, "ste_backward_mask")
    return gx


def low_bit_weight(w, scale, w_bits):
    """Elementwise part of QuantizeLinear's 1-/2-bit weight branch (utils_quant.py:203-242), forward value of
    `q.detach() - w.detach() + w`.  `scale`: [rows, 1] / [rows] per-row or 0-dim layerwise, same dtype as w."""
    code = _prep(w, "low_bit_weight")
    if w.dim() != 2:
        raise ValueError("low_bit_weight expects a 2-D weight")
    wc = w if w.is_contiguous() else w.contiguous()
    sc = scale.to(w.dtype).contiguous()
    per_row = 1 if sc.numel() == w.shape[0] and sc.numel() != 1 else 0
    if not per_row and sc.numel() != 1:
        raise ValueError(f"scale has {sc.numel()} elements for a weight with {w.shape[0]} rows")
    out = torch.empty_like(wc)
    if wc.numel():
        with _DeviceOf(w):
            rc = _lib.lib().fq_w12_fwd(wc.data_ptr(), sc.data_ptr(), out.data_ptr(), wc.shape[0], wc.shape[1], int(w_bits),
                                       per_row, code, _stream(w))
        _lib.check(rc, "low_bit_weight")
    return out


def low_bit_weight_fused(w, w_bits):
    """The 1-/2-bit branch in ONE launch, per-row mean|w| reduced inside the kernel in ATen's own summation order (fq_w12_fwd_rows):
    bit-identical to `abs().mean(dim=1)` + the elementwise chain on this device.  -> (out, scale[rows]) or None when the shape is not
    served (the caller then takes ATen's reduction + fq_w12_fwd)."""
    code = _prep(w, "low_bit_weight_fused")
    if code == _lib.DTYPE_F64 or w.dim() != 2 or not w.is_contiguous() or w.numel() == 0:
        return None
    out = torch.empty_like(w)
    scale = torch.empty(w.shape[0], dtype=w.dtype, device=w.device)
    L = _lib.lib()
    rc = _on_device(w, lambda st: L.fq_w12_fwd_rows(w.data_ptr(), out.data_ptr(), scale.data_ptr(), w.shape[0], w.shape[1], int(w_bits), code, st))
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "low_bit_weight_fused")
    return out, scale


def sym_quantize(x, num_bits, layerwise=False, want_bounds=False):
    """SymQuantizer.forward (utils_quant.py:37-74).  -> y, or (y, row_bounds) if want_bounds."""
    y, bounds, _, _ = _rowwise("sym", x, num_bits, layerwise, want_bounds, False)
    return (y, bounds) if want_bounds else y


def asym_quantize(x, num_bits, layerwise=False, want_bounds=False):
    """AsymQuantizer.forward (utils_quant.py:96-149)."""
    y, bounds, _, _ = _rowwise("asym", x, num_bits, layerwise, want_bounds, False)
    return (y, bounds) if want_bounds else y


def sym_quantize_debug(x, num_bits, layerwise=False):
    """-> (y, idx int32, s float32[rows]) -- the bin indices the parity tests compare bit-exactly."""
    y, _, idx, scale = _rowwise("sym", x, num_bits, layerwise, False, True)
    return y, idx, scale


def asym_quantize_debug(x, num_bits, layerwise=False):
    """-> (y, idx int32, {alpha, beta} float32[rows, 2])"""
    y, _, idx, scale = _rowwise("asym", x, num_bits, layerwise, False, True)
    return y, idx, scale


def ste_backward(grad_output, x, lo, hi, row_bounds=None, rows_cols_hint=None):
    """STE mask (utils_quant.py:83-87): grad where lo < x < hi (or x is NaN), else 0."""
    code = _prep(x, "ste_backward")
    if grad_output.device != x.device:
        raise RuntimeError("ste_backward: grad_output and input live on different devices")
    if grad_output.shape != x.shape:
        raise RuntimeError(f"ste_backward: shape mismatch {tuple(grad_output.shape)} vs {tuple(x.shape)}")
    if grad_output.dtype != x.dtype:  # cannot happen through the autograd Functions (output dtype == input dtype)
        raise NotImplementedError(f"ste_backward: grad dtype {grad_output.dtype} != input dtype {x.dtype}")
    L = _lib.lib()
    if not (grad_output.is_contiguous() and x.is_contiguous()) and grad_output.numel() and code != _lib.DTYPE_F64:
        gv, xv = rows_view(grad_output), rows_view(x)   # strided rows of the gradient and / or the saved input: served in the kernel
        if gv is not False and xv is not False:
            gx, ov = _strided_out(grad_output)
            if ov is not False:
                rows, cols = rows_cols(tuple(x.shape), False)
                with _DeviceOf(x):
                    rc = L.fq_ste_bwd_v(grad_output.data_ptr(), _rv(gv), x.data_ptr(), _rv(xv), gx.data_ptr(), _rv(ov), rows, cols, float(lo), float(hi),
                                        None, code, _stream(x))
                if rc != _lib.ERR_UNSUPPORTED:
                    _lib.check(rc, "ste_backward[strided]")
                    global _views_served
                    _views_served += 1
                    return gx
    g = grad_output if grad_output.is_contiguous() else grad_output.contiguous()
    xc = x if x.is_contiguous() else x.contiguous()
    gx = torch.empty_like(g)
    if g.numel() == 0:
        return gx
    with _DeviceOf(x):
        if row_bounds is not None:
            rows, cols = rows_cols_hint
            rc = L.fq_ste_bwd_rows(g.data_ptr(), xc.data_ptr(), gx.data_ptr(), rows, cols, float(lo), float(hi),
                                   row_bounds.data_ptr(), code, _stream(x))
        else:
            rc = L.fq_ste_bwd(g.data_ptr(), xc.data_ptr(), gx.data_ptr(), g.numel(), float(lo), float(hi), code, _stream(x))
    _lib.check(rc, "ste_backward")
    return gx


# ---- the integer side of the forward: packed bins + scales for export, scale pre-pass --------------------
_CONTAINERS = {"int4": _lib.BINS_INT4, "int8": _lib.BINS_INT8, "int16": _lib.BINS_INT16}


class QuantExport:
    """Result of sym_export / asym_export.
      bins      packed integer bins: int8 / int16 tensor of the input's shape, or (int4) uint8 [rows, ceil(cols/2)] with
                element 2k in the low nibble of byte k
      scales    float32 [rows, 2]: Sym {s, t2 = s + 1e-6};  Asym {a = alpha + 1e-8, beta}
      overflow  int32 [rows]: elements the container saturated (0 everywhere <=> dequantize() == the fake-quant forward)
    """
    __slots__ = ("kind", "bins", "scales", "overflow", "container", "num_bits", "shape", "rows", "cols", "dtype")

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def unpacked(self):
        """bins as an int32 tensor [rows, cols]"""
        b = self.bins
        if self.container != "int4":
            v = b.reshape(self.rows, self.cols).to(torch.int32)
            return v & 0xFFFF if (self.kind == "asym" and self.container == "int16") else v   # unsigned 16-bit bins
        lo, hi = (b & 0xF).to(torch.int32), (b >> 4).to(torch.int32)
        v = torch.stack((lo, hi), dim=-1).reshape(self.rows, -1)[:, : self.cols]
        return torch.where(v >= 8, v - 16, v) if self.kind == "sym" else v

    def dequantize(self):
        """the fake-quant forward's value, recomputed from bins + scales with the reference's op order (fp32 tensors ->
        round to the tensor dtype after every op); bit-identical to SymQuantizer / AsymQuantizer.forward on rows with
        overflow == 0 (cpu_eager semantics: true divisions)."""
        q = self.unpacked().to(torch.float32)
        sc = self.scales
        dt = self.dtype
        if self.kind == "sym":
            y = (q / sc[:, 1:2]).to(dt)
        else:
            S = torch.tensor(float(2 ** self.num_bits - 1), device=q.device)  # a tensor divisor: true division (a Python scalar
            y = (q / S).to(dt).to(torch.float32)                               # would become a multiply by 1/S on the GPU)
            y = (y * sc[:, 0:1]).to(dt).to(torch.float32)
            y = (y + sc[:, 1:2]).to(dt)
        return y.reshape(self.shape)


def default_container(kind, num_bits, dtype):
    """smallest container that is lossless for every input: Sym bins reach +-(qmax + 1) in bf16 at 8 bits (no clamp in
    the reference), so 8-bit bf16 defaults to int16; ask for "int8" explicitly to get the saturating deployment format."""
    if kind == "asym":
        return "int4" if num_bits <= 4 else "int8" if num_bits <= 8 else "int16"
    if num_bits <= 4:
        return "int4"
    if num_bits <= 7 or (num_bits == 8 and dtype != torch.bfloat16):
        return "int8"
    return "int16"


def _export(kind, x, num_bits, layerwise, container, autocast):
    what = f"{kind}_export"
    code = _prep(x, what)
    rows, cols = rows_cols(tuple(x.shape), layerwise)
    if x.numel() == 0:
        raise RuntimeError(f"{what}: empty tensor")
    container = container or default_container(kind, num_bits, x.dtype)
    cc = _CONTAINERS.get(container)
    if cc is None:
        raise ValueError(f"{what}: container must be one of {sorted(_CONTAINERS)}, got {container!r}")
    xc = x if x.is_contiguous() else x.contiguous()
    L = _lib.lib()
    nbytes = L.fq_export_bins_bytes(rows, cols, cc)
    raw = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    scales = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
    overflow = torch.empty(rows, dtype=torch.int32, device=x.device)
    with _DeviceOf(x):
        if kind == "sym":
            rc = L.fq_sym_export(xc.data_ptr(), raw.data_ptr(), scales.data_ptr(), overflow.data_ptr(), rows, cols, int(num_bits), cc, code,
                                 _SEM_AUTOCAST if autocast else _semantics, 1 if autocast else 0, _stream(x))
        else:
            rc = L.fq_asym_export(xc.data_ptr(), raw.data_ptr(), scales.data_ptr(), overflow.data_ptr(), rows, cols, int(num_bits), cc, code,
                                  _semantics, _stream(x))
    _lib.check(rc, what)
    if container == "int8":
        bins = (raw.view(torch.int8) if kind == "sym" else raw).view(xc.shape)
    elif container == "int16":
        bins = raw.view(torch.int16).view(xc.shape)   # Asym 16-bit bins 32768..65535 read as negative int16: use unpacked()
    else:
        bins = raw.view(rows, (cols + 1) // 2)
    return QuantExport(kind=kind, bins=bins, scales=scales, overflow=overflow, container=container, num_bits=int(num_bits),
                       shape=tuple(xc.shape), rows=rows, cols=cols, dtype=x.dtype)


def sym_export(x, num_bits, layerwise=False, container=None, autocast=None):
    """SymQuantizer's integer bins `torch.round(input * s)` (utils_quant.py:71-72) packed into int4 / int8 / int16 + per-row
    {s, t2}.  autocast: None = follow torch.is_autocast_enabled (as the forward does), or force True / False."""
    ac = autocast_active(x) if autocast is None else bool(autocast)
    return _export("sym", x, num_bits, layerwise, container, ac)


def asym_export(x, num_bits, layerwise=False, container=None):
    """AsymQuantizer's bins `torch.round(input_normalized * s)` (utils_quant.py:144-146), unsigned, + per-row {alpha+1e-8, beta}"""
    return _export("asym", x, num_bits, layerwise, container, False)


def sym_row_scales(x, num_bits, layerwise=False, autocast=None):
    """per-row {s, t2} of SymQuantizer.forward without the elementwise pass (fq_sym_row_scales) -> float32 [rows, 2]"""
    code = _prep(x, "sym_row_scales")
    rows, cols = rows_cols(tuple(x.shape), layerwise)
    if x.numel() == 0:
        raise RuntimeError("sym_row_scales: empty tensor")
    ac = autocast_active(x) if autocast is None else bool(autocast)
    xc = x if x.is_contiguous() else x.contiguous()
    scales = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
    with _DeviceOf(x):
        rc = _lib.lib().fq_sym_row_scales(xc.data_ptr(), scales.data_ptr(), rows, cols, int(num_bits), code, _SEM_AUTOCAST if ac else _semantics, 1 if ac else 0,
                                          -2.0, 2.0, None, None, 0, _stream(x))
    _lib.check(rc, "sym_row_scales")
    return scales
