"""Drop-in replacement for LLM-QAT's `models/utils_quant.py`.

Same three names, same signatures, same attributes, same exceptions:

    SymQuantizer.apply(input, clip_val, num_bits, layerwise)     reference :31-87
    AsymQuantizer.apply(input, clip_val, num_bits, layerwise)    reference :90-162
    QuantizeLinear(*kargs, symmetric=True, bias=False, w_bits=32, a_bits=32,
                   act_layerwise=False, weight_layerwise=False)  reference :165-254

so `models/modeling_llama_quant.py` (which imports them by name, :51) and everything above it
(`train.py`, `utils/kd_trainer.py`) run unchanged.  The eager ATen op chains are replaced by
single-pass HIP kernels for gfx950 (see INTEGRATION.md for the one-line switch).
"""
import ctypes
import logging
import os
import sys
import threading
import weakref

import torch
import torch.nn as nn

from . import _node, compiled, cpu_tensors, ops

# How the backward learns which gradients to zero (results are identical in all three modes):
#   "mask"   (default) the forward records per-row value bounds + a 1-bit/element STE mask for rows
#            that can be clipped; the backward reads g (+ mask) only -- x is neither re-read nor
#            kept alive by the autograd node.  bf16: 4 + 4.1 instead of 4 + 6 bytes/element.
#   "bounds" the forward records only the per-row bounds; the backward re-reads x for rows that
#            can be clipped (weights practically never can).
#   "plain"  nothing recorded; the backward re-reads x everywhere (the reference's data flow).
_BACKWARD_MODE = os.environ.get("LLMQAT_AMD_BACKWARD", "mask")
if os.environ.get("LLMQAT_AMD_ROW_BOUNDS", "1") == "0":
    _BACKWARD_MODE = "plain"


def set_backward_mode(mode):
    global _BACKWARD_MODE
    if mode not in ("mask", "bounds", "plain"):
        raise ValueError(mode)
    _BACKWARD_MODE = mode


def get_backward_mode():
    return _BACKWARD_MODE


def _clip_pair(clip_val):
    """(lo, hi) of a clip tensor as Python floats, read from the tensor on every call, as the reference does (:85-86).  The model's
    clip tensors live on the CPU (plain attributes, modeling_llama_quant.py:251-252): reading two floats costs ~1 us.  (Rounds 2-3
    memoised the pair per tensor object and version; a write through `.data` goes past that version counter)"""
    if clip_val is _CLIP:
        return -2.0, 2.0
    if clip_val.dim():
        lo, hi = clip_val.tolist()[:2]
        return float(lo), float(hi)
    v = float(clip_val.item())
    return v, v


# ---- counters of what the stateful host logic did (llm_qat_amd.stats()): every optimisation below that can silently fall back says so here
_stats = {}

# The C++ autograd nodes (csrc/fq_autograd_node.cpp): the straight-line cases -- a QuantizeLinear's operand pair, a sibling's weight-only launch,
# K and V at the KV-cache hooks, a bare SymQuantizer / AsymQuantizer.apply in mask mode -- without Python on the autograd engine's thread.
# Optional: without the file (or with LLMQAT_AMD_CPP_NODE=0 / cpp_node(False)) the Python nodes below serve -- same launches, same bits --
# and everything off the straight line is theirs anyway.  Bound to the kernel library and calibrated at the end of this module.
_cnode = _node.load()
_USE_CNODE = _cnode is not None


def cpp_node(flag=True):
    """use the C++ autograd nodes (default: on when _fq_node.so is built and loads); -> whether they are in use now"""
    global _USE_CNODE
    _USE_CNODE = bool(flag) and _cnode is not None and _cnode_ready
    return _USE_CNODE


def host_node():
    """-> "c++" or "python (<why>)": which autograd nodes the straight-line paths build"""
    return "c++" if _USE_CNODE else "python (%s)" % (_node.status() if _cnode is None else ("guard not calibrated" if not _cnode_ready else "switched off"))


_cnode_ready = False


def _count(name, n=1):
    _stats[name] = _stats.get(name, 0) + n


def stats(reset=False):
    """-> dict of counters since import / the last reset:
      pair_launch / single_launch        QuantizeLinear forwards served by one two-tensor launch / by separate calls
      act_share_hit / act_share_miss     sibling projections that found their input already fake-quantized / that quantized it
      kv_pair_launch / kv_pair_hit / kv_pair_discarded / kv_pair_learned_off
                                         K+V speculation at the unchanged hooks: launched, V served from it, V result thrown away,
                                         call signatures that stopped pairing after a wrong guess
      wcache_fill / wcache_hit           weight-quant cache (opt-in)
      inplace_taken / inplace_refused:<reason>
                                         weight gradients masked where they stand vs copied, by the guard's reason
                                         (uncalibrated, no_refcount_api, storage, anomaly, py_refs, cxx_refs, storage_refs, base_refs)
      share_disabled:no_region_api / kv_pair_disabled:no_region_api / kv_pair_disabled:no_functorch_api
                                         calls that ran the reference's launch structure because a private torch API the stateful logic
                                         leans on is missing in this build (fail closed; warned once)
      w12_fused_unverified               1-/2-bit weights that took ATen's own abs + mean because the one-launch kernel's restatement of
                                         ATen's summation order did not verify on this device / torch build
      cpp_pair_forward / cpp_weight_forward / cpp_pair_backward / cpp_one_backward / cpp_slow_backward
                                         what the C++ autograd nodes did (_fq_node.so; `host_node()` says whether it is loaded): operand-pair and
                                         weight-only launches made from C++, pair / one-tensor (K, V) backwards, and backwards handed back to the
                                         Python nodes' code"""
    out = dict(_stats)
    if reset:
        _stats.clear()
    if _cnode is not None:      # what the C++ node counted (its guard's decisions, its launches): same names
        for k, v in _cnode.counters(reset).items():
            out[k] = out.get(k, 0) + v
    return out


def _graph_aware(backward):
    """For every backward below (they launch kernels, which autograd cannot look into).  The reference's backward is made of
    differentiable ops (`grad_output.clone()` + two masked assignments, :83-87), so `create_graph=True` works there; rounds 1-3
    answered it with @once_differentiable's error.  The op is `keep * g` with a mask that does not depend on g, so when a graph is
    asked for -- grad mode is ON inside backward -- the kernels run once on ONES to learn what they keep, and `torch.where` applies
    that to the real gradients: the same values (a masked NaN becomes 0, as with the masked assignment), differentiable in g to any
    order, at the cost of one extra launch and two ATen ops in that rare mode only.  Without a graph the kernels run as before."""
    def wrapper(ctx, *grads):
        if not torch.is_grad_enabled():
            return backward(ctx, *grads)
        with torch.no_grad():
            keeps = backward(ctx, *[None if g is None else torch.ones_like(g) for g in grads])
        outs = list(keeps)
        for i, g in enumerate(grads):
            k = keeps[i]
            if g is None or k is None:
                continue
            outs[i] = torch.where(k != 0, g if g.dtype == k.dtype else g.to(k.dtype), torch.zeros((), dtype=k.dtype, device=k.device))
        return tuple(outs)
    return wrapper


# _no_gradient: every Function below sets ctx.set_materialize_grads(False) and answers a gradient that never arrived with None.  The K/V
# speculation (point 7) can make a tensor an input of a node whose result nobody uses; with materialised zeros the parameters behind that
# tensor would end a step with a ZERO .grad where the reference leaves None -- and an optimizer treats the two differently (weight decay,
# moments).  With None all the way, a dead branch stays dead (tests/test_gpu_random_programs.py found this).


class _Raw:
    """What one fake-quant forward produced, apart from any autograd node: the result as plain data plus everything a backward needs.
    `_FakeQuantFunction` makes one and attaches it to its own node; a shared activation (point 1 below) is one `_Raw` that every sibling
    projection wraps in a node of its OWN (`_SharedAct`), so that no two callers ever share a piece of graph."""
    __slots__ = ("out", "mode", "saved", "clip", "rows_cols", "grad_dtype")

    def __init__(self, out, mode, saved=(), clip=None, rows_cols=None, grad_dtype=None):
        self.out, self.mode, self.saved, self.clip, self.rows_cols, self.grad_dtype = out, mode, saved, clip, rows_cols, grad_dtype


def _compute(kind, input, clip_val, num_bits, layerwise, narrow, need_grad):
    """One fake-quant forward -> _Raw.  mode: "cpu" | "none" (no backward will run) | "mask" | "mask_wide" | "bounds" | "plain"."""
    if input.device.type == "cpu":   # opt-in, plain torch ops (cpu_tensors.py): never a fallback for a CUDA tensor
        if not cpu_tensors.ENABLED:
            cpu_tensors.refuse(input, f"{kind}_quantize")
        return _Raw(cpu_tensors.forward(kind, input, num_bits, layerwise), "cpu", (input, clip_val))   # reference :45 / :104
    ac = kind == "sym" and ops.autocast_active(input)
    if not need_grad:  # no backward will run (eval, frozen input): nothing to record or save
        if ac:
            return _Raw(ops.sym_forward_autocast(input, num_bits, layerwise, wide=not (narrow and ops.autocast_narrow_ok(input)))[0], "none")
        return _Raw(ops.sym_quantize(input, num_bits, layerwise) if kind == "sym" else ops.asym_quantize(input, num_bits, layerwise), "none")
    if ac:
        # The reference under torch.autocast("cuda"): fp32 arithmetic behind the reciprocal, fp32 result (or, for QuantizeLinear's own
        # operands, that result rounded once -- exactly what F.linear's autocast cast does next).  The engine casts the reference's fp32
        # gradient to the input dtype: grad_dtype does it up front.
        mode = _BACKWARD_MODE
        lo, hi = _clip_pair(clip_val)
        narrow = narrow and ops.autocast_narrow_ok(input)
        out, side, rows, cols, got = ops.sym_forward_autocast(input, num_bits, layerwise, wide=not narrow, lo=lo, hi=hi,
                                                              train=None if mode == "plain" else mode)
        if got == "mask":   # bounds + mask (6 % of the tensor) as a SAVED tensor: hooks / checkpointing see it; a fp32 result's mask has its own layout
            return _Raw(out, "mask" if narrow else "mask_wide", (side,), (lo, hi), (rows, cols), input.dtype)
        if got == "bounds":
            return _Raw(out, "bounds", (input, clip_val, side), (lo, hi), (rows, cols), input.dtype)
        return _Raw(out, "plain", (input, clip_val), (lo, hi), (rows, cols), input.dtype)
    mode = "plain" if input.dtype == torch.float64 else _BACKWARD_MODE   # float64: the reference's data flow (saved input)
    if mode == "mask":
        lo, hi = _clip_pair(clip_val)
        res = ops.train_forward(kind, input, num_bits, layerwise, lo, hi)
        if res is not None:
            out, side, rows, cols = res
            # The side buffer (row bounds + STE bit mask) is a SAVED tensor, not a ctx attribute: saved-tensor hooks (save_on_cpu,
            # non-reentrant checkpointing's discard) see it and can drop / offload it like any other activation.  The input itself is
            # not needed again and is not saved.
            return _Raw(out, "mask", (side,), (lo, hi), (rows, cols))
        mode = "bounds"
    fn = ops.sym_quantize if kind == "sym" else ops.asym_quantize
    if mode == "bounds":
        out, row_bounds = fn(input, num_bits, layerwise, want_bounds=True)
        return _Raw(out, "bounds", (input, clip_val, row_bounds), None, ops.rows_cols(tuple(input.shape), layerwise))
    return _Raw(fn(input, num_bits, layerwise), "plain", (input, clip_val))  # reference :45 / :104 -- the unclipped input itself


def _attach(ctx, raw, inplace_grad=False):
    ctx.fq_mode, ctx.clip, ctx.rows_cols, ctx.grad_dtype = raw.mode, raw.clip, raw.rows_cols, raw.grad_dtype
    ctx.fq_inplace = inplace_grad  # a QuantizeLinear's own weight: its gradient may be masked where it stands (point 6 below)
    ctx.fq_st = _state().ref       # whose forward pass this node belongs to (_backward_started)
    ctx.set_materialize_grads(False)     # a gradient that never arrives stays None (see _no_gradient)
    if raw.saved:
        ctx.save_for_backward(*raw.saved)


class _FakeQuantFunction(torch.autograd.Function):
    @staticmethod
    def _fwd(kind, ctx, input, clip_val, num_bits, layerwise, narrow=False, inplace_grad=False):
        if type(num_bits) is not int:
            num_bits = ops.bits_arg(num_bits)   # 8.0 -> 8; 7.5 or a tensor: refused (ops.bits_arg says why)
        if torch.compiler.is_compiling():  # Dynamo traces forward/backward of the Function: same kernels as custom ops (compiled.py)
            ctx.save_for_backward(input, clip_val)
            ctx.fq_mode = "compiled"
            return compiled.fake_quant(kind, input, clip_val, num_bits, layerwise, narrow)
        raw = _compute(kind, input, clip_val, num_bits, layerwise, narrow, ctx.needs_input_grad[0])
        _attach(ctx, raw, inplace_grad)
        return raw.out

    @staticmethod
    @_graph_aware
    def backward(ctx, grad_output):
        if ctx.fq_mode == "compiled":
            input, clip_val = ctx.saved_tensors
            return compiled.fake_quant_bwd(grad_output, input, clip_val), None, None, None
        if ctx.fq_mode == "none" or grad_output is None:   # nothing to mask: the input needs no gradient (the engine is here for a clip_val
            return None, None, None, None                  # that requires grad, which gets None, :87), or no gradient arrived (_no_gradient)
        if ctx.fq_mode == "cpu":
            _backward_started(ctx.fq_st)
            input, clip_val = ctx.saved_tensors
            return cpu_tensors.backward(grad_output, input, clip_val), None, None, None
        inplace = ctx.fq_inplace and _INPLACE_WGRAD and _inplace_ok(grad_output)  # (before anything else takes a reference)
        _backward_started(ctx.fq_st)
        if ctx.fq_mode == "mask_wide":  # fp32 gradient of the fp32 result -> masked gradient in the input dtype, one pass
            lo, hi = ctx.clip
            rows, cols = ctx.rows_cols
            (side,) = ctx.saved_tensors
            return ops.train_backward_wide(grad_output, side, rows, cols, lo, hi, ctx.grad_dtype), None, None, None
        if ctx.grad_dtype is not None and grad_output.dtype != ctx.grad_dtype:
            grad_output = grad_output.to(ctx.grad_dtype)  # autocast: fp32 grad of the fp32 output; zeroing commutes with the cast
            inplace = ctx.fq_inplace and _INPLACE_WGRAD   # a fresh tensor of our own
        if ctx.fq_mode == "mask":
            lo, hi = ctx.clip
            rows, cols = ctx.rows_cols
            (side,) = ctx.saved_tensors
            return ops.train_backward(grad_output, side, rows, cols, lo, hi, inplace=inplace), None, None, None
        saved = ctx.saved_tensors  # reference :83 / :158: (input, clip_val) [+ the row bounds in "bounds" mode]
        input, clip_val = saved[0], saved[1]
        lo, hi = _clip_pair(clip_val)
        bounds = saved[2] if len(saved) > 2 else None
        if bounds is not None and not (input.is_contiguous() and grad_output.is_contiguous()):
            bounds = None
        grad_input = ops.ste_backward(grad_output, input, lo, hi, row_bounds=bounds, rows_cols_hint=ctx.rows_cols)
        return grad_input, None, None, None


def _apply_over_raw(kind, input, clip_val, num_bits, layerwise):
    """SymQuantizer.apply / AsymQuantizer.apply for the ordinary training call -- a CUDA input that needs a gradient, a clip that does not,
    mask backward -- with the C++ one-tensor node behind the result (no Python on the engine's thread in its backward).  Same launch, same
    saved side buffer, same values as `_FakeQuantFunction`; anything else returns None and the Function itself serves."""
    if not (isinstance(input, torch.Tensor) and input.is_cuda and input.requires_grad and torch.is_grad_enabled() and _BACKWARD_MODE == "mask"
            and isinstance(clip_val, torch.Tensor) and not clip_val.requires_grad and input.dtype is not torch.float64)\
            or torch.compiler.is_compiling() or _functorch_active is None or _functorch_active():
        return None
    if type(num_bits) is not int:
        num_bits = ops.bits_arg(num_bits)
    raw = _compute(kind, input, clip_val, num_bits, layerwise, False, True)
    if raw.mode in ("mask", "mask_wide"):
        rows, cols = raw.rows_cols
        return _cnode.one_node(input, raw.out, raw.saved[0], rows, cols, raw.clip[0], raw.clip[1], _state().cell)
    return _SharedAct.apply(input, raw, True)     # (bounds / plain data flows: the Python node over the same result)


class SymQuantizer(_FakeQuantFunction):
    """uniform symmetric (absmax) fake quantization; straight-through gradient masked to clip_val"""

    @staticmethod
    def forward(ctx, input, clip_val, num_bits, layerwise):
        return _FakeQuantFunction._fwd("sym", ctx, input, clip_val, num_bits, layerwise)

    @classmethod
    def apply(cls, input, clip_val, num_bits, layerwise):
        # the unchanged KV-cache hooks (two consecutive apply calls on k_proj's and v_proj's outputs): one launch, see point 7
        if _PAIR_KV and cls is SymQuantizer and not layerwise and type(num_bits) is int:
            out = _kv_hook(input, clip_val, num_bits)
            if out is not None:
                return out
        if _USE_CNODE and cls is SymQuantizer:
            out = _apply_over_raw("sym", input, clip_val, num_bits, layerwise)
            if out is not None:
                return out
        return super().apply(input, clip_val, num_bits, layerwise)


class _SymQuantizerOperand(_FakeQuantFunction):
    """SymQuantizer for QuantizeLinear's own operands: identical, except that under autocast it hands F.linear the
    fp32 result already rounded to the operand dtype (the value F.linear's autocast cast would produce from it)."""

    @staticmethod
    def forward(ctx, input, clip_val, num_bits, layerwise):
        return _FakeQuantFunction._fwd("sym", ctx, input, clip_val, num_bits, layerwise, narrow=True)


class _SymQuantizerWeight(_FakeQuantFunction):
    """_SymQuantizerOperand for the WEIGHT of a QuantizeLinear: additionally, its gradient -- F.linear's fresh wgrad, which
    no one else holds -- is masked where it stands and handed on by reference (no copy for rows that cannot clip)."""

    @staticmethod
    def forward(ctx, input, clip_val, num_bits, layerwise):
        return _FakeQuantFunction._fwd("sym", ctx, input, clip_val, num_bits, layerwise, narrow=True, inplace_grad=True)


class AsymQuantizer(_FakeQuantFunction):
    """min-max (affine) fake quantization; same straight-through gradient"""

    @staticmethod
    def forward(ctx, input, clip_val, num_bits, layerwise):
        return _FakeQuantFunction._fwd("asym", ctx, input, clip_val, num_bits, layerwise)

    @classmethod
    def apply(cls, input, clip_val, num_bits, layerwise):
        if _USE_CNODE and cls is AsymQuantizer:
            out = _apply_over_raw("asym", input, clip_val, num_bits, layerwise)
            if out is not None:
                return out
        return super().apply(input, clip_val, num_bits, layerwise)


class _LowBitWeightCpu(torch.autograd.Function):
    """the same branch for CPU tensors (opt-in, cpu_tensors.py); identity gradient"""

    @staticmethod
    def forward(ctx, w, w_bits, layerwise):
        ctx.set_materialize_grads(False)
        return cpu_tensors.low_bit_weight(w, w_bits, layerwise)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output, None, None


class _LowBitWeight(torch.autograd.Function):
    """forward value of `quan_weights_no_grad.detach() - real_weights.detach() + real_weights` (:240-242);
    its gradient w.r.t. real_weights is the identity."""

    @staticmethod
    def forward(ctx, w, scale, w_bits):
        ctx.set_materialize_grads(False)
        return ops.low_bit_weight(w, scale, w_bits)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output, None, None


class _LowBitWeightFused(torch.autograd.Function):
    """the same value from the one-launch kernel (row mean in ATen's order); identity gradient"""

    @staticmethod
    def forward(ctx, w, w_bits):
        ctx.set_materialize_grads(False)
        res = ops.low_bit_weight_fused(w, w_bits)
        if res is None:
            raise _NotServed()
        return res[0]

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output, None


class _NotServed(Exception):
    pass


# 1-/2-bit weights in ONE launch with the row mean reduced in-kernel IN ATen'S OWN SUMMATION ORDER (fq_w12_fwd_rows): bit-identical to
# `abs().mean(dim=1)` + the elementwise chain on this device, so it is the default since round 4 (LLMQAT_AMD_W12_FUSED=0 /
# fuse_low_bit_mean(False) go back to ATen's abs + mean followed by fq_w12_fwd: three launches).  Shapes the kernel does not serve
# (weight_layerwise, rows < 8, cols < 256 or not a multiple of 4, ...) take those three launches anyway.
_W12_FUSED = os.environ.get("LLMQAT_AMD_W12_FUSED", "1") != "0"


def fuse_low_bit_mean(flag=True):
    global _W12_FUSED
    _W12_FUSED = bool(flag)


# The one-launch kernel is bit-identical to the reference only as long as its restatement of ATen's reduction tree (torch's Reduce.cuh:
# block shape, vectorisation, shuffle order, the num_mp-dependent output split) IS what this torch build does on this device -- a
# CPX-partitioned MI355X (32 CUs) or another torch version may configure the reduction differently.  So it verifies itself: the first
# 1-/2-bit weight of each (device, dtype) runs a probe covering both row configurations (a wave per row: cols <= 8128; eight waves per
# row above) against live `abs().mean(dim=1)`; any differing scale switches the fused path off for that (device, dtype) -- ATen's own
# reductions then run in front of fq_w12_fwd, three launches, still bit-identical -- and stats() counts `w12_fused_unverified`.
_w12_verdict = {}


def _w12_fused_verified(w):
    key = (w.device.index, w.dtype)
    ok = _w12_verdict.get(key)
    if ok is None:
        if torch.cuda.is_current_stream_capturing():
            return False      # a probe needs a host read-back: not during graph capture (the three-launch path is capturable)
        ok = True
        with torch.no_grad():
            g = torch.Generator(device=w.device).manual_seed(1234)
            for rows, cols in ((16, 512), (9, 8128), (8, 8132), (8, 11008)):
                p = (torch.randn(rows, cols, generator=g, device=w.device) * 0.02).to(w.dtype)
                res = ops.low_bit_weight_fused(p, 1)
                if res is not None and not torch.equal(res[1], p.abs().mean(dim=1)):
                    ok = False
                    break
        _w12_verdict[key] = ok
    if not ok:
        _count("w12_fused_unverified")
    return ok


# 6. A weight's rows practically never reach the STE clip, so its backward is the identity -- and the reference's
#    `grad_output.clone()` (:84) exists only to be mutated by the two masked writes (:85-86).  For the WEIGHT operand of a
#    QuantizeLinear the gradient that arrives is F.linear's freshly computed wgrad (the quantized weight has exactly one
#    consumer, inside this module), owned by nobody else: it is masked where it stands (fq_ste_bwd_mask with gx == g; rows
#    that cannot clip are not touched at all) and handed on by reference -- no 4 B/element copy per weight per step.
#    Gradients of anything a caller can see (SymQuantizer.apply results, activations) are never touched in place.
_INPLACE_WGRAD = os.environ.get("LLMQAT_AMD_INPLACE_WEIGHT_GRAD", "1") != "0"


def inplace_weight_grad(flag=True):
    global _INPLACE_WGRAD
    _INPLACE_WGRAD = bool(flag)
    if _cnode is not None:
        _cnode.set_inplace(_INPLACE_WGRAD)


#    This deviates from PyTorch's rule for Function.backward ("never modify grad_outputs in place"), so it is GUARDED at run
#    time: the gradient is masked where it stands only if this node provably holds the only handle on it --
#      * it covers its whole storage (contiguous, offset 0, storage size == tensor size): never a slice of somebody's buffer;
#      * no other Python object refers to it (sys.getrefcount against a calibrated baseline: a tensor hook that stashes the
#        gradient, or one that substitutes its own tensor, raises the count) and no other C++ holder does (Tensor._use_count();
#        if it is a view -- F.linear's wgrad arrives as a view of a temporary -- the base has no other holder either);
#      * anomaly mode is off (it keeps gradients for its diagnostics).
#    Anything else takes the copying launch: same values, the reference's data flow (tests/test_gpu_features.py
#    ::test_inplace_weight_gradient_is_guarded).  What a hook is handed is the gradient BEFORE this node's mask, as with the
#    reference's clone -- hooks run before the node.
_ref_base = {}
_storage_use_count = getattr(torch._C, "_storage_Use_Count", None)   # holders of the StorageImpl itself (aliases made without view tracking)
_log = logging.getLogger("llm_qat_amd")


def _storage_holders(g):
    return _storage_use_count(g.untyped_storage()._cdata)


def _grad_counts(g):
    return sys.getrefcount(g), g._use_count()


def _base_counts(g):
    b = g._base
    return sys.getrefcount(b), b._use_count()


def _calibrate_grad_counts():
    """Reference counts of a gradient as it arrives in a backward written like the ones below (backward(ctx, gw, gx)), with nobody
    else holding it.  Run once at import on tiny CPU tensors, through the same decorators, F.linear producing the gradient as in
    QuantizeLinear.  Plus the holders of a StorageImpl for a tensor that owns it alone and for a view of a temporary (F.linear's
    wgrad arrives as the latter): an alias made WITHOUT view tracking -- `g.data`, `set_`, aten.alias under a key exclusion -- has
    its own TensorImpl and no `_base`, so only the storage's own count shows it."""
    class _Named(torch.autograd.Function):
        @staticmethod
        def forward(ctx, w, x):
            return w * 1.0, x * 1.0

        @staticmethod
        @_graph_aware
        def backward(ctx, gw, gx):
            _ref_base["named"] = _grad_counts(gw)
            return gw, gx

    _ref_base["base"] = _base_counts(torch.zeros(2, 4).view(4, 2))   # a view of a temporary nobody else holds
    _ref_base["storage_plain"] = _storage_holders(torch.zeros(8))
    _ref_base["storage_view"] = _storage_holders(torch.zeros(2, 4).view(4, 2))
    with torch.inference_mode(False), torch.enable_grad():
        w = torch.zeros(2, 4, requires_grad=True)
        x = torch.zeros(3, 4, requires_grad=True)
        wq, xq = _Named.apply(w, x)
        nn.functional.linear(xq, wq).sum().backward()


def _owns_storage(g):
    """the tensor IS its storage: contiguous, offset 0, nothing before or after it in the allocation"""
    return g.is_contiguous() and g.storage_offset() == 0 and g.untyped_storage().nbytes() == g.numel() * g.element_size()


def _refuse(reason):
    _count("inplace_refused:" + reason)
    return False


def _inplace_ok(g):
    """may this backward mask grad_output `g` where it stands?  Called directly from the backward with the argument itself
    (`_inplace_ok(gw)`) so that the reference count compares with the calibration.  Every refusal is counted by reason (stats())."""
    base = _ref_base.get("named")
    if base is None:
        return _refuse("uncalibrated")
    if _storage_use_count is None or not hasattr(g, "_use_count"):   # the private counters the guard reads: without them, never in place
        return _refuse("no_refcount_api")
    if sys.getrefcount(g) > base[0]:
        return _refuse("py_refs")
    if g._use_count() > base[1]:
        return _refuse("cxx_refs")
    if not _owns_storage(g):
        return _refuse("storage")
    if torch.is_anomaly_enabled():
        return _refuse("anomaly")
    if g._base is None:
        if _storage_holders(g) > _ref_base["storage_plain"]:
            return _refuse("storage_refs")
    else:
        bc, bb = _base_counts(g), _ref_base["base"]  # F.linear's wgrad arrives as a view of a temporary: fine if nobody else can reach it
        if bc[0] > bb[0] or bc[1] > bb[1]:
            return _refuse("base_refs")
        if _storage_holders(g) > _ref_base["storage_view"]:
            return _refuse("storage_refs")
    _count("inplace_taken")
    return True


_CLIP = torch.tensor([-2.0, 2.0])  # the literal the reference rebuilds on every call (:198, :245)

# ---------------------------------------------------------------------------------------------
# Call-site redundancy the reference has, removed without changing any result:
#
# 1. q_proj / k_proj / v_proj (modeling_llama_quant.py:313,317,318) and gate_proj / up_proj (:235)
#    fake-quantize the SAME activation tensor with the same bits.  `_shared_activation` remembers the
#    last fake-quantized activation per thread AS PLAIN DATA (a `_Raw`: the values + the side buffer a
#    backward needs) and hands it to the sibling projections: the kernel runs once, and since round 5
#    EVERY SIBLING GETS AN AUTOGRAD NODE OF ITS OWN over it (`_SharedAct`, or the module's `_PairNode`
#    together with its weight) -- forward launches nothing, backward is that sibling's own STE launch.
#    That is the reference's graph exactly (one node per module, :246): the input's gradient is the sum
#    of the siblings' MASKED gradients in the engine's own order, bit for bit in every program, each
#    sibling's graph can be run and freed on its own (`a.sum().backward(); b.sum().backward()`), and
#    nothing is shared but read-only data.  (Rounds 1-4 handed the siblings one node: one backward launch
#    less per group, but a second backward over it raised, and a consumer of the same input created AFTER
#    the siblings saw that input's gradient summed in another association order -- ADVICE / VERDICT r04.)
#    A hit requires the very same tensor object at the same version and address, the same grad mode /
#    autocast state / backward mode / stream / saved-tensor-hooks region; what is remembered is let go of
#    when the next backward starts, or replaced by the next module that quantizes an activation the same way.
#    Limit, stated: a write THROUGH `x.data` (or `x.set_`-ing the same storage back) bumps no version counter
#    and moves no address, so a sibling called after it is served the activation as quantized before the
#    write -- the reference model never does this; `conservative(True)` / `share_activation_quant(False)`
#    is the switch (tests/test_gpu_share_sequences.py::test_write_through_data_is_a_stated_limit).
# 2. Under activation checkpointing every weight is fake-quantized twice per step (first forward, then
#    the recompute -- reentrant or not) although it has not changed.  With the weight cache on (opt-in:
#    it keeps one quantized copy per layer alive from the first use until the second), the second use
#    within a step reuses the first result.  The key holds the parameter's identity, version counter and
#    storage address, so an optimizer step / load_state_dict / .data swap can never be served stale.
# ---------------------------------------------------------------------------------------------
_SHARE_ACT = os.environ.get("LLMQAT_AMD_SHARE_ACT", "1") != "0"
_WEIGHT_CACHE = os.environ.get("LLMQAT_AMD_WEIGHT_CACHE", "0") in ("1", "persistent")
_WEIGHT_CACHE_PERSISTENT = os.environ.get("LLMQAT_AMD_WEIGHT_CACHE", "0") == "persistent"


def share_activation_quant(flag=True):
    global _SHARE_ACT
    _SHARE_ACT = bool(flag)


def enable_weight_quant_cache(flag=True, persistent=False):
    """flag: reuse a weight's fake-quantized value for its second use within a step (checkpoint recompute).
    persistent: keep it until the weight itself changes (optimizer step), so gradient-accumulation micro-batches
    reuse it too -- costs one quantized copy per layer for the whole run."""
    global _WEIGHT_CACHE, _WEIGHT_CACHE_PERSISTENT
    _WEIGHT_CACHE = bool(flag)
    _WEIGHT_CACHE_PERSISTENT = bool(flag) and bool(persistent)


_MODE_CODE = {"mask": 0, "bounds": 1, "plain": 2}


class _ThreadState:
    """Everything the host logic remembers between calls, per FORWARD thread (thread-local: it dies with its thread, so a short-lived
    evaluation or DataParallel replica thread leaves nothing behind).  A backward runs on the autograd engine's threads: its nodes carry
    a weak reference to the state of the thread whose forward built them (`ctx.fq_st`), which is how `_backward_started` reaches it."""
    __slots__ = ("acts", "outs", "_kv", "epoch", "ref", "cell", "cepoch", "cpending", "cseen", "__weakref__")

    def __init__(self):
        self.acts = {}        # key -> (weakref(input), its version, its address, _Raw, the result's version, needs grad, region, stream)
        self.outs = []        # the last few QuantizeLinear outputs of this thread, in order (weakly): what the KV hooks pair
        self._kv = None       # the pending half of a K + V launch (`kv`: the property below tells the C++ nodes when one is pending)
        self.epoch = 0        # fake-quant backward passes started on graphs this thread built
        self.ref = weakref.ref(self)
        # the C++ node cannot touch this object from the engine's thread: its backward bumps a counter cell instead, and `_state()` --
        # the first thing every forward call does -- notices and lets go then (the same forgetting, at the thread's next look-up)
        self.cell = _cnode.epoch_new() if _cnode is not None else 0
        self.cepoch = ctypes.c_int64.from_address(self.cell) if self.cell else None
        self.cpending = ctypes.c_int64.from_address(self.cell + 8) if self.cell else None
        self.cseen = self.cepoch.value if self.cell else 0
        if self.cell:
            _cell_states[self.cell] = self.ref

    @property
    def kv(self):
        return self._kv

    @kv.setter
    def kv(self, value):
        # a pending V result pins a tensor + its side buffer: the one thing a C++ node's backward takes the GIL for (csrc/fq_autograd_node.cpp)
        self._kv = value
        if self.cpending is not None:
            self.cpending.value = 0 if value is None else 1

    def __del__(self):
        try:
            if self.cell:
                _cell_states.pop(self.cell, None)
            if self.cell and _cnode is not None:
                _cnode.epoch_free(self.cell)   # (a node that outlives its thread bumps a recycled cell: a spurious forgetting, never a stale hit)
        except Exception:  # noqa: BLE001 -- interpreter shutdown: the extension may be gone before the last thread state
            pass


_tls = threading.local()
_cell_states = {}      # epoch cell -> weakref(_ThreadState): how a C++ node's backward reaches the forward thread's state (_forget_from_cpp)


def _forget_from_cpp(cell):
    """called by a C++ node's backward, under the GIL, when its forward thread has a V result pending that nobody asked for"""
    ref = _cell_states.get(cell)
    st = ref() if ref is not None else None
    if st is not None:
        st.cseen = st.cepoch.value
        _forget(st)


def _state():
    st = getattr(_tls, "st", None)
    if st is None:
        st = _tls.st = _ThreadState()
    elif st.cepoch is not None and st.cepoch.value != st.cseen:
        st.cseen = st.cepoch.value
        _forget(st)
    return st


def _forget(st):
    st.epoch += 1
    if st.acts:
        st.acts = {}    # (rebinding, not clearing: the forward thread may be reading the old dict)
    if st.kv is not None:
        _kv_discard(st)


def _backward_started(ref):
    """every fake-quant backward calls this first, with the (weak) state of the thread whose forward built its node: what THAT thread
    remembered before (shared activations, a pending V of the K/V hooks) is let go of now -- side buffers included -- and a K/V guess made
    before this point is never honoured afterwards.  Per forward thread, not global: another thread's forward pass (a DataParallel
    replica, an evaluation thread) keeps what it remembered."""
    st = ref() if ref is not None else None
    if st is None:      # the forward thread is gone (or the C++ node has told its cell already)
        return
    _forget(st)


# The saved-tensor-hooks region API is private (torch._C._autograd._top_saved_tensors_default_hooks).  Without it the host logic cannot
# tell a checkpointed region from its surroundings, and remembering anything across calls could hand a non-reentrant checkpoint's first
# pass something its recompute cannot repeat (CheckpointError): so WITHOUT IT NOTHING IS REMEMBERED -- activation sharing and the K/V
# pairing switch themselves off (fail closed), stats() counts `share_disabled:no_region_api`, and a warning says so once.
_top_hooks = getattr(torch._C._autograd, "_top_saved_tensors_default_hooks", None)
_functorch_active = getattr(torch._C, "_are_functorch_transforms_active", None)
_warned = set()


def _disabled(what, why):
    _count(f"{what}_disabled:{why}")
    if (what, why) not in _warned:
        _warned.add((what, why))
        _log.warning("llm_qat_amd: %s is switched off: %s is not available in this torch build (same results, the reference's launch structure; "
                     "stats() counts it under %s_disabled:%s)", what, why, what, why)
    return False


def _memory_ok(what):
    """may the host logic remember something across calls?  (fail closed on a missing private API)"""
    return True if _top_hooks is not None else _disabled(what, "no_region_api")


def _region():
    """Which saved-tensor-hooks context this call runs in (its pack hook: a fresh object per context; None outside any).  Non-reentrant
    activation checkpointing is such a context -- one for the first pass, another for the recompute -- and requires both passes to save
    the same tensors: something remembered OUTSIDE a checkpointed region must not be used INSIDE it (the recompute, which starts from an
    empty memory, could not repeat that), so what is remembered is only handed out within the region it was made in.  A checkpoint
    around a whole decoder layer -- the reference's, modeling_llama_quant.py:732-747 -- contains all siblings and loses nothing.
    Callers have checked `_memory_ok()` first."""
    h = _top_hooks(True)
    return None if h is None else h[0]


def _state_word(x):
    """everything ambient that decides which arithmetic / data flow a call takes, folded into one int (part of the cache keys)"""
    return (_MODE_CODE[_BACKWARD_MODE] + 4 * ops._semantics + (8 if torch.is_grad_enabled() else 0)
            + (16 if ops.autocast_active(x) else 0))


def _act_lookup(st, key, x, region, stream):
    ent = st.acts.get(key)
    if ent is not None:
        rin, ver_in, addr, raw, ver_out, needs_grad, reg, strm = ent
        # (requires_grad can be switched on a leaf between two sibling calls without touching its version counter: a result recorded
        # without side buffers must not be handed to a call that needs a backward, nor the reverse)
        if (rin() is x and ver_in == x._version and addr == x.data_ptr() and ver_out == raw.out._version and needs_grad == x.requires_grad
                and reg is region and strm == stream):
            return raw
    return None


def _act_store(st, key, x, raw, region, stream):
    """One entry per key: the next module that quantizes an activation with the same settings -- the next layer's q_proj / gate_proj --
    replaces it, and the first fake-quant backward that starts drops them all (`_backward_started`), so at most one fake-quantized
    activation per key outlives its siblings, and none outlives the forward pass.  A dead input can never match (`rin() is x` on a
    dead weak reference is False), so id reuse is harmless."""
    st.acts[key] = (weakref.ref(x), x._version, x.data_ptr(), raw, raw.out._version, x.requires_grad, region, stream)


class _SharedAct(torch.autograd.Function):
    """One sibling's OWN autograd node over an activation some launch has already fake-quantized (a `_Raw`): forward launches nothing,
    backward is the ordinary STE backward of `_FakeQuantFunction` over the raw's side buffer."""

    @staticmethod
    def forward(ctx, x, raw, own=False):
        _attach(ctx, raw)
        if own:               # the raw is this call's alone (_apply_over_raw): its tensor becomes the output as it is
            return raw.out
        return raw.out.view_as(raw.out)   # a tensor of this node's own: the raw stays plain data for the next sibling

    backward = _FakeQuantFunction.backward


def _wrap_shared(x, raw):
    if raw.mode == "none" or not (torch.is_grad_enabled() and x.requires_grad):
        return raw.out
    out = _SharedAct.apply(x, raw)
    return out


def _shared_activation(quantizer, x, num_bits, layerwise):
    if quantizer is SymQuantizer:
        quantizer = _SymQuantizerOperand
    if not _SHARE_ACT or x.is_inference() or torch.compiler.is_compiling() or not _memory_ok("share"):
        return quantizer.apply(x, _CLIP, num_bits, layerwise)   # (inference tensors have no version counter: nothing is remembered about them)
    st = _state()
    key = (quantizer, num_bits, layerwise, _state_word(x))
    region, stream = _region(), (ops._stream(x) if x.is_cuda else 0)
    raw = _act_lookup(st, key, x, region, stream)
    if raw is None:
        _count("act_share_miss")
        if type(num_bits) is not int:
            num_bits = ops.bits_arg(num_bits)
        raw = _compute("asym" if quantizer is AsymQuantizer else "sym", x, _CLIP, num_bits, layerwise, quantizer is _SymQuantizerOperand,
                       torch.is_grad_enabled() and x.requires_grad)
        _act_store(st, key, x, raw, region, stream)
    else:
        _count("act_share_hit")
    return _wrap_shared(x, raw)


# 3. QuantizeLinear needs its weight [out, in] and its input [tokens, in] fake-quantized at the same moment, and both
#    reduce over `in` (same row length = same launch shape): one two-tensor launch forward, one backward (F.linear's
#    backward produces both gradients together), instead of two each -- a launch carries ~2.8 us of fixed cost.
#    A sibling that finds its input already fake-quantized launches its weight alone and still takes ONE backward launch for
#    both of its gradients (its own `_PairNode` over its weight's result and the shared activation data).
_PAIR = os.environ.get("LLMQAT_AMD_PAIR_OPERANDS", "1") != "0"


def pair_operands(flag=True):
    global _PAIR
    _PAIR = bool(flag)


class _PairNode(torch.autograd.Function):
    """Autograd node over a QuantizeLinear's two operands: the results of one two-tensor launch, or a weight's own launch + an
    activation a sibling has already fake-quantized.  The two results have one consumer, that module's F.linear, and so live or die
    together.  (K and V of the KV hooks share a forward launch but never a node.)  The clip is the module's own literal [-2, 2]."""

    @staticmethod
    def forward(ctx, weight, input, res, code, view_x):
        wq, xq, side_w, side_x, rows_w, rows_x, cols = res
        ctx.fq = (rows_w, rows_x, cols, code, weight.dtype, _state().ref)
        ctx.fq_foreign = False
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(side_w, side_x)  # saved tensors (either may be None): visible to saved-tensor hooks
        # (wq / xq are fresh tensors of the launch that nothing else refers to and become this node's outputs as they are; an xq that a
        # sibling made is plain data, or the output of that sibling's node: this node gets a tensor of its own over the same memory)
        if view_x:
            xq = xq.view_as(xq)
        # An operand that needs no gradient (frozen weight, input without grad) gets a result that needs none either, as
        # SymQuantizer.apply gives in the reference: F.linear's backward then skips the wgrad / dgrad GEMM it would
        # otherwise run only for this node to throw the result away.
        nw, nx = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (nw and nx):
            ctx.mark_non_differentiable(*[t for t, need in ((wq, nw), (xq, nx)) if not need])
        return wq, xq

    @staticmethod
    @_graph_aware
    def backward(ctx, gw, gx):
        # (runs on the autograd engine's device thread, where every line of Python costs 2-3x what it costs on the caller's --
        # tools/host_pieces.py --: the straight-line case is the C++ node's, this is the general form and the node's way out)
        inplace_w = _INPLACE_WGRAD and gw is not None and not ctx.fq_foreign and _inplace_ok(gw)
        rows_w, rows_x, cols, code, dtype, st = ctx.fq
        _backward_started(st)
        side_w, side_x = ctx.saved_tensors
        need_w, need_x = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if need_w and gw is not None and gw.dtype != dtype:
            gw, inplace_w = gw.to(dtype), _INPLACE_WGRAD   # a fresh tensor of our own
        elif not need_w:
            gw = None
        gx = gx.to(dtype) if (need_x and gx is not None and gx.dtype != dtype) else (gx if need_x else None)
        if gw is None and gx is None:
            return None, None, None, None, None
        ow, ox = ops.pair_backward(gw, gx, side_w, side_x, rows_w, rows_x, cols, -2.0, 2.0, inplace_w=inplace_w)
        return ow, ox, None, None, None


class _CppCtx:
    """what _PairNode.backward reads from its ctx, for a backward the C++ node hands back (see _pair_backward_from_cpp)"""
    __slots__ = ("fq", "fq_foreign", "saved_tensors", "needs_input_grad")


_CODE_DTYPE = {v: k for k, v in ops._DTYPES.items()}


def _pair_backward_from_cpp(gw, gx, side_w, side_x, rows_w, rows_x, cols, code, need_w, need_x):
    """The C++ node's way out of its straight line (csrc/fq_autograd_node.cpp: a gradient of another dtype or none at all, create_graph,
    an unaligned gradient): the Python node's backward, called under the GIL on the engine's thread.  Never in place from here -- the
    reference counts `_inplace_ok` compares are those of a Python Function's arguments -- and the node has told its epoch cell already."""
    ctx = _CppCtx()
    ctx.fq = (rows_w, rows_x, cols, code, _CODE_DTYPE[code], None)
    ctx.fq_foreign = True
    ctx.saved_tensors = (side_w, side_x)
    ctx.needs_input_grad = (need_w, need_x, False, False, False)
    out = _PairNode.backward(ctx, gw, gx)
    return out[0], out[1]


def quantize_kv(key_states, value_states, clip_val_k, clip_val_v, num_bits):
    """The two KV-cache hooks of the attention block (models/modeling_llama_quant.py:320-327),

        key_states   = self.act_quantizer_k.apply(key_states,   self.act_clip_val_k, self.kv_bits, False)
        value_states = self.act_quantizer_v.apply(value_states, self.act_clip_val_v, self.kv_bits, False)

    in ONE forward launch (K and V are [bsz, q_len, hidden] tensors of the same dtype: same row length, same launch shape).
    Results and gradients are bit-identical to the two calls; under autocast both come back in fp32, as the reference's do.  Falls back
    to the two calls whenever the pair is not served.  Each result has its OWN autograd node over its own side buffer (two backward
    launches, as in the reference): one node over both would tie V's producer into K's graph, and a result that is then never used
    would drag a dead branch into the backward pass (a checkpointed producer would be recomputed and hand zero gradients to its parameters
    where the reference leaves None; a hook on the unused result would be called with None) -- tests/test_gpu_random_programs.py."""
    k, v = key_states, value_states
    if type(num_bits) is not int:
        num_bits = ops.bits_arg(num_bits)
    if torch.compiler.is_compiling():
        return (compiled.fake_quant("sym", k, clip_val_k, num_bits, False), compiled.fake_quant("sym", v, clip_val_v, num_bits, False))
    lo, hi = _clip_pair(clip_val_k)
    if (_PAIR and _BACKWARD_MODE == "mask" and k.is_cuda and _clip_pair(clip_val_v) == (lo, hi) and 1 <= num_bits < 32
            and k.dim() <= 3 and v.dim() <= 3):
        grad = torch.is_grad_enabled()
        need_k, need_v = grad and k.requires_grad, grad and v.requires_grad
        res = ops.pair_forward(k, v, num_bits, num_bits, lo, hi, need_k, need_v, wide=True)
        if res is not None:
            kq, vq, side_k, side_v, rows_k, rows_v, cols = res
            return (_precomputed(k, kq, side_k, rows_k, cols, (lo, hi)) if need_k else kq,
                    _precomputed(v, vq, side_v, rows_v, cols, (lo, hi)) if need_v else vq)
    return (SymQuantizer.apply(k, clip_val_k, num_bits, False), SymQuantizer.apply(v, clip_val_v, num_bits, False))


# 7. The KV-cache hooks as the reference writes them (models/modeling_llama_quant.py:317-327),
#        key_states = self.k_proj(hidden_states);  value_states = self.v_proj(hidden_states)
#        key_states   = SymQuantizer.apply(key_states,   clip_k, kv_bits, False)
#        value_states = SymQuantizer.apply(value_states, clip_v, kv_bits, False)
#    are two launches forward and two backward on tensors of one shape.  With the call site UNTOUCHED: every QuantizeLinear
#    notes its output (a weak reference, per thread); when SymQuantizer.apply receives such an
#    output and the very next noted output has the same shape / dtype / device / stream (K, then V), both are fake-quantized in ONE
#    forward launch, K's result is returned and V's is kept -- as plain data, outside any graph -- for the apply call that follows,
#    which must present that very tensor, unmodified, with the same clip / bits / grad mode / autocast state / stream and no fake-quant
#    backward in between; anything else discards it (the speculation then cost one tensor's forward, nothing else).  K and V each get
#    their OWN autograd node over their own side buffer (two backward launches, as in the reference; the explicit quantize_kv() call
#    does the same): a node over both would tie V's producer into K's graph -- before anyone asked for V, or although its result ends
#    up unused.  The decision depends only on the call sequence, so a checkpointed forward and its recompute build the same graph.
#    Results and gradients are bit-identical to the two calls (tests/test_tiny_llama.py, tests/test_gpu_features.py).
#    LLMQAT_AMD_PAIR_KV=0 / pair_kv_hooks(False) turn it off; so does a torch build without the private APIs it leans on (fail closed).
_PAIR_KV = os.environ.get("LLMQAT_AMD_PAIR_KV", "1") != "0"


def pair_kv_hooks(flag=True):
    global _PAIR_KV
    _PAIR_KV = bool(flag)


def _note_output(st, out):
    """QuantizeLinear.forward: remember (weakly) the last few outputs of this thread, in order, with the stream that produced them"""
    if out.is_inference():   # no version counter (torch.inference_mode): such outputs are never paired
        return
    rec = st.outs
    if len(rec) >= 4:
        del rec[0]
    rec.append((weakref.ref(out), out._version, _region() if _top_hooks is not None else None, ops._stream(out)))


def _kv_state(st, clip_val, num_bits, stream):
    return (_clip_pair(clip_val), num_bits, torch.is_grad_enabled(), torch.is_autocast_enabled("cuda"), _BACKWARD_MODE, ops._semantics, st.epoch, _region(), stream)


_kv_off = set()   # call signatures whose speculation was thrown away once: they stop pairing


def _kv_discard(st):
    """a V result nobody asked for: forget it (and its side buffer), and stop guessing for that call signature"""
    stash, st.kv = st.kv, None
    if stash is None:
        return
    _count("kv_pair_discarded")
    if stash[4] not in _kv_off:
        _kv_off.add(stash[4])
        _count("kv_pair_learned_off")


def _kv_hook(x, clip_val, num_bits):
    """-> the fake-quantized x if it is served from / by a K+V pair launch, else None (the ordinary single call runs)"""
    if (not (_PAIR and _BACKWARD_MODE == "mask" and x.is_cuda and 1 <= num_bits < 32 and x.dim() <= 3) or x.is_inference()
            or torch.compiler.is_compiling()):
        return None
    if _functorch_active is None:
        _disabled("kv_pair", "no_functorch_api")
        return None
    if _functorch_active() or not _memory_ok("kv_pair"):
        return None
    st = _state()
    stream = ops._stream(x)
    stash = st.kv
    if stash is not None:
        ref, ver, vres, state, sig = stash
        if ref() is x and ver == x._version and state == _kv_state(st, clip_val, num_bits, stream):
            st.kv = None
            _count("kv_pair_hit")
            vq, side_v, rows_v, cols, clip = vres   # V: quantized together with K a moment ago; its autograd node is built only now
            return vq if side_v is None else _precomputed(x, vq, side_v, rows_v, cols, clip)
        _kv_discard(st)
    rec = st.outs
    if not rec:
        return None
    for i in range(len(rec) - 1):
        if rec[i][0]() is x:
            if rec[i][1] != x._version:
                return None
            region = _region()
            if rec[i][2] is not region or rec[i + 1][2] is not region:
                return None     # K's or V's projection ran in another saved-tensor-hooks region (see _region)
            if rec[i][3] != stream or rec[i + 1][3] != stream:
                return None     # produced on another stream than the one this launch would read them on: no pairing
            v, vver = rec[i + 1][0](), rec[i + 1][1]
            if (v is None or v is x or vver != v._version or v.shape != x.shape or v.dtype != x.dtype or v.device != x.device
                    or not v.is_contiguous() or not x.is_contiguous() or v.requires_grad != x.requires_grad):
                return None
            lo, hi = _clip_pair(clip_val)
            sig = (x.shape, x.dtype, num_bits, lo, hi)
            if sig in _kv_off:
                return None
            grad = torch.is_grad_enabled()
            need = grad and x.requires_grad
            res = ops.pair_forward(x, v, num_bits, num_bits, lo, hi, need, need, wide=True)
            if res is None:
                return None
            _count("kv_pair_launch")
            # One launch forward, but NOT one autograd node: until the V call arrives, V's result is plain data (no graph refers to the
            # tensor V), so a guess that turns out wrong leaves nothing behind -- in particular it cannot make V's producer reachable from
            # a loss that never uses V (a checkpointed producer would be recomputed and hand ZERO gradients to its parameters where the
            # reference leaves None: tests/test_gpu_random_programs.py).  Each of K and V gets its own node over its own side buffer.
            kq, vq, side_k, side_v, rows_k, rows_v, cols = res
            st.kv = (weakref.ref(v), v._version, (vq, side_v if need else None, rows_v, cols, (lo, hi)), _kv_state(st, clip_val, num_bits, stream), sig)
            return _precomputed(x, kq, side_k, rows_k, cols, (lo, hi)) if need else kq
    return None


def reset_learned_state():
    """forget what the host logic has learned about call sites (K/V signatures that stopped pairing) and everything the CALLING thread
    remembers between calls (a pending V result, the last fake-quantized activations, the outputs it noted)"""
    _kv_off.clear()
    st = _state()
    st.acts, st.kv, st.outs = {}, None, []


def conservative(flag=True):
    """One switch for "the reference's structure, only faster": with it on, every reference call is exactly one kernel launch and one
    autograd node, gradients are written out of place, and nothing is remembered between calls -- no operand pairing, no shared
    activation fake-quant, no K/V pairing at the hooks, no weight cache, no in-place weight gradient.  Results are bit-identical either way (tests/test_tiny_llama.py::test_conservative_mode_is_bit_identical); the switch
    exists to take the stateful host logic out of the picture when debugging a training run.  `conservative(False)` restores the
    defaults (not the environment's settings).  Environment: LLMQAT_AMD_CONSERVATIVE=1."""
    pair_operands(not flag)
    share_activation_quant(not flag)
    pair_kv_hooks(not flag)
    inplace_weight_grad(not flag)
    if flag:
        enable_weight_quant_cache(False)


if os.environ.get("LLMQAT_AMD_CONSERVATIVE", "0") == "1":
    conservative(True)


class _ReuseQuantizedWeight(torch.autograd.Function):
    """Autograd node over an already computed (value, row bounds, STE mask) triple of a weight:
    forward launches nothing, backward is the ordinary STE backward."""

    @staticmethod
    def forward(ctx, weight, cached, clip_val):
        y, bounds, mask, rows_cols = cached
        ctx.rows_cols = rows_cols
        ctx.clip = _clip_pair(clip_val)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(weight if mask is None else None, bounds, mask)
        return y.view_as(y)

    @staticmethod
    @_graph_aware
    def backward(ctx, grad_output):
        if grad_output is None:
            return None, None, None
        inplace = _INPLACE_WGRAD and _inplace_ok(grad_output)
        lo, hi = ctx.clip
        weight, row_bounds, ste_mask = ctx.saved_tensors
        if ste_mask is not None:
            rows, cols = ctx.rows_cols
            return ops.ste_backward_mask(grad_output, lo, hi, row_bounds, ste_mask, rows, cols, inplace=inplace), None, None
        bounds = row_bounds if grad_output.is_contiguous() and weight.is_contiguous() else None
        return ops.ste_backward(grad_output, weight, lo, hi, row_bounds=bounds, rows_cols_hint=ctx.rows_cols), None, None


class _PrecomputedAct(torch.autograd.Function):
    """Autograd node over ONE tensor that a multi-tensor launch has already fake-quantized: forward launches nothing, backward is that
    tensor's own STE launch.  Used where the tensors of a launch must NOT share a node: with the weight cache on (it saves exactly what
    _SymQuantizerOperand saves in mask mode, so a checkpointed forward that pairs and its recompute that does not record the same
    tensors), and for the K / V speculation at the unchanged hooks (point 7: a V result nobody asks for must leave no trace in the graph).
    y may be the reference's fp32 result under autocast (then fp32 gradients come back: the wide backward)."""

    @staticmethod
    def forward(ctx, x, res, rows, cols, clip):
        y, side = res     # (inside a tuple: not inputs of this node -- y is the launch's fresh result and becomes the output as it is, so that
        ctx.rows_cols, ctx.clip, ctx.dtype, ctx.fq_st = (rows, cols), clip, x.dtype, _state().ref     # in-place ops on it work as in the reference)
        ctx.wide = y.dtype != x.dtype
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(side)
        return y

    @staticmethod
    @_graph_aware
    def backward(ctx, grad_output):
        if grad_output is None:
            return None, None, None, None, None
        _backward_started(ctx.fq_st)
        (side,) = ctx.saved_tensors
        lo, hi = ctx.clip
        rows, cols = ctx.rows_cols
        if ctx.wide:
            return ops.train_backward_wide(grad_output, side, rows, cols, lo, hi, ctx.dtype), None, None, None, None
        g = grad_output if grad_output.dtype == ctx.dtype else grad_output.to(ctx.dtype)
        return ops.train_backward(g, side, rows, cols, lo, hi), None, None, None, None


class _CppCtx1:
    """what _PrecomputedAct.backward reads from its ctx, for a backward the C++ one-tensor node hands back"""
    __slots__ = ("rows_cols", "clip", "dtype", "fq_st", "wide", "saved_tensors")


def _one_backward_from_cpp(g, side, rows, cols, lo, hi, code, wide):
    """the C++ one-tensor node's way out of its straight line (another gradient dtype, create_graph, strided or unaligned gradients)"""
    ctx = _CppCtx1()
    ctx.rows_cols, ctx.clip, ctx.dtype, ctx.fq_st, ctx.wide, ctx.saved_tensors = (rows, cols), (lo, hi), _CODE_DTYPE[code], None, wide, (side,)
    return _PrecomputedAct.backward(ctx, g)[0]


def _precomputed(x, y, side, rows, cols, clip):
    """a node of x's own over a result some launch has already produced: the C++ one (csrc/fq_autograd_node.cpp::FqOneNode) or _PrecomputedAct"""
    if _USE_CNODE and x.is_cuda:
        return _cnode.one_node(x, y, side, rows, cols, clip[0], clip[1], _state().cell)
    return _PrecomputedAct.apply(x, (y, side), rows, cols, clip)


class QuantizeLinear(nn.Linear):
    _fq_plan = None   # (input shape, input dtype, weight dtype, ops.pair_plan): the launch plan of the last input shape (a plain attribute)

    def __init__(self, *kargs, symmetric=True, bias=False, w_bits=32, a_bits=32, act_layerwise=False,
                 weight_layerwise=False):
        super().__init__(*kargs, bias=False)  # `bias` is accepted and ignored, as in the reference (:176)
        self.w_bits = w_bits
        self.a_bits = a_bits
        self.act_layerwise = act_layerwise
        self.weight_layerwise = weight_layerwise
        if 2 < self.a_bits < 32:
            self.act_quantizer = SymQuantizer if symmetric else AsymQuantizer   # (attribute absent otherwise, as in the reference :184-188)
        self._act_kind = "sym" if symmetric else "asym"  # what torch.compile's trace reads (a class identity test does not trace)

    def _low_bit_weight(self, w):
        """1- and 2-bit branches (reference :202-242): mean-|w| scale, sign / 2-level rounding, identity gradient (the reference's
        detach trick).  One launch where the kernel reproduces ATen's summation order (a sum is order dependent: that keeps the scale
        bit-identical to the reference's on this device); otherwise ATen's own abs + mean, then the ~10 elementwise kernels as one."""
        if w.device.type == "cpu":
            if not cpu_tensors.ENABLED:
                cpu_tensors.refuse(w, "low_bit_weight")
            return _LowBitWeightCpu.apply(w, self.w_bits, self.weight_layerwise)
        if _W12_FUSED and not self.weight_layerwise and w.is_cuda and w.is_contiguous() and _w12_fused_verified(w):
            try:
                return _LowBitWeightFused.apply(w, self.w_bits)
            except _NotServed:
                pass
        with torch.no_grad():
            absmean = w.abs().mean() if self.weight_layerwise else w.abs().mean(dim=1, keepdim=True)
            sc = absmean if self.w_bits == 1 else 2 * absmean
        return _LowBitWeight.apply(w, sc, self.w_bits)

    def _wcache_key(self):
        w = self.weight
        return (id(w), w._version, w.data_ptr(), self.w_bits, self.weight_layerwise, _MODE_CODE[_BACKWARD_MODE] + 4 * ops._semantics + (16 if ops.autocast_active(w) else 0))

    def _quantized_weight(self):
        w = self.weight
        ac = ops.autocast_active(w)
        if not _WEIGHT_CACHE or not w.is_cuda or w.is_inference() or (ac and not ops.autocast_narrow_ok(w)):
            # (an fp16 weight inside autocast(bf16), or the reverse, gets the reference's fp32 result and an fp32 gradient:
            # the plain node handles both dtypes; the cache's node works in the weight's dtype only)
            return _SymQuantizerWeight.apply(w, _CLIP, self.w_bits, self.weight_layerwise)
        key = self._wcache_key()
        ent = getattr(self, "_fq_wcache", None)
        if ent is not None and ent[0] == key:
            cached = ent[1]
            _count("wcache_hit")
            if not _WEIGHT_CACHE_PERSISTENT:
                self._fq_wcache = None  # second use within the step (the checkpoint recompute): done with it
        else:
            rc = ops.rows_cols(tuple(w.shape), self.weight_layerwise)
            bounds = mask = None
            if ac:  # autocast arithmetic, result rounded once to the weight dtype (see _SymQuantizerOperand)
                y, side, rows, _, got = ops.sym_forward_autocast(w, self.w_bits, self.weight_layerwise, wide=False,
                                                                 train=None if _BACKWARD_MODE == "plain" else _BACKWARD_MODE)
                if got == "mask":
                    bounds, mask = side[: rows * 8].view(torch.float32).view(rows, 2), side[rows * 8:]
                elif got == "bounds":
                    bounds = side
            else:
                res = ops.quantize_train("sym", w, self.w_bits, self.weight_layerwise, -2.0, 2.0) if _BACKWARD_MODE == "mask" else None
                if res is not None:
                    y, bounds, mask = res
                elif _BACKWARD_MODE == "plain":
                    y = ops.sym_quantize(w, self.w_bits, self.weight_layerwise)
                else:
                    y, bounds = ops.sym_quantize(w, self.w_bits, self.weight_layerwise, want_bounds=True)
            cached = (y, bounds, mask, rc)
            self._fq_wcache = (key, cached)
            _count("wcache_fill")
        if torch.is_grad_enabled() and w.requires_grad:
            return _ReuseQuantizedWeight.apply(w, cached, _CLIP)  # launches nothing; backward = the ordinary STE
        return cached[0]

    def _pair_forward(self, input_):
        """weight and input of this module under ONE autograd node: both in one launch, or -- when a sibling projection has already
        fake-quantized this input -- the weight's own launch + the remembered activation.  None when not applicable (then the ordinary
        two calls run).  What depends only on shapes / dtype / device is decided once per module and input shape (`ops.pair_plan`)."""
        if not (_PAIR and _BACKWARD_MODE == "mask"):
            return None
        weight = self.weight
        plan = self._fq_plan
        # (the module's settings are plain attributes a caller may change after construction: they are part of what a plan is valid for)
        if (plan is None or plan[0] != input_.shape or plan[1] is not input_.dtype or plan[2] is not weight.dtype
                or plan[4] != (self.w_bits, self.a_bits, self.act_layerwise, self.weight_layerwise, getattr(self, "act_quantizer", None))):
            ok = (3 <= self.w_bits < 32 and 2 < self.a_bits < 32 and getattr(self, "act_quantizer", None) is SymQuantizer and not self.act_layerwise
                  and not self.weight_layerwise)
            plan = self._fq_plan = (input_.shape, input_.dtype, weight.dtype, ops.pair_plan(weight, input_) if ok else None,
                                    (self.w_bits, self.a_bits, self.act_layerwise, self.weight_layerwise, getattr(self, "act_quantizer", None)),
                                    (ops.bits_arg(self.w_bits), ops.bits_arg(self.a_bits)) if ok else None)
        pp = plan[3]
        if pp is None or _WEIGHT_CACHE or not (weight.is_contiguous() and input_.is_contiguous()) or input_.is_inference() or weight.is_inference():
            return self._pair_forward_general(input_)
        grad = torch.is_grad_enabled()
        need_w, need_x = grad and weight.requires_grad, grad and input_.requires_grad
        ac = pp[0] != 0 and torch.is_autocast_enabled("cuda")      # (code 0: fp32 tensors are untouched by autocast)
        if ac and torch.get_autocast_dtype("cuda") is not weight.dtype:
            return self._pair_forward_general(input_)
        share = _SHARE_ACT and _top_hooks is not None
        if share:
            st = _state()
            key = (_SymQuantizerOperand, self.a_bits, False, _MODE_CODE["mask"] + 4 * ops._semantics + (8 if grad else 0) + (16 if ac else 0))
            region, stream = _region(), ops._raw_stream(pp[6])
            raw = _act_lookup(st, key, input_, region, stream)
            if raw is not None:
                # a sibling projection already quantized this activation: only the weight is left to do.  One node over both where the
                # remembered data has the pair's shape (mask mode, operand dtype); otherwise the two ordinary nodes
                if raw.mode not in ("mask", "none") or raw.out.dtype is not weight.dtype:
                    return None
                if _USE_CNODE:     # the weight's launch and the node in one call into C++
                    out = _cnode.weight_forward_node(weight, input_, raw.out, raw.saved[0] if need_x else None, pp[0], pp[1], pp[2], pp[3], pp[4], plan[5][0],
                                                     need_w, need_x, ac, ops._SEM_AUTOCAST if ac else ops._semantics, st.cell)
                    if out is not None:
                        _count("act_share_hit")
                        _count("single_launch")
                        return out
                res = ops.weight_forward(weight, self.w_bits, -2.0, 2.0, need_w)
                if res is None:
                    return None
                _count("act_share_hit")
                _count("single_launch")
                wq, side_w, rows_w, cols = res
                if not (need_w or need_x):
                    return wq, raw.out
                if _USE_CNODE:
                    return _cnode.pair_node(weight, input_, wq, raw.out, side_w, raw.saved[0] if need_x else None, rows_w, pp[3], cols, pp[0], True, st.cell)
                return _PairNode.apply(weight, input_, (wq, raw.out, side_w, raw.saved[0] if need_x else None, rows_w, pp[3], cols), pp[0], True)
        elif _SHARE_ACT:
            _memory_ok("share")   # counts / warns: the region API is missing, nothing is remembered
        if _USE_CNODE:
            # allocations, the launch and the node in one call into C++ (csrc/fq_autograd_node.cpp::pair_forward)
            w_bits, a_bits = plan[5]
            out = _cnode.pair_forward(weight, input_, pp[0], pp[1], pp[2], pp[3], pp[4], pp[5], w_bits, a_bits, need_w, need_x, ac,
                                      ops._SEM_AUTOCAST if ac else ops._semantics, (st if share else _state()).cell)
            if out is None:
                return self._pair_forward_general(input_)
            side_x = out[2]
        else:
            res = ops.pair_forward(weight, input_, self.w_bits, self.a_bits, -2.0, 2.0, need_w, need_x)
            if res is None:
                return self._pair_forward_general(input_)
            out = _PairNode.apply(weight, input_, res, pp[0], False) if (need_w or need_x) else (res[0], res[1])
            side_x = res[3]
        _count("pair_launch")
        if share:
            # remembered for the sibling projections: the data (for a sibling's view of it: _PairNode / _SharedAct) + its side buffer.
            # out[1] is this node's own output tensor when there is a node; siblings never use it as a tensor of their graph.
            _count("act_share_miss")
            st.acts[key] = (weakref.ref(input_), input_._version, input_.data_ptr(), _Raw(out[1], "mask", (side_x,), (-2.0, 2.0), (pp[3], pp[1])) if need_x
                            else _Raw(out[1], "none"), out[1]._version, input_.requires_grad, region, stream)
        return out[0], out[1]

    def _pair_forward_general(self, input_):
        """the same decision without a plan: strided operands, the weight cache, inference tensors, dtype mixes (ops.pair_forward's checks)"""
        if not (3 <= self.w_bits < 32 and 2 < self.a_bits < 32):
            return None
        weight = self.weight
        if input_.is_inference() or weight.is_inference():
            return None   # no version counters under torch.inference_mode: nothing is paired, shared or remembered
        if self.act_quantizer is not SymQuantizer or self.act_layerwise or self.weight_layerwise:
            return None
        wkey = None
        if _WEIGHT_CACHE:
            # With the weight cache on, the FIRST use of a weight in a step still shares a launch with its input and fills the
            # cache from it; the second use (the checkpoint recompute) finds the entry and launches nothing for the weight.
            wkey = self._wcache_key()
            ent = getattr(self, "_fq_wcache", None)
            if ent is not None and ent[0] == wkey:
                return None
        grad = torch.is_grad_enabled()
        need_w, need_x = grad and weight.requires_grad, grad and input_.requires_grad
        share = _SHARE_ACT and _memory_ok("share")
        if share:
            st = _state()
            key = (_SymQuantizerOperand, self.a_bits, False, _state_word(input_))
            region, stream = _region(), (ops._stream(input_) if input_.is_cuda else 0)
            if _act_lookup(st, key, input_, region, stream) is not None:
                return None   # a sibling already quantized this activation: the weight's own node + a _SharedAct over the remembered data
        res = ops.pair_forward(weight, input_, self.w_bits, self.a_bits, -2.0, 2.0, need_w or wkey is not None, need_x)
        if res is None:
            return None
        _count("pair_launch")
        code = ops._DTYPES[weight.dtype]
        if share:
            _count("act_share_miss")
            rows_x, cols = res[5], res[6]
            _act_store(st, key, input_, _Raw(res[1], "mask", (res[3],), (-2.0, 2.0), (rows_x, cols)) if need_x else _Raw(res[1], "none"), region, stream)
        if wkey is not None:
            # bounds + mask are recorded even without grad (the recompute pass's backward needs them), and the results are
            # wrapped in the SAME nodes the recompute pass will build (_ReuseQuantizedWeight for the weight, a side-buffer
            # node for the input), so non-reentrant checkpointing sees identical saved tensors in both passes
            rows_w, rows_x, cols = res[4], res[5], res[6]
            side_w, side_x = res[2], res[3]
            cached = (res[0], side_w[: rows_w * 8].view(torch.float32).view(rows_w, 2), side_w[rows_w * 8:], ops.rows_cols(tuple(weight.shape), False))
            self._fq_wcache = (wkey, cached)
            _count("wcache_fill")
            wq = _ReuseQuantizedWeight.apply(weight, cached, _CLIP) if need_w else res[0]
            xq = _precomputed(input_, res[1], side_x, rows_x, cols, (-2.0, 2.0)) if need_x else res[1]
            return wq, xq
        if need_w or need_x:
            return _PairNode.apply(weight, input_, res, code, share)
        return res[0], res[1]

    def export_weight(self, container=None):
        """The integer form of this layer's fake-quantized weight for an inference export: packed bins (int4 for
        w_bits <= 4, int8 / int16 above) + per-output-channel {s, t2} (ops.QuantExport; `dequantize()` gives back the
        value the forward multiplies with, bit for bit where overflow == 0).  Serves the w_bits >= 3 path (:195-201)."""
        if not 3 <= self.w_bits < 32:
            raise ValueError(f"export_weight serves 3 <= w_bits < 32 (SymQuantizer weights), this layer has w_bits={self.w_bits}")
        return ops.sym_export(self.weight.detach(), self.w_bits, self.weight_layerwise, container=container)

    def _forward_compiled(self, input_):
        """forward while torch.compile traces: the same kernels as custom ops, no Python-side caches (compiled.py)"""
        if self.w_bits >= 32:
            weight = self.weight
        elif self.w_bits >= 3:
            weight = compiled.fake_quant("sym", self.weight, _CLIP, self.w_bits, self.weight_layerwise, narrow=True)
        else:
            with torch.no_grad():
                absmean = self.weight.abs().mean() if self.weight_layerwise else self.weight.abs().mean(dim=1, keepdim=True)
                sc = absmean if self.w_bits == 1 else 2 * absmean
            weight = compiled.low_bit_weight_op(self.weight, sc, self.w_bits)
        if 2 < self.a_bits < 32:
            input_ = compiled.fake_quant(self._act_kind, input_, _CLIP, self.a_bits, self.act_layerwise, narrow=True)
        out = nn.functional.linear(input_, weight)
        if self.bias is not None:
            out += self.bias.view(1, -1).expand_as(out)
        return out

    def forward(self, input_):
        assert len(self.weight.size()) == 2
        if torch.compiler.is_compiling():
            return self._forward_compiled(input_)
        pair = self._pair_forward(input_)
        if pair is not None:
            weight, input_ = pair
        else:
            _count("single_launch")
            if self.w_bits >= 32:
                weight = self.weight
            elif self.w_bits >= 3:
                weight = self._quantized_weight()
            else:
                weight = self._low_bit_weight(self.weight)
            if 2 < self.a_bits < 32:
                input_ = _shared_activation(self.act_quantizer, input_, self.a_bits, self.act_layerwise)
        out = nn.functional.linear(input_, weight)
        if self.bias is not None:
            out += self.bias.view(1, -1).expand_as(out)
        if _PAIR_KV and out.is_cuda:
            _note_output(_state(), out)   # the KV-cache hooks may follow (point 7): ~1 us of bookkeeping
        return out


try:
    if _storage_use_count is None:
        raise RuntimeError("torch._C._storage_Use_Count is not available in this torch build")
    _calibrate_grad_counts()
except Exception as _e:  # noqa: BLE001 -- without a baseline _inplace_ok() answers False: every gradient takes the copying launch
    _ref_base.clear()
    _log.warning("llm_qat_amd: the in-place weight-gradient guard could not be calibrated (%r): every weight gradient takes the copying "
                 "launch (same results; stats() counts them under inplace_refused:uncalibrated)", _e)


def _calibrate_cnode():
    """The C++ node's in-place guard learns, in its own backward, what the reference counts of a gradient nobody else holds look like
    (csrc/fq_autograd_node.cpp::inplace_ok): F.linear's wgrad (a view of a temporary), a reshaped one, a plain fresh tensor -- tiny CPU
    tensors, once at import.  Then the node is bound to the kernel library's entry points and to the Python backward it falls back on."""
    global _cnode_ready, _USE_CNODE
    _cnode.arm_probe(True)
    try:
        with torch.inference_mode(False), torch.enable_grad():
            for loss in (lambda wq, xq: nn.functional.linear(xq, wq).sum(), lambda wq, xq: (wq * 2.0).sum() + xq.sum(),
                         lambda wq, xq: (wq.view(-1) * 2.0).sum() + xq.sum()):
                w = torch.zeros(2, 4, requires_grad=True)
                x = torch.zeros(3, 4, requires_grad=True)
                loss(*_cnode.probe_node(w, x)).backward()
    finally:
        _cnode.arm_probe(False)
    base = _cnode.baselines()
    if "view" not in base or "plain" not in base:
        raise RuntimeError(f"reference-count baselines incomplete: {base}")
    _cnode.set_inplace(_INPLACE_WGRAD)
    if not _node.bind(_pair_backward_from_cpp, _one_backward_from_cpp, _forget_from_cpp):
        raise RuntimeError("could not bind the node to the kernel library")
    _cnode_ready = True
    _USE_CNODE = True


if _cnode is not None:
    _USE_CNODE = False
    try:
        _calibrate_cnode()
    except Exception as _e:  # noqa: BLE001 -- the Python node serves (same launches, same results); host_node() says why
        _log.warning("llm_qat_amd: the C++ autograd node is not used (%r): QuantizeLinear builds its Python node", _e)

