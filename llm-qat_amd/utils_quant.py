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
import os

import torch
import torch.nn as nn

from . import ops

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
    lo, hi = clip_val.tolist()[:2] if clip_val.dim() else (clip_val.item(),) * 2
    return float(lo), float(hi)


class _FakeQuantFunction(torch.autograd.Function):
    _kind = None

    @staticmethod
    def _fwd(kind, ctx, input, clip_val, num_bits, layerwise):
        mode = _BACKWARD_MODE if ctx.needs_input_grad[0] else "plain"
        ctx.fq_mode = "plain"
        ctx.row_bounds = ctx.ste_mask = None
        if mode != "plain":
            ctx.rows_cols = ops.rows_cols(tuple(input.shape), layerwise)
        if mode == "mask":
            lo, hi = _clip_pair(clip_val)
            res = ops.quantize_train(kind, input, num_bits, layerwise, lo, hi)
            if res is not None:
                out, ctx.row_bounds, ctx.ste_mask = res
                ctx.fq_mode, ctx.clip = "mask", (lo, hi)
                ctx.in_shape = input.shape
                return out  # the input itself is not needed again
            mode = "bounds"
        ctx.save_for_backward(input, clip_val)  # reference :45 / :104 -- the unclipped input itself
        fn = ops.sym_quantize if kind == "sym" else ops.asym_quantize
        if mode == "bounds":
            out, ctx.row_bounds = fn(input, num_bits, layerwise, want_bounds=True)
            ctx.fq_mode = "bounds"
        else:
            out = fn(input, num_bits, layerwise)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        if ctx.fq_mode == "mask":
            lo, hi = ctx.clip
            rows, cols = ctx.rows_cols
            return ops.ste_backward_mask(grad_output, lo, hi, ctx.row_bounds, ctx.ste_mask, rows, cols), None, None, None
        input, clip_val = ctx.saved_tensors  # reference :83 / :158
        lo, hi = _clip_pair(clip_val)
        bounds = ctx.row_bounds
        if bounds is not None and not (input.is_contiguous() and grad_output.is_contiguous()):
            bounds = None
        grad_input = ops.ste_backward(grad_output, input, lo, hi, row_bounds=bounds,
                                      rows_cols_hint=getattr(ctx, "rows_cols", None))
        return grad_input, None, None, None


class SymQuantizer(_FakeQuantFunction):
    """uniform symmetric (absmax) fake quantization; straight-through gradient masked to clip_val"""

    @staticmethod
    def forward(ctx, input, clip_val, num_bits, layerwise):
        return _FakeQuantFunction._fwd("sym", ctx, input, clip_val, num_bits, layerwise)


class AsymQuantizer(_FakeQuantFunction):
    """min-max (affine) fake quantization; same straight-through gradient"""

    @staticmethod
    def forward(ctx, input, clip_val, num_bits, layerwise):
        return _FakeQuantFunction._fwd("asym", ctx, input, clip_val, num_bits, layerwise)


_CLIP = torch.tensor([-2.0, 2.0])  # the literal the reference rebuilds on every call (:198, :245)


class QuantizeLinear(nn.Linear):
    def __init__(self, *kargs, symmetric=True, bias=False, w_bits=32, a_bits=32, act_layerwise=False,
                 weight_layerwise=False):
        super().__init__(*kargs, bias=False)  # `bias` is accepted and ignored, as in the reference (:176)
        self.w_bits = w_bits
        self.a_bits = a_bits
        self.act_layerwise = act_layerwise
        self.weight_layerwise = weight_layerwise
        if 2 < self.a_bits < 32:
            self.act_quantizer = SymQuantizer if symmetric else AsymQuantizer

    def _low_bit_weight(self, w):
        """1- and 2-bit branches (reference :202-242): mean-|w| scale, sign / 2-level rounding,
        identity gradient through the detach trick.  Eager torch for now."""
        dims = None if self.weight_layerwise else 1
        absmean = w.abs().mean() if dims is None else w.abs().mean(dim=1, keepdim=True)
        if self.w_bits == 1:
            sc = absmean.detach()
            q = sc * torch.sign(w / sc)
        else:
            levels = 2 ** (self.w_bits - 1)
            bound = 1 - 1e-2
            sc = (2 * absmean).detach()
            q = sc * (torch.round(torch.clamp(w / sc, -bound, bound) * levels - 0.5) + 0.5) / levels
        return q.detach() - w.detach() + w

    def forward(self, input_):
        assert len(self.weight.size()) == 2
        if self.w_bits >= 32:
            weight = self.weight
        elif self.w_bits >= 3:
            weight = SymQuantizer.apply(self.weight, _CLIP, self.w_bits, self.weight_layerwise)
        else:
            weight = self._low_bit_weight(self.weight)
        if 2 < self.a_bits < 32:
            input_ = self.act_quantizer.apply(input_, _CLIP, self.a_bits, self.act_layerwise)
        out = nn.functional.linear(input_, weight)
        if self.bias is not None:
            out += self.bias.view(1, -1).expand_as(out)
        return out
