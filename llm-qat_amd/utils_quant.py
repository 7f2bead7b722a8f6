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

# Record per-row value bounds in forward so backward can skip re-reading x for rows that
# cannot be clipped (weights practically always): 4 instead of 6 bytes/element of HBM traffic.
_ROW_BOUNDS = os.environ.get("LLMQAT_AMD_ROW_BOUNDS", "1") != "0"


def _clip_pair(clip_val):
    lo, hi = clip_val.tolist()[:2] if clip_val.dim() else (clip_val.item(),) * 2
    return float(lo), float(hi)


class _FakeQuantFunction(torch.autograd.Function):
    _kind = None

    @staticmethod
    def _fwd(kind, ctx, input, clip_val, num_bits, layerwise):
        ctx.save_for_backward(input, clip_val)  # reference :45 / :104 -- the unclipped input itself
        use_bounds = _ROW_BOUNDS and ctx.needs_input_grad[0]
        fn = ops.sym_quantize if kind == "sym" else ops.asym_quantize
        if use_bounds:
            out, bounds = fn(input, num_bits, layerwise, want_bounds=True)
            ctx.row_bounds = bounds
            ctx.rows_cols = ops.rows_cols(tuple(input.shape), layerwise)
        else:
            out = fn(input, num_bits, layerwise)
            ctx.row_bounds = None
        return out

    @staticmethod
    def backward(ctx, grad_output):
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
