"""The reference's eager-PyTorch op chain, restated op for op -- TEST INFRASTRUCTURE ONLY.

Second oracle beside fq_oracle.c.  It executes the same ATen ops, in the same order, as
models/utils_quant.py does (SURVEY §3.3 lists the dispatched sequence), so:
  * on CPU it is bit-equal to the reference (checked against tests/golden in the CPU tier)
    and is what bench.py times as `cpu_baseline` -- the reference's CPU path on the GPU
    box's own host cores;
  * on the GPU it shows what the reference's eager path computes on the device
    ("device-eager" semantics) and is the like-for-like "GPU before" timing
    (9 + 5 kernel launches per fwd+bwd instead of 1 + 1).
The reference file itself cannot travel to the GPU box; this restatement can.
"""
import torch


def _row_stat(fn, x, layerwise):
    """per-row statistic broadcast back to x's shape (utils_quant.py:50-70 / :110-143)"""
    if layerwise:
        return fn(x, None).expand_as(x)
    if x.dim() <= 3:
        return fn(x, -1).expand_as(x)
    if x.dim() == 4:
        flat = x.view(x.shape[0], x.shape[1], -1)
        return fn(flat, -1).unsqueeze(-1).expand_as(x)
    raise ValueError


def _max(t, dim):
    return torch.max(t) if dim is None else torch.max(t, dim=dim, keepdim=True)[0]


def _min(t, dim):
    return torch.min(t) if dim is None else torch.min(t, dim=dim, keepdim=True)[0]


def sym_forward(x, num_bits, layerwise=False, want_idx=False):
    """utils_quant.py:50-72"""
    top = _row_stat(lambda t, d: _max(torch.abs(t), d), x, layerwise)
    s = (2 ** (num_bits - 1) - 1) / (top + 1e-6)   # int / Tensor -> reciprocal() * int
    idx = torch.round(x * s)
    y = idx.div(s + 1e-6)
    return (y, idx, s) if want_idx else y


def asym_forward(x, num_bits, layerwise=False, want_idx=False):
    """utils_quant.py:110-147 (the min is reduced twice there too)"""
    alpha = _row_stat(_max, x, layerwise) - _row_stat(_min, x, layerwise)
    beta = _row_stat(_min, x, layerwise)
    n = (x - beta) / (alpha + 1e-8)
    levels = 2 ** num_bits - 1
    idx = torch.round(n * levels)
    y = idx.div(levels) * (alpha + 1e-8) + beta
    return (y, idx, alpha, beta) if want_idx else y


def ste_backward(grad_output, x, clip_val):
    """utils_quant.py:83-87"""
    gx = grad_output.clone()
    gx[x.ge(clip_val[1])] = 0
    gx[x.le(clip_val[0])] = 0
    return gx


class EagerSym(torch.autograd.Function):
    """autograd wrapper so whole-module comparisons (QuantizeLinear, tiny LLaMA) can run the eager chain"""

    @staticmethod
    def forward(ctx, x, clip_val, num_bits, layerwise):
        ctx.save_for_backward(x, clip_val)
        return sym_forward(x, num_bits, layerwise)

    @staticmethod
    def backward(ctx, g):
        x, clip_val = ctx.saved_tensors
        return ste_backward(g, x, clip_val), None, None, None


class EagerAsym(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, clip_val, num_bits, layerwise):
        ctx.save_for_backward(x, clip_val)
        return asym_forward(x, num_bits, layerwise)

    @staticmethod
    def backward(ctx, g):
        x, clip_val = ctx.saved_tensors
        return ste_backward(g, x, clip_val), None, None, None
