"""CPU tensors (opt-in): the reference accepts tensors on any device (models/utils_quant.py:37), and BASELINE configs[0] is a
tiny-LLaMA QAT step on CPU.  This module serves CPU tensors -- and nothing else -- with plain torch ops in the reference's op order
(:50-72, :110-147, :83-87, :203-238), so that the drop-in classes run where no GPU exists.

It is NOT a fallback: it is selected by `tensor.device.type == "cpu"` only, and only after `llm_qat_amd.allow_cpu_tensors(True)` /
LLMQAT_AMD_CPU_TENSORS=1 (the default keeps raising).  A CUDA tensor never comes here: if the HIP library is missing or a launch
fails, the GPU path raises.  Nothing here touches the test infrastructure; ATen's CPU kernels ARE the reference's CPU path.
"""
import os

import torch

ENABLED = os.environ.get("LLMQAT_AMD_CPU_TENSORS", "0") == "1"


def allow_cpu_tensors(flag=True):
    global ENABLED
    ENABLED = bool(flag)


def refuse(x, what):
    raise RuntimeError(f"{what}: tensor is on '{x.device}'. llm_qat_amd runs on MI355X and has no CPU fallback; move the tensor to the GPU. "
                       "(CPU tensors are served, with plain torch ops, only after llm_qat_amd.allow_cpu_tensors(True) / LLMQAT_AMD_CPU_TENSORS=1.)")


def _stat(x, layerwise, largest):
    """The statistic one scale is made of, KEPT SMALL: shape [..., 1] (ndim <= 3), [d0, d1, 1, 1] (4-D) or 0-dim (layerwise), and left to
    broadcast in the elementwise ops that follow.  The reference expands it to the input's shape first (:57, :67, :118-140); every element
    then sees the same scalar operands either way, so the values are the same bit for bit, and the per-row part of the chain runs on
    rows instead of on elements (what tests/test_cpu_tensors.py checks against the reference's own fixtures)."""
    pick = torch.max if largest else torch.min
    if layerwise:
        return pick(x)                                   # :51 / :111-112
    if x.dim() <= 3:
        return pick(x, dim=-1, keepdim=True)[0]          # :56 / :116-118
    if x.dim() == 4:
        return pick(x.view(x.shape[0], x.shape[1], -1), dim=-1, keepdim=True)[0].unsqueeze(-1)   # :63-66 / :127-139
    raise ValueError(f"fake-quant expects at most 4 dimensions, got {x.dim()}")   # :70


def forward(kind, x, num_bits, layerwise):
    if kind == "sym":
        s = (2 ** (num_bits - 1) - 1) / (_stat(torch.abs(x), layerwise, True) + 1e-6)    # int / Tensor: reciprocal() * int, per row
        return torch.round(x * s).div(s + 1e-6)                                          # :72
    beta = _stat(x, layerwise, False)
    span = (_stat(x, layerwise, True) - beta) + 1e-8                                     # alpha + 1e-8, used twice (:144, :147)
    levels = 2 ** num_bits - 1
    return torch.round((x - beta) / span * levels).div(levels) * span + beta


def backward(grad_output, x, clip_val):
    gx = grad_output.clone()
    gx[x.ge(clip_val[1])] = 0
    gx[x.le(clip_val[0])] = 0
    return gx


def low_bit_weight(w, w_bits, layerwise):
    """forward VALUE of the 1-/2-bit branches incl. the detach trick (:203-242); the caller supplies the identity gradient"""
    m = torch.mean(abs(w)) if layerwise else torch.mean(abs(w), dim=1, keepdim=True)
    if w_bits == 1:
        q = m * torch.sign(w / m)
    else:
        sc, nb, cv = 2 * m, 2 ** (w_bits - 1), 1 - 1e-2
        q = sc * (torch.round(torch.clamp(w / sc, -cv, cv) * nb - 0.5) + 0.5) / nb
    return q - w + w
