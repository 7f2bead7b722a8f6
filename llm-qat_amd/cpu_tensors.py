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


def _per_row(reduce, x, layerwise):
    if layerwise:
        return reduce(x, None).expand_as(x)
    if x.dim() <= 3:
        return reduce(x, -1).expand_as(x)
    if x.dim() == 4:
        return reduce(x.view(x.shape[0], x.shape[1], -1), -1).unsqueeze(-1).expand_as(x)
    raise ValueError(f"fake-quant expects at most 4 dimensions, got {x.dim()}")   # :70


def _amax(t, d):
    return torch.max(t) if d is None else torch.max(t, dim=d, keepdim=True)[0]


def _amin(t, d):
    return torch.min(t) if d is None else torch.min(t, dim=d, keepdim=True)[0]


def forward(kind, x, num_bits, layerwise):
    if kind == "sym":
        top = _per_row(lambda t, d: _amax(torch.abs(t), d), x, layerwise)
        s = (2 ** (num_bits - 1) - 1) / (top + 1e-6)
        return torch.round(x * s).div(s + 1e-6)
    lo = _per_row(_amin, x, layerwise)
    alpha = _per_row(_amax, x, layerwise) - lo
    levels = 2 ** num_bits - 1
    return torch.round((x - lo) / (alpha + 1e-8) * levels).div(levels) * (alpha + 1e-8) + lo


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
