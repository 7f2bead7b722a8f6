"""torch.compile entry of the drop-in (reachable in LLM-QAT through HF's `--torch_compile`, utils/kd_trainer.py:281-286).

The eager path launches the kernels through ctypes from inside `torch.autograd.Function`s, which Dynamo cannot trace
(every quantizer call is a graph break).  Here the same launches are registered as `torch.library` custom ops: opaque
to Dynamo, with a fake-tensor shape rule and an autograd formula, so a compiled `QuantizeLinear` / attention block is
ONE graph.  `utils_quant` routes here while `torch.compiler.is_compiling()`; results are the eager path's, bit for bit
(same kernels).  Data flow of the compiled path is the reference's (the backward re-reads the saved input, :45 / :83):
what a compiled graph keeps alive is planned by the partitioner, so the STE-mask / sharing / pairing machinery of the
eager path is not replicated here.
"""
import torch

from . import ops

_KINDS = {"sym": 0, "asym": 1}


# mode: 0 = arithmetic in the tensor's own dtype; 1 = autocast arithmetic, result rounded once to the tensor dtype
# (QuantizeLinear's operands); 2 = autocast arithmetic, fp32 result (what the reference returns under autocast)
@torch.library.custom_op("llmqat_amd::fake_quant", mutates_args=(), device_types="cuda")
def fake_quant_op(x: torch.Tensor, clip: torch.Tensor, kind: int, num_bits: int, layerwise: bool, mode: int) -> torch.Tensor:
    if mode:
        return ops.sym_forward_autocast(x, num_bits, layerwise, wide=mode == 2)[0]
    return ops.sym_quantize(x, num_bits, layerwise) if kind == 0 else ops.asym_quantize(x, num_bits, layerwise)


@fake_quant_op.register_fake
def _(x, clip, kind, num_bits, layerwise, mode):
    if x.dim() > 4:
        raise ValueError(f"fake-quant expects at most 4 dimensions, got {x.dim()}")  # utils_quant.py:70
    return torch.empty_like(x, dtype=torch.float32 if mode == 2 else x.dtype)


@torch.library.custom_op("llmqat_amd::fake_quant_bwd", mutates_args=(), device_types="cuda")
def fake_quant_bwd(grad_output: torch.Tensor, x: torch.Tensor, clip: torch.Tensor) -> torch.Tensor:
    lo, hi = clip.tolist()[:2] if clip.dim() else (clip.item(),) * 2
    g = grad_output if grad_output.dtype == x.dtype else grad_output.to(x.dtype)  # autocast: fp32 gradient of the fp32 result
    return ops.ste_backward(g, x, float(lo), float(hi))


@fake_quant_bwd.register_fake
def _(grad_output, x, clip):
    return torch.empty_like(grad_output, dtype=x.dtype)


def _setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])  # reference :45 / :104: (input, clip_val)


def _backward(ctx, grad_output):
    x, clip = ctx.saved_tensors
    return fake_quant_bwd(grad_output, x, clip), None, None, None, None, None


fake_quant_op.register_autograd(_backward, setup_context=_setup)


@torch.library.custom_op("llmqat_amd::low_bit_weight", mutates_args=(), device_types="cuda")
def low_bit_weight_op(w: torch.Tensor, scale: torch.Tensor, w_bits: int) -> torch.Tensor:
    return ops.low_bit_weight(w, scale, w_bits)


@low_bit_weight_op.register_fake
def _(w, scale, w_bits):
    return torch.empty_like(w, memory_format=torch.contiguous_format)


low_bit_weight_op.register_autograd(lambda ctx, g: (g, None, None))  # the detach trick: identity gradient (:240-242)


def fake_quant(kind, x, clip_val, num_bits, layerwise, narrow=False):
    """SymQuantizer / AsymQuantizer .apply while compiling."""
    mode = 0
    if kind == "sym" and ops.autocast_active(x):
        mode = 1 if (narrow and ops.autocast_narrow_ok(x)) else 2
    return fake_quant_op(x, clip_val, _KINDS[kind], int(num_bits), bool(layerwise), mode)
