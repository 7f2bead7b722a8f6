"""CUDA autocast's cast policy, applied to CPU tensors -- TEST INFRASTRUCTURE ONLY.

LLM-QAT trains under `torch.autocast("cuda", bf16)` (run_train.sh:17-18 `--bf16 True` -> utils/kd_trainer.py:106).  On the
device that changes the arithmetic of `SymQuantizer.forward` (models/utils_quant.py:71-72): `reciprocal` -- the
`int / Tensor` of :71 -- is on autocast's fp32 list, so everything behind it runs in fp32 and the result is fp32.  CPU autocast
has a different op list, and the reference file cannot travel to the GPU box, so the only way to pin that arithmetic to the
reference's OWN code is to run that code here, on CPU tensors, with the device's cast policy imposed from outside:

  AutocastCasts   (TorchDispatchMode: sees every ATen op the reference executes, below autograd -- where the real autocast
                  dispatch key rewrites `reciprocal`, inside SymQuantizer.forward's no-grad region)
      aten.reciprocal(16-bit tensor)         -> input cast to fp32 first              [autocast "fp32" list]
      any OTHER op on CUDA autocast's lists  -> raises, unless its inputs already have the dtype the policy would give
                                               them (so an incomplete emulation fails loudly instead of producing fixtures)
      device_scalars=True additionally models what ATen's GPU kernels do with a Python scalar ADDED to a 16-bit tensor:
          add(16-bit tensor, python float)   -> fp32 add of the scalar as an fp32 "opmath" value, one rounding
        (on CPU the scalar is first rounded to the tensor dtype; the two differ only for rows whose |max| is below ~4e-5:
        the library's `sem` knob, DESIGN.md "Numerics".  device_scalars=False leaves the reference's CPU behaviour alone.)
      it also keeps the outputs of `aten.round` / the first `aten.mul` behind `reciprocal`: the reference's own bin indices
      and scale, not a restatement.
  LinearCast      (TorchFunctionMode: above autograd, like the Autocast dispatch key, so the casts are recorded by autograd
                  and the gradients flow back through them exactly as under torch.autocast)
      F.linear(input, weight, bias)          -> floating tensors cast to the autocast dtype  [autocast "lower precision" list]
      and keeps the operands as handed to the GEMM.

The GPU tier validates the emulation itself: the live `torch.autocast("cuda")` ATen chain on the MI355X must reproduce the same
fixtures bit for bit (tests/test_gpu_autocast_golden.py).
"""
import torch
import torch.nn.functional as F
from torch.overrides import TorchFunctionMode
from torch.utils._python_dispatch import TorchDispatchMode

aten = torch.ops.aten
_16 = (torch.bfloat16, torch.float16)

# CUDA autocast's op lists (aten/src/ATen/autocast_mode.cpp of torch 2.x), by base name -- used as a tripwire only
FP32_OPS = {"acos", "asin", "cosh", "erfinv", "exp", "expm1", "log", "log10", "log2", "log1p", "reciprocal", "rsqrt", "sinh", "tan",
            "pow", "softplus", "layer_norm", "native_layer_norm", "group_norm", "frobenius_norm", "nuclear_norm", "cosine_similarity",
            "poisson_nll_loss", "cosine_embedding_loss", "nll_loss", "nll_loss2d", "hinge_embedding_loss", "kl_div", "l1_loss",
            "smooth_l1_loss", "huber_loss", "mse_loss", "margin_ranking_loss", "multilabel_margin_loss", "soft_margin_loss",
            "triplet_margin_loss", "multi_margin_loss", "binary_cross_entropy_with_logits", "dist", "pdist", "cdist", "renorm",
            "logsumexp", "prod", "softmax", "log_softmax", "cumprod", "cumsum", "sum", "linalg_vector_norm", "linalg_matrix_norm", "norm"}
LOWER_OPS = {"_convolution", "conv1d", "conv2d", "conv3d", "conv_tbc", "conv_transpose1d", "conv_transpose2d", "conv_transpose3d",
             "convolution", "prelu", "addmm", "addmv", "addr", "matmul", "einsum", "mm", "mv", "linalg_vecdot", "linear", "addbmm",
             "baddbmm", "bmm", "chain_matmul", "linalg_multi_dot", "lstm_cell", "gru_cell", "rnn_tanh_cell", "rnn_relu_cell",
             "scaled_dot_product_attention"}
PROMOTE_OPS = {"addcdiv", "addcmul", "atan2", "bilinear", "cross", "dot", "grid_sampler", "index_put", "tensordot", "scatter_add", "cat", "stack"}


def _tensors(args, kwargs):
    out = []
    for a in list(args) + list(kwargs.values()):
        if isinstance(a, torch.Tensor):
            out.append(a)
        elif isinstance(a, (list, tuple)):
            out.extend(t for t in a if isinstance(t, torch.Tensor))
    return out


class AutocastCasts(TorchDispatchMode):
    def __init__(self, autocast_dtype, device_scalars):
        super().__init__()
        self.autocast_dtype, self.device_scalars = autocast_dtype, device_scalars
        self.rounds, self.scales, self._after_recip = [], [], False
        self.cast_ops = []   # which listed ops were met (the generator records them in the manifest)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        pkt = getattr(func, "overloadpacket", None)
        name = pkt.__name__ if pkt is not None else str(func)
        if name == "reciprocal":
            self.cast_ops.append("reciprocal")
            x = args[0]
            out = func(x.float() if x.dtype in _16 else x)
            self._after_recip = True
            return out
        if name in FP32_OPS or name in PROMOTE_OPS:
            fl = [t for t in _tensors(args, kwargs) if t.is_floating_point()]
            if name in FP32_OPS and any(t.dtype in _16 for t in fl):
                raise RuntimeError(f"autocast policy emulation is incomplete: {func} (fp32 list) met a 16-bit tensor")
            if name in PROMOTE_OPS and len({t.dtype for t in fl}) > 1:
                raise RuntimeError(f"autocast policy emulation is incomplete: {func} (promote list) met mixed dtypes")
        if name in LOWER_OPS:
            if any(t.is_floating_point() and t.dtype != self.autocast_dtype for t in _tensors(args, kwargs)):
                raise RuntimeError(f"autocast policy emulation is incomplete: {func} (lower-precision list) reached ATen uncast")
            self.cast_ops.append(name)
        if (self.device_scalars and func in (aten.add.Tensor, aten.add.Scalar) and isinstance(args[0], torch.Tensor) and args[0].dtype in _16
                and isinstance(args[1], float) and kwargs.get("alpha", 1) == 1):
            return (args[0].float() + args[1]).to(args[0].dtype)
        out = func(*args, **kwargs)
        if name == "round":
            self.rounds.append(out)
        elif name == "mul" and self._after_recip:
            self.scales.append(out)   # reciprocal() * (2**(b-1)-1): the scale s of utils_quant.py:71
            self._after_recip = False
        return out


class LinearCast(TorchFunctionMode):
    def __init__(self, autocast_dtype):
        super().__init__()
        self.autocast_dtype, self.operands = autocast_dtype, []

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is F.linear:
            cast = [a.to(self.autocast_dtype) if isinstance(a, torch.Tensor) and a.is_floating_point() else a for a in args]
            self.operands.append((cast[0].detach(), cast[1].detach()))
            return func(*cast, **kwargs)
        return func(*args, **kwargs)


class cuda_autocast_policy:
    """`with cuda_autocast_policy(torch.bfloat16, device_scalars) as p:` -- both modes; p.casts / p.linear hold what they kept"""

    def __init__(self, autocast_dtype, device_scalars):
        self.casts, self.linear = AutocastCasts(autocast_dtype, device_scalars), LinearCast(autocast_dtype)

    def __enter__(self):
        self.linear.__enter__()
        self.casts.__enter__()
        return self

    def __exit__(self, *exc):
        self.casts.__exit__(*exc)
        self.linear.__exit__(*exc)
