"""ATen's GPU treatment of PYTHON SCALARS, applied to CPU tensors -- TEST INFRASTRUCTURE ONLY.

Outside autocast the reference's op chain (models/utils_quant.py:71-72, :144-147) meets Python scalars in two places where ATen's CPU and
GPU kernels do not compute the same thing:

    `max_input + 1e-6`, `s + 1e-6`, `alpha + 1e-8`   add(Tensor, python float)
        CPU: the scalar is first rounded to the tensor's dtype, then a dtype add       (16-bit tensors only differ: 1e-6 -> 9.98e-07 in bf16)
        GPU: the scalar stays an fp32 "opmath" value: fp32 add, ONE rounding to the tensor dtype
    `.div(s)` with s = 2**bits - 1                   div(Tensor, python int)
        CPU: a true division by the scalar
        GPU: multiply by the fp32 reciprocal of the scalar (`a * (1 / b)`, ATen's div_true_kernel_cuda for a CPU-scalar divisor),
             rounded once to the tensor dtype

Nothing else on the path differs (`int / Tensor` is reciprocal() * int on both; mul by a scalar keeps the scalar in opmath on both;
tensor (op) tensor is the same arithmetic).  The reference file cannot travel to the GPU box, so -- exactly as tests/autocast_policy.py does
for autocast's casts -- the reference's OWN code is run here on CPU tensors with those two rules imposed from outside (a TorchDispatchMode:
it sees every ATen op the reference executes), and tests/golden/make_golden_device_scalars.py records what it produced.  The GPU tier closes
the loop: the live, non-autocast ATen chain on the MI355X must reproduce the same fixtures bit for bit
(tests/test_gpu_device_scalars.py::test_live_aten_reproduces_the_fixture), which validates the imposed policy itself.
"""
import numpy as np
import torch
from torch.utils._python_dispatch import TorchDispatchMode

aten = torch.ops.aten
_16 = (torch.bfloat16, torch.float16)


class DeviceScalars(TorchDispatchMode):
    """also keeps the outputs of every `aten.round` (the reference's own bin indices) and counts what it rewrote"""

    def __init__(self):
        super().__init__()
        self.rounds, self.rewrote = [], {"add": 0, "div": 0}

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in (aten.add.Tensor, aten.add.Scalar) and isinstance(args[0], torch.Tensor) and isinstance(args[1], (float, int)) \
                and not isinstance(args[1], bool) and kwargs.get("alpha", 1) == 1 and args[0].is_floating_point():
            x = args[0]
            if x.dtype in _16:
                self.rewrote["add"] += 1
                return (x.float() + float(np.float32(args[1]))).to(x.dtype)   # fp32 opmath scalar, one rounding
            # fp32 / fp64 tensors: the scalar is converted to the tensor's dtype on both devices -- nothing to change
        if func in (aten.div.Tensor, aten.div.Scalar) and isinstance(args[0], torch.Tensor) and isinstance(args[1], (float, int)) \
                and not isinstance(args[1], bool) and args[0].is_floating_point() and not kwargs.get("rounding_mode"):
            x = args[0]
            self.rewrote["div"] += 1
            if x.dtype == torch.float64:
                return x * (1.0 / float(args[1]))
            inv = float(np.float32(1.0) / np.float32(args[1]))                # the reciprocal is taken in fp32 opmath
            return (x.float() * inv).to(x.dtype) if x.dtype in _16 else x * torch.tensor(inv, dtype=torch.float32)
        out = func(*args, **kwargs)
        pkt = getattr(func, "overloadpacket", None)
        if pkt is not None and pkt.__name__ == "round":
            self.rounds.append(out)
        return out
