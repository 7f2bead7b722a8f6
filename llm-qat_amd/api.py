"""Public surface of the package."""
from . import ops
from .cpu_tensors import allow_cpu_tensors
from ._lib import FakeQuantLibraryError
from .ops import get_semantics, set_semantics
from .utils_quant import (AsymQuantizer, QuantizeLinear, SymQuantizer, conservative, cpp_node, enable_weight_quant_cache, fuse_low_bit_mean, get_backward_mode, host_node,
                          inplace_weight_grad, pair_kv_hooks, pair_operands, quantize_kv, reset_learned_state, set_backward_mode, share_activation_quant, stats)

__version__ = "0.5.0"
__all__ = ["SymQuantizer", "AsymQuantizer", "QuantizeLinear", "ops", "set_semantics", "get_semantics", "set_backward_mode", "get_backward_mode",
           "share_activation_quant", "enable_weight_quant_cache", "pair_operands", "quantize_kv", "fuse_low_bit_mean", "inplace_weight_grad", "pair_kv_hooks",
           "conservative", "cpp_node", "host_node", "stats", "reset_learned_state", "allow_cpu_tensors", "FakeQuantLibraryError"]
