"""Public surface of the package."""
from . import ops
from ._lib import FakeQuantLibraryError
from .ops import get_semantics, set_semantics
from .utils_quant import (AsymQuantizer, QuantizeLinear, SymQuantizer, conservative, enable_weight_quant_cache, fuse_low_bit_mean, fuse_qlinear, get_backward_mode, group_siblings, inplace_weight_grad,
                          pair_kv_hooks, pair_operands, quantize_kv, set_backward_mode, share_activation_quant)

__version__ = "0.3.0"
__all__ = ["SymQuantizer", "AsymQuantizer", "QuantizeLinear", "ops", "set_semantics", "get_semantics", "set_backward_mode", "get_backward_mode", "share_activation_quant", "enable_weight_quant_cache", "pair_operands", "quantize_kv", "fuse_qlinear", "fuse_low_bit_mean", "group_siblings", "inplace_weight_grad", "pair_kv_hooks", "conservative",
           "FakeQuantLibraryError"]
