"""Loader of the C++ autograd node (csrc/fq_autograd_node.cpp -> _fq_node.so, built by build.py::build_node).

The node is host code only: it replaces the Python `_PairNode` of utils_quant.py on QuantizeLinear's straight-line path so that the
backward does not run Python on the autograd engine's thread.  The kernels it launches are the same C-ABI entry points of
libllmqat_fakequant.so (handed over as function pointers in `bind`); results are the same bits either way.  Without the file -- or with
LLMQAT_AMD_CPP_NODE=0 -- utils_quant uses its Python node and says so in `llm_qat_amd.stats()["cpp_node"]`-style status (`status()`).
"""
import ctypes
import importlib.util
import os

from . import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
# LLMQAT_AMD_NODE points the loader at another build of the node (a sanitizer build: tests/README.md), as LLMQAT_AMD_LIB does for the kernels
NODE_PATH = os.environ.get("LLMQAT_AMD_NODE") or os.path.join(HERE, "_fq_node.so")

_mod = None
_why = None
_bound = False


def load():
    """-> the extension module or None (status() says why)"""
    global _mod, _why
    if _mod is not None or _why is not None:
        return _mod
    if os.environ.get("LLMQAT_AMD_CPP_NODE", "1") == "0":
        _why = "switched off (LLMQAT_AMD_CPP_NODE=0)"
        return None
    if not os.path.exists(NODE_PATH):
        _why = "not built (python llm-qat_amd/build.py)"
        return None
    try:
        import torch
        built_for = open(NODE_PATH + ".built_for").read().strip() if os.path.exists(NODE_PATH + ".built_for") else "?"
        if built_for != torch.__version__:      # a C++ extension is tied to the PyTorch it was compiled against: never load it under another one
            raise RuntimeError(f"built for PyTorch {built_for}, this is {torch.__version__}: run python llm-qat_amd/build.py")
        spec = importlib.util.spec_from_file_location("_fq_node", NODE_PATH)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        if mod.abi_version != _lib.ABI_VERSION:
            raise RuntimeError(f"built for C ABI {mod.abi_version}, the package speaks {_lib.ABI_VERSION}")
    except Exception as e:  # noqa: BLE001 -- another torch build than the one it was compiled against, a stale file: the Python node serves
        _why = f"failed to load: {e!r}"
        return None
    _mod = mod
    return mod


def bind(slow_backward, slow_backward_one, forget):
    """hand the kernel library's entry points (addresses out of the ctypes binding) and utils_quant's general backward to the node"""
    global _bound
    if _bound or _mod is None:
        return _bound
    lib = _lib.lib()
    addr = lambda name: ctypes.cast(lib[name], ctypes.c_void_p).value  # noqa: E731  (lib[name]: the symbol itself, whatever the attribute holds)
    _mod.bind(addr("fq_sym_fwd_pair"), addr("fq_ste_bwd_mask_pair"), addr("fq_sym_fwd_multi"), addr("fq_ste_bwd_mask"), addr("fq_ste_bwd_mask_wide"),
              addr("fq_last_error"), slow_backward, slow_backward_one, forget)
    _bound = True
    return True


def status():
    return "loaded" if _mod is not None else (_why or "not loaded yet")
