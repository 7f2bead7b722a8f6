"""CPU tier: the C-ABI library loads and exports every symbol include/*.h declares; host-side logic
(argument validation, shape -> [rows, cols] mapping, module surface) behaves like the reference's.
No kernel is launched here."""
import ctypes
import glob
import os
import re

import pytest
import torch

from conftest import ROOT


def header_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(fq_[a-z0-9_]+)\s*\(", src))
    return names


def test_library_exports_every_declared_symbol():
    from llm_qat_amd import _lib
    L = _lib.lib()
    declared = header_symbols()
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.fq_version() == _lib.ABI_VERSION
    assert b"gfx950" in L.fq_build_info()


def test_header_cites_reference_lines():
    src = open(os.path.join(ROOT, "include", "llmqat_fakequant.h")).read()
    for cite in ("utils_quant.py:37-74", "utils_quant.py:96-149", "utils_quant.py:77-87"):
        assert cite in src


def test_argument_validation_without_gpu():
    """error codes come back before any HIP call; nothing throws across the boundary"""
    from llm_qat_amd import _lib
    L = _lib.lib()
    assert L.fq_sym_fwd(None, None, 4, 4, 4, 9, 0, None, None, 0, None) == -1      # dtype
    assert b"dtype" in L.fq_last_error()
    assert L.fq_sym_fwd(None, None, 4, 4, 0, 1, 0, None, None, 0, None) == -2      # bits (1..31 are served: 1-bit Sym has the single level 0, as in the reference)
    assert L.fq_sym_fwd(None, None, 4, 4, 32, 1, 0, None, None, 0, None) == -2
    assert L.fq_asym_fwd(None, None, 4, 4, 32, 1, 0, None, None, 0, None) == -2
    assert L.fq_sym_fwd(None, None, 4, 4, 4, 1, 5, None, None, 0, None) == -7      # semantics
    assert L.fq_sym_fwd(None, None, -1, 4, 4, 1, 0, None, None, 0, None) == -3     # shape
    assert L.fq_sym_fwd(None, None, 4, 4, 4, 1, 0, None, None, 0, None) == -4      # null
    assert L.fq_sym_fwd(None, None, 0, 4, 4, 1, 0, None, None, 0, None) == 0       # empty: ok, no launch
    assert L.fq_last_error() == b""
    assert L.fq_ste_bwd(None, None, None, 0, -2.0, 2.0, 1, None) == 0
    assert L.fq_ste_bwd(None, None, None, 8, -2.0, 2.0, 1, None) == -4
    assert L.fq_ste_bwd(None, None, None, 8, -2.0, 2.0, 4, None) == -1      # (3 = float64 is served since round 3)
    assert L.fq_ste_bwd_rows(None, None, None, 2, 4, -2.0, 2.0, None, 3, None) == -1   # float64: no training-mode side buffers
    buf = ctypes.create_string_buffer(64)
    p = ctypes.addressof(buf)
    assert L.fq_sym_fwd(p, p, 1, 8, 4, 1, 0, None, None, 0, None) == -7            # in-place rejected
    with pytest.raises(ValueError):
        _lib.check(-2, "x")
    with pytest.raises(RuntimeError):
        _lib.check(-6, "x")
    assert L.fq_rowwise_workspace_bytes(4096, 11008, 1) == 0
    assert L.fq_rowwise_workspace_bytes(1, 4096 * 11008, 1) == 8
    assert L.fq_rowwise_workspace_bytes(3, 40000, 0) == 24
    # STE mask: 1 bit per element, rounded up to whole 64-vector groups; 0 = shape not served
    assert L.fq_ste_mask_bytes(4096, 11008, 1) == 4096 * 172 * 8             # a plain bitmap: ceil(11008 / 64) = 172 words per row
    assert L.fq_ste_mask_bytes(2048, 4096, 1) == 2048 * 4096 // 8
    assert L.fq_ste_mask_bytes(4, 100, 1) == 0 and L.fq_ste_mask_bytes(4, 100, 0) == 4 * 2 * 8    # fp32: 25 vectors, 100 bits -> 2 words
    assert L.fq_ste_mask_bytes(1, 4096 * 11008, 1) == 0
    assert L.fq_sym_fwd_train(p, p + 16, 1, 8, 4, 1, 0, -2.0, 2.0, None, None, 0, None) == -4
    assert L.fq_ste_bwd_mask(p, p + 16, 1, 100, -2.0, 2.0, p, p, 64, 1, None) == -8


def test_rows_cols_mapping_follows_reference_granularity():
    from llm_qat_amd.ops import rows_cols
    assert rows_cols((4096, 11008), False) == (4096, 11008)       # weight: per output channel
    assert rows_cols((1, 2048, 4096), False) == (2048, 4096)      # activation / KV: per token
    assert rows_cols((2, 3, 4, 5), False) == (6, 20)              # 4-D: per (d0, d1)   utils_quant.py:60-68
    assert rows_cols((2, 3, 4, 5), True) == (1, 120)              # layerwise           utils_quant.py:50-51
    assert rows_cols((9,), False) == (1, 9)
    assert rows_cols((), False) == (1, 1)
    with pytest.raises(ValueError):                               # utils_quant.py:70
        rows_cols((1, 2, 3, 4, 5), False)
    from oracle import oracle as O                                # the oracle maps shapes the same way
    for shp in [(7,), (3, 5), (2, 3, 4), (2, 3, 4, 5)]:
        for lw in (False, True):
            assert tuple(rows_cols(shp, lw)) == tuple(O.rows_cols(shp, lw))


def test_module_surface_matches_reference():
    from llm_qat_amd.utils_quant import AsymQuantizer, QuantizeLinear, SymQuantizer
    lin = QuantizeLinear(16, 8, bias=True, w_bits=4, a_bits=8)
    assert lin.bias is None                                       # bias forced off  (:176)
    assert list(lin.state_dict().keys()) == ["weight"]
    assert (lin.w_bits, lin.a_bits, lin.act_layerwise, lin.weight_layerwise) == (4, 8, False, False)
    assert lin.act_quantizer is SymQuantizer
    assert QuantizeLinear(16, 8, symmetric=False, a_bits=8).act_quantizer is AsymQuantizer
    assert not hasattr(QuantizeLinear(16, 8, a_bits=2), "act_quantizer")    # a_bits<=2 disables act quant (:184)
    assert not hasattr(QuantizeLinear(16, 8, a_bits=32), "act_quantizer")
    assert isinstance(lin, torch.nn.Linear)
    for q in (SymQuantizer, AsymQuantizer):
        assert issubclass(q, torch.autograd.Function)
    # w_bits>=32 and a_bits>=32: plain linear, runs anywhere (no kernel involved)
    plain = QuantizeLinear(16, 8)
    x = torch.randn(3, 16)
    torch.testing.assert_close(plain(x), torch.nn.functional.linear(x, plain.weight))


def test_cpu_tensors_fail_loudly_no_fallback():
    """the default: CPU tensors are refused (the opt-in torch-op path for them is tests/test_cpu_tensors.py)"""
    from llm_qat_amd.utils_quant import QuantizeLinear, SymQuantizer
    with pytest.raises(RuntimeError, match="no CPU"):
        SymQuantizer.apply(torch.randn(4, 8), torch.tensor([-2.0, 2.0]), 8, False)
    with pytest.raises(RuntimeError, match="no CPU"):
        QuantizeLinear(8, 4, w_bits=4, a_bits=8)(torch.randn(2, 8))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from llm_qat_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.FakeQuantLibraryError):
        _lib.lib()


def test_num_bits_must_mean_what_it_means_in_the_reference():
    """The reference turns num_bits into `2 ** (num_bits - 1) - 1` in Python and divides that int by a tensor (reciprocal * int, :71): an
    integral float is the same thing; a fractional float (fractional levels) or a tensor (true Tensor / Tensor division: other roundings)
    would silently compute something else if truncated -- refused loudly, before any device work"""
    import torch
    from llm_qat_amd import ops
    from llm_qat_amd.utils_quant import AsymQuantizer, SymQuantizer, quantize_kv
    assert ops.bits_arg(8) == 8 and ops.bits_arg(8.0) == 8 and ops.bits_arg(True) == 1
    clip, x = torch.tensor([-2.0, 2.0]), torch.zeros(2, 8)
    for q in (SymQuantizer, AsymQuantizer):
        with pytest.raises(ValueError, match="integral"):
            q.apply(x, clip, 7.5, False)
        with pytest.raises(TypeError, match="Python int"):
            q.apply(x, clip, torch.tensor(8), False)
    with pytest.raises(TypeError, match="Python int"):
        quantize_kv(x, x, clip, clip, torch.tensor(4))


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under llm-qat_amd/ may reference it"""
    pkg = os.path.join(ROOT, "llm-qat_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                for pat in (r"import\s+oracle", r"from\s+oracle", r"oracle[/.]", r"libfq_oracle", r"\bfqo_"):
                    assert not re.search(pat, src), (os.path.join(dirpath, f), pat)


def test_product_never_reaches_into_tools():
    """tools/ holds experiments and measurements (the quantize-on-load GEMM, the int8 consumer of the export format): the package
    neither imports them nor loads their libraries"""
    pkg = os.path.join(ROOT, "llm-qat_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                for pat in (r"libfq_qlinear_exp", r"libfq_int8_epilogue", r"int8_linear", r"tools[/.](qlinear|int8_linear)"):
                    assert not re.search(pat, src), (os.path.join(dirpath, f), pat)


def test_oracle_is_only_reached_from_the_allowed_places():
    """Besides tests/: __graft_entry__ (build() compiles it, smoke() checks against it) and bench.py's cpu_baseline leg
    (cpu_baseline, cpu_c_port, parity_gate) -- nowhere else: no tool, no GPU leg of the benchmark."""
    import ast
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tools")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), os.path.join(dirpath, f)
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    allowed = {"cpu_baseline", "cpu_c_port", "parity_gate"}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef):
            for sub in ast.walk(node):
                if isinstance(sub, ast.ImportFrom) and (sub.module or "").split(".")[0] == "oracle":
                    assert node.name in allowed, f"bench.py:{node.name} imports oracle"
                if isinstance(sub, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in sub.names):
                    assert node.name in allowed, f"bench.py:{node.name} imports oracle"
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    assert not any((getattr(n, "module", None) or "").startswith("oracle") or any(a.name.startswith("oracle") for a in n.names) for n in top)


def test_reciprocal_multiply_with_correction_equals_ieee_division(tmp_path):
    """The kernels replace per-row IEEE divisions (fp16 chains, AsymQuantizer, the autocast path) by a reciprocal
    multiply + Markstein correction (fq_device.h: div_exact).  tests/c_host/markstein_check.c replays that sequence on
    the host CPU against `/` over 1e8 quotients spanning the ranges the kernels admit: zero mismatches."""
    import subprocess
    exe = str(tmp_path / "markstein_check")
    subprocess.run(["gcc", "-O2", "-mfma", "-o", exe, os.path.join(ROOT, "tests", "c_host", "markstein_check.c"), "-lm"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and " 0 mismatches" in out.stdout, out.stdout[-2000:]


# ---- round 2 host logic (no kernel is launched) ------------------------------------------------------------------------
def test_export_container_rules_and_host_side_unpacking():
    """ops.default_container picks the smallest LOSSLESS container (8-bit bf16 needs int16: the reference has no clamp and
    reaches bin +128); QuantExport.unpacked / .dequantize (pure torch host code) reproduce the reference's `idx` / `y`
    fixtures when fed the oracle's packed bytes -- on CPU tensors, no GPU involved."""
    import numpy as np
    import torch
    from conftest import golden, bits_equal
    from oracle import oracle as O
    import llm_qat_amd
    ops = llm_qat_amd.ops
    assert ops.default_container("sym", 4, torch.bfloat16) == "int4" and ops.default_container("sym", 3, torch.float32) == "int4"
    assert ops.default_container("sym", 8, torch.bfloat16) == "int16" and ops.default_container("sym", 8, torch.float16) == "int8"
    assert ops.default_container("sym", 7, torch.bfloat16) == "int8" and ops.default_container("sym", 16, torch.float32) == "int16"
    assert ops.default_container("asym", 4, torch.bfloat16) == "int4" and ops.default_container("asym", 8, torch.bfloat16) == "int8"
    TD = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}
    n = 0
    for kind in ("sym", "asym"):
        G = golden(f"{kind}_fwd.npz")
        for c in G.cases[::3]:
            dt, bits = c["dtype"], c["bits"]
            rows, cols = O.rows_cols(c["shape"], c["layerwise"])
            for container in ("int4", "int8", "int16"):
                ob, osc, oov = O.export(kind, G.arr(c, "x"), rows, cols, bits, container, dt)
                raw = torch.from_numpy(ob.copy())
                if container == "int8":
                    bins = (raw.view(torch.int8) if kind == "sym" else raw).view(rows, cols)
                elif container == "int16":
                    bins = raw.view(torch.int16).view(rows, cols)
                else:
                    bins = raw
                e = ops.QuantExport(kind=kind, bins=bins, scales=torch.from_numpy(osc.copy()), overflow=torch.from_numpy(oov.copy()), container=container,
                                    num_bits=bits, shape=(rows, cols), rows=rows, cols=cols, dtype=TD[dt])
                assert (e.unpacked().numpy() == O.unpack_bins(ob, cols, container, kind == "sym")).all(), (c["name"], container)
                ok = oov == 0
                if not ok.any():
                    continue
                y = e.dequantize()
                got = y.numpy() if dt == "fp32" else y.view(torch.int16).numpy().view(np.uint16)
                want = G.arr(c, "y").reshape(rows, cols)
                zero = (lambda a: np.where(a == 0, 0, a)) if dt == "fp32" else (lambda a: np.where((a & 0x7FFF) == 0, 0, a))
                assert bits_equal(zero(got[ok]), zero(want[ok]), dt), (c["name"], container)
                n += 1
    assert n > 100


def test_round2_entry_points_validate_arguments_without_a_gpu():
    import ctypes
    from llm_qat_amd import _lib
    L = _lib.lib()
    assert L.fq_export_bins_bytes(4, 63, _lib.BINS_INT4) == 4 * 32 and L.fq_export_bins_bytes(0, 63, _lib.BINS_INT4) == 0
    assert L.fq_export_bins_bytes(4, 64, 7) == 0
    bad = ctypes.c_void_p(16)
    assert L.fq_sym_export(bad, None, None, None, 4, 64, 8, _lib.BINS_INT8, 1, 0, 0, None) == -4          # bins NULL
    assert L.fq_sym_export(bad, bad, None, None, 4, 64, 8, _lib.BINS_NONE, 1, 0, 0, None) == -7           # needs a container
    assert L.fq_sym_export(bad, bad, None, None, 4, 64, 40, _lib.BINS_INT8, 1, 0, 0, None) == -2          # bits
    assert L.fq_sym_export(bad, bad, None, None, 4, 64, 8, _lib.BINS_INT8, 0, 0, 1, None) == -1           # autocast on fp32
    assert L.fq_asym_export(bad, bad, None, None, 0, 64, 8, _lib.BINS_INT8, 1, 0, None) == 0              # empty: nothing to do
    assert L.fq_sym_row_scales(bad, None, 4, 64, 8, 1, 0, 0, -2.0, 2.0, None, None, 0, None) == -4        # nothing to produce
    assert L.fq_sym_fwd_autocast(bad, ctypes.c_void_p(32), 4, 64, 8, 1, 7, 1, -2.0, 2.0, None, None, 0, None, 0, None) == -7   # unknown sem (ABI 4)
    assert L.fq_sym_fwd_autocast(bad, ctypes.c_void_p(32), 4, 64, 8, 0, 1, 1, -2.0, 2.0, None, None, 0, None, 0, None) == -1   # fp32: not an autocast tensor
    with pytest.raises(AttributeError):   # ABI 4: the fused-GEMM experiment (test hooks in its signature) left the product library
        L.fq_qlinear_fwd
    t = (_lib.FwdTensor * 5)()
    assert L.fq_sym_fwd_multi(5, t, 64, 1, 0, 0, -2.0, 2.0, None) == -7                                   # at most 4 tensors
    assert L.fq_sym_fwd_multi(2, t, 64, 1, 0, 0, -2.0, 2.0, None) in (-2, -3, -4)                         # zeroed descriptors are rejected
    assert L.fq_w12_fwd_rows(bad, bad, None, 4, 64, 1, 1, None) == -8        # rows < 8 / cols < 256: ATen's reduce configuration is another one there
    assert L.fq_w12_fwd_rows(bad, bad, None, 4, 64, 3, 1, None) == -2
    assert b"bits" in L.fq_last_error() or b"w_bits" in L.fq_last_error()


def test_conservative_switch_turns_every_stateful_host_optimisation_off_and_back_on():
    """llm_qat_amd.conservative(True): one launch + one autograd node per reference call, nothing remembered between calls
    (the GPU tier checks that results stay bit-identical: tests/test_tiny_llama.py::test_conservative_mode_is_bit_identical)"""
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    names = ("_PAIR", "_SHARE_ACT", "_PAIR_KV", "_INPLACE_WGRAD", "_WEIGHT_CACHE")
    before = {n: getattr(UQ, n) for n in names}
    try:
        llm_qat_amd.enable_weight_quant_cache(True)
        llm_qat_amd.conservative(True)
        assert not any(getattr(UQ, n) for n in names), {n: getattr(UQ, n) for n in names}
        llm_qat_amd.conservative(False)
        assert UQ._PAIR and UQ._SHARE_ACT and UQ._PAIR_KV and UQ._INPLACE_WGRAD      # the defaults
        assert not UQ._WEIGHT_CACHE   # opt-ins stay off
    finally:
        UQ._PAIR, UQ._SHARE_ACT, UQ._PAIR_KV, UQ._INPLACE_WGRAD = before["_PAIR"], before["_SHARE_ACT"], before["_PAIR_KV"], before["_INPLACE_WGRAD"]
        llm_qat_amd.enable_weight_quant_cache(before["_WEIGHT_CACHE"])


def test_abi_rejects_null_pointers_and_hostile_sizes_before_any_launch():
    """Fuzz of every exported entry point with NULL pointers and hostile integers / floats (no GPU needed: validation comes first): each
    call RETURNS -- a validation code, or 0 for an empty shape -- and none gets as far as a launch (FQ_ERR_LAUNCH), i.e. a NULL can never
    reach a kernel."""
    import ctypes
    import random
    from llm_qat_amd import _lib
    L = _lib.lib()
    rng = random.Random(0)
    ints = [-(2 ** 63), -(2 ** 31) - 1, -2, -1, 0, 1, 2, 3, 4, 7, 8, 16, 31, 32, 33, 64, 255, 256, 4096, 11008, 2 ** 31 - 1, 2 ** 31, 2 ** 32, 2 ** 40, 2 ** 62]
    floats = [0.0, -0.0, 1.0, -2.0, 2.0, float("inf"), float("-inf"), float("nan"), 1e-45, 3e38]
    launching = [n for n in _lib.EXPORTS if n not in ("fq_version", "fq_build_info", "fq_last_error", "fq_rowwise_workspace_bytes", "fq_ste_mask_bytes",
                                                      "fq_export_bins_bytes")]
    calls = 0
    for name in launching:
        f = getattr(L, name)
        for _ in range(400):
            args = []
            for t in f.argtypes:
                if t is ctypes.c_void_p:
                    args.append(None)
                elif t is ctypes.c_float:
                    args.append(rng.choice(floats))
                elif t is ctypes.c_int64:
                    args.append(rng.choice(ints))
                elif t is ctypes.c_size_t:
                    args.append(rng.choice([v for v in ints if v >= 0]))
                elif t is ctypes.c_int:
                    args.append(max(-(2 ** 31), min(2 ** 31 - 1, rng.choice(ints))))
                else:   # POINTER(struct) of the multi-tensor entry points: NULL, or a table of zeroed slots
                    args.append(None if rng.random() < 0.5 else (t._type_ * _lib.MAX_TENSORS)())
            rc = f(*args)
            calls += 1
            assert rc in (0, -1, -2, -3, -4, -5, -7, -8), f"{name}{tuple(args)} -> {rc}: {L.fq_last_error().decode(errors='replace')}"
    assert calls == 400 * len(launching)
