"""GPU tier (-m gpu): the HIP kernels, called through the C ABI, against
  (1) the golden vectors the real reference produced (tests/golden),
  (2) the CPU oracle on seeded inputs at sizes the oracle finishes in seconds,
  (3) size-independent properties at BASELINE.json's full sizes.
Bar: bin indices bit-exact; dequantized values bit-exact too (the north star allows 1e-6 rel;
the tolerance used here is ZERO, except that any NaN equals any NaN); gradients bit-exact.
"""
import numpy as np
import pytest
import torch

from conftest import bits_equal, golden, mismatch_report, to_f32
from oracle import oracle as O

pytestmark = pytest.mark.gpu

TD = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def dev_from(a, dtype):
    if dtype == "fp32":
        return torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).cuda().view(TD[dtype])


def np_from(t):
    t = t.detach().contiguous().cpu()
    return t.numpy() if t.dtype in (torch.float32, torch.int32) else t.view(torch.int16).numpy().view(np.uint16)


@pytest.fixture(scope="module")
def ops():
    import llm_qat_amd
    from llm_qat_amd import _lib
    _lib.lib()  # fail loudly if the HIP library is absent
    llm_qat_amd.set_semantics("cpu_eager")
    return llm_qat_amd.ops


# ------------------------------------------------------------------------------------------
# (1) golden fixtures
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["sym", "asym"])
def test_forward_golden(ops, kind):
    G = golden(f"{kind}_fwd.npz")
    fn = ops.sym_quantize_debug if kind == "sym" else ops.asym_quantize_debug
    failures = []
    for c in G.cases:
        dt = c["dtype"]
        x = dev_from(G.arr(c, "x"), dt)
        y, idx, scale = fn(x, c["bits"], c["layerwise"])
        torch.cuda.synchronize()
        y_np, idx_np = np_from(y), np_from(idx)
        if not (idx_np == G.arr(c, "idx")).all():
            failures.append(f"{c['name']}: idx {int((idx_np != G.arr(c, 'idx')).sum())} differ")
        elif not bits_equal(y_np, G.arr(c, "y"), dt):
            failures.append(f"{c['name']}: y {mismatch_report(y_np, G.arr(c, 'y'), dt)}")
        if kind == "sym":
            if not bits_equal(np_from(scale), to_f32(G.arr(c, "scale"), dt), "fp32"):
                failures.append(f"{c['name']}: scale")
        # the product entry point (no debug outputs) must give the same y
        y2 = (ops.sym_quantize if kind == "sym" else ops.asym_quantize)(x, c["bits"], c["layerwise"])
        if not bits_equal(np_from(y2), y_np, dt):
            failures.append(f"{c['name']}: product path != debug path")
    assert not failures, "\n".join(failures[:20])


def test_ste_backward_golden(ops):
    G = golden("ste_bwd.npz")
    for c in G.cases:
        dt = c["dtype"]
        clip = G.arr(c, "clip")
        gx = ops.ste_backward(dev_from(G.arr(c, "g"), dt), dev_from(G.arr(c, "x"), dt), float(clip[0]), float(clip[1]))
        got, want = np_from(gx), G.arr(c, "gx")
        a, b = (got.view(np.uint32), want.view(np.uint32)) if got.dtype == np.float32 else (got, want)
        assert (a == b).all(), c["name"]


@pytest.mark.parametrize("mode", ["mask", "bounds", "plain"])
def test_autograd_functions_golden(ops, mode):
    """SymQuantizer / AsymQuantizer .apply + .backward through autograd, CPU clip tensor as in the reference;
    all three backward data flows give the reference's gradient bit for bit."""
    import llm_qat_amd
    from llm_qat_amd.utils_quant import AsymQuantizer, SymQuantizer
    G = golden("ste_bwd.npz")
    prev = llm_qat_amd.get_backward_mode()
    llm_qat_amd.set_backward_mode(mode)
    try:
        _autograd_golden(G, SymQuantizer, AsymQuantizer)
    finally:
        llm_qat_amd.set_backward_mode(prev)


def _autograd_golden(G, SymQuantizer, AsymQuantizer):
    for c in G.cases:
        dt = c["dtype"]
        quant = SymQuantizer if c["quant"] == "SymQuantizer" else AsymQuantizer
        x = dev_from(G.arr(c, "x"), dt).requires_grad_(True)
        y = quant.apply(x, torch.from_numpy(G.arr(c, "clip")), c["bits"], False)
        assert y.dtype == x.dtype and y.shape == x.shape and y.device == x.device
        y.backward(dev_from(G.arr(c, "g"), dt))
        got, want = np_from(x.grad), G.arr(c, "gx")
        a, b = (got.view(np.uint32), want.view(np.uint32)) if got.dtype == np.float32 else (got, want)
        assert (a == b).all(), c["name"]


def test_quantize_linear_golden(ops):
    """QuantizeLinear drop-in: same ctor, state_dict == ['weight'], outputs/grads match the reference's.
    The GEMM (F.linear) is rocBLAS on the device vs MKL on the CPU, so out/grads are compared with a
    tolerance; the OPERANDS the module hands to the GEMM are compared bit for bit with the reference's own
    (`opx` / `opw` in the fixture, round 4)."""
    import torch.nn.functional as F
    from llm_qat_amd.utils_quant import QuantizeLinear
    G = golden("quantize_linear.npz")
    real, seen = F.linear, {}

    def spy(inp, weight, bias=None):
        seen["x"], seen["w"] = inp.detach().clone(), weight.detach().clone()
        return real(inp, weight, bias)

    for c in G.cases:
        dt = c["dtype"]
        kw = {k: c[k] for k in ("w_bits", "a_bits", "symmetric", "act_layerwise", "weight_layerwise") if k in c}
        lin = QuantizeLinear(c["in_features"], c["out_features"], bias=True, **kw)
        assert lin.bias is None and list(lin.state_dict().keys()) == ["weight"]
        lin = lin.cuda()
        lin.weight.data = dev_from(G.arr(c, "w"), dt)
        x = dev_from(G.arr(c, "x"), dt).requires_grad_(True)
        F.linear = torch.nn.functional.linear = spy
        try:
            out = lin(x)
        finally:
            F.linear = torch.nn.functional.linear = real
        if c["w_bits"] >= 3:   # (the 1-/2-bit scale is a float SUM: the device's differs from the CPU fixture's in the last place)
            assert bits_equal(np_from(seen["w"]), G.arr(c, "opw"), dt), f"{c['name']}: weight operand {mismatch_report(np_from(seen['w']), G.arr(c, 'opw'), dt)}"
        assert bits_equal(np_from(seen["x"]), G.arr(c, "opx"), dt), f"{c['name']}: input operand {mismatch_report(np_from(seen['x']), G.arr(c, 'opx'), dt)}"
        out.backward(dev_from(G.arr(c, "go"), dt))
        tol = dict(rtol=2e-2, atol=2e-2) if dt == "bf16" else dict(rtol=1e-4, atol=1e-5)
        for name, got in (("out", out), ("gw", lin.weight.grad), ("gx", x.grad)):
            want = torch.from_numpy(to_f32(G.arr(c, name), dt))
            torch.testing.assert_close(got.float().cpu(), want, msg=lambda m: f"{c['name']} {name}: {m}", **tol)
        # STE mask on the weight gradient is exact: weights beyond the clip get exactly zero
        if 3 <= c["w_bits"] < 32:
            gw = lin.weight.grad.float().cpu()
            assert gw[3, 5] == 0 and gw[4, 6] == 0


# ------------------------------------------------------------------------------------------
# (2) seeded random inputs vs the CPU oracle
# ------------------------------------------------------------------------------------------
SHAPES = [(1, 8), (3, 16), (5, 24), (2, 40), (7, 256), (4, 264), (9, 512), (3, 520), (6, 1024), (2, 2048), (5, 2056),
          (3, 4096), (2, 5120), (2, 11008), (2, 13824), (2, 16384), (1, 16392), (2, 32768), (1, 65536),
          (3, 1), (4, 7), (5, 33), (3, 255), (2, 1001), (2, 4097), (1, 40000), (2, 40001)]


def make_input(rng, shape, dtype, style):
    if style == "weight":
        x = rng.standard_normal(shape).astype(np.float32) * 0.02
    elif style == "act":
        x = rng.standard_normal(shape).astype(np.float32)
        x[rng.random(shape) < 1e-3] *= 20.0
    else:  # mixed row scales
        x = rng.standard_normal(shape).astype(np.float32) * rng.choice([1e-6, 1e-4, 0.02, 1.0, 50.0], size=tuple(shape[:-1]) + (1,)).astype(np.float32)
    t = torch.from_numpy(x).to(TD[dtype])
    return np_from(t), t.cuda()


@pytest.mark.parametrize("dtype", ["bf16", "fp32", "fp16"])
@pytest.mark.parametrize("kind", ["sym", "asym"])
def test_forward_vs_oracle(ops, kind, dtype):
    rng = np.random.default_rng({"sym": 100, "asym": 200}[kind] + {"bf16": 1, "fp32": 2, "fp16": 3}[dtype])
    fn = ops.sym_quantize_debug if kind == "sym" else ops.asym_quantize_debug
    for shape in SHAPES:
        for bits, style in ((4, "weight"), (8, "act"), (3, "mixed"), (16, "mixed")):
            x_np, x = make_input(rng, shape, dtype, style)
            y, idx, _ = fn(x, bits, False)
            if kind == "sym":
                yo, io, _ = O.sym_fwd(x_np, shape[0], shape[1], bits, dtype)
            else:
                yo, io, _, _ = O.asym_fwd(x_np, shape[0], shape[1], bits, dtype)
            assert (np_from(idx) == io).all(), f"{kind} {dtype} {shape} b{bits}: idx"
            assert bits_equal(np_from(y), yo, dtype), f"{kind} {dtype} {shape} b{bits}: {mismatch_report(np_from(y), yo, dtype)}"


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
@pytest.mark.parametrize("kind", ["sym", "asym"])
def test_layerwise_and_4d_vs_oracle(ops, kind, dtype):
    """layerwise = one row (two-pass kernels once the tensor is long); 4-D = (d0*d1) rows."""
    rng = np.random.default_rng(11)
    fn = ops.sym_quantize_debug if kind == "sym" else ops.asym_quantize_debug
    ofn = O.sym_fwd if kind == "sym" else O.asym_fwd
    for shape in [(4, 50), (2, 3, 64), (64, 1024), (96, 4096), (3, 5, 7, 11), (2, 2, 16, 8), (33, 4001)]:
        x_np, x = make_input(rng, shape, dtype, "act")
        for layerwise in (True, False):
            rows, cols = O.rows_cols(shape, layerwise)
            y, idx, _ = fn(x, 8, layerwise)
            res = ofn(x_np, rows, cols, 8, dtype)
            assert (np_from(idx) == res[1]).all(), f"{kind} {dtype} {shape} lw={layerwise}"
            assert bits_equal(np_from(y), res[0], dtype), f"{kind} {dtype} {shape} lw={layerwise}"
    # NaN / Inf inside a long layerwise tensor (two-pass path): NaN poisons everything
    x_np, x = make_input(rng, (64, 4096), dtype, "act")
    x[17, 1234] = float("nan")
    x_np = np_from(x)
    y, idx, _ = fn(x, 8, True)
    res = ofn(x_np, 1, 64 * 4096, 8, dtype)
    assert (np_from(idx) == res[1]).all() and bits_equal(np_from(y), res[0], dtype)


@pytest.mark.parametrize("dtype", ["bf16", "fp32", "fp16"])
def test_ste_vs_oracle(ops, dtype):
    rng = np.random.default_rng(3)
    for n in (1, 7, 8, 64, 1000, 8192, 8200, 1 << 20, (1 << 20) + 3):
        x_np, x = make_input(rng, (1, n), dtype, "act")
        g_np, g = make_input(rng, (1, n), dtype, "weight")
        for lo, hi in ((-2.0, 2.0), (-0.5, 0.75), (-0.3009, 0.3009)):
            gx = ops.ste_backward(g, x, lo, hi)
            want = O.ste_bwd(g_np, x_np, lo, hi, dtype)
            assert bits_equal(np_from(gx), want, dtype), f"{dtype} n={n} clip=({lo},{hi})"


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_ste_row_bounds_path_equals_plain(ops, dtype):
    """fq_ste_bwd_rows (skips x for rows that cannot be clipped) == fq_ste_bwd == oracle."""
    rng = np.random.default_rng(4)
    for shape in [(64, 4096), (16, 11008), (5, 8200), (3, 264)]:
        x_np, x = make_input(rng, shape, dtype, "mixed")   # some rows tiny (safe), some exceed the clip
        x[1, 5] = float("nan")
        x_np = np_from(x)
        g_np, g = make_input(rng, shape, dtype, "weight")
        for kind in ("sym", "asym"):
            y, bounds = (ops.sym_quantize if kind == "sym" else ops.asym_quantize)(x, 8, False, want_bounds=True)
            gx = ops.ste_backward(g, x, -2.0, 2.0, row_bounds=bounds, rows_cols_hint=shape)
            want = O.ste_bwd(g_np, x_np, -2.0, 2.0, dtype)
            assert bits_equal(np_from(gx), want, dtype), f"{kind} {dtype} {shape}"
        b = bounds.cpu().numpy()
        xf = to_f32(x_np, dtype)
        ok = ~np.isnan(xf).any(axis=1)
        assert (b[ok, 0] == xf[ok].max(axis=1)).all() and (b[ok, 1] == xf[ok].min(axis=1)).all()


@pytest.mark.parametrize("dtype", ["bf16", "fp32", "fp16"])
@pytest.mark.parametrize("kind", ["sym", "asym"])
def test_train_mode_mask_path_vs_oracle(ops, kind, dtype):
    """fq_*_fwd_train + fq_ste_bwd_mask: forward values and the gradient (computed WITHOUT x) are bit-equal to
    the oracle's; covers partial 64-vector groups, multi-chunk rows, safe + unsafe + NaN rows, custom clips."""
    rng = np.random.default_rng(77)
    shapes = [(5, 8), (3, 264), (7, 512), (4, 520), (6, 4096), (3, 11008), (2, 13824), (2, 32768), (9, 64), (2, 65536)]
    for shape in shapes:
        x_np, x = make_input(rng, shape, dtype, "mixed")
        if shape[0] > 2:
            x[1, shape[1] // 2] = float("nan")
            x[2, 0] = 2.0
            x[2, shape[1] - 1] = -2.0
            x_np = np_from(x)
        g_np, g = make_input(rng, shape, dtype, "weight")
        for lo, hi, bits in ((-2.0, 2.0, 8), (-0.5, 0.75, 4), (-0.3009, 0.3009, 4)):
            res = ops.quantize_train(kind, x, bits, False, lo, hi)
            if res is None:
                assert shape[1] * (4 if dtype == "fp32" else 2) > 8192 * 16, f"{shape} should be served"
                continue
            y, bounds, mask = res
            if kind == "sym":
                yo, _, _ = O.sym_fwd(x_np, shape[0], shape[1], bits, dtype)
            else:
                yo, _, _, _ = O.asym_fwd(x_np, shape[0], shape[1], bits, dtype)
            assert bits_equal(np_from(y), yo, dtype), f"{kind} {dtype} {shape} fwd"
            gx = ops.ste_backward_mask(g, lo, hi, bounds, mask, shape[0], shape[1])
            want = O.ste_bwd(g_np, x_np, lo, hi, dtype)
            assert bits_equal(np_from(gx), want, dtype), f"{kind} {dtype} {shape} clip=({lo},{hi}): {mismatch_report(np_from(gx), want, dtype)}"


def test_mask_mode_does_not_keep_input_alive(ops):
    import llm_qat_amd
    from llm_qat_amd.utils_quant import SymQuantizer
    assert llm_qat_amd.get_backward_mode() == "mask"
    x = torch.randn(64, 4096, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    saved = []
    with torch.autograd.graph.saved_tensors_hooks(lambda t: (saved.append(t), t)[1], lambda t: t):   # (works for the C++ node and the Python one)
        y = SymQuantizer.apply(x, torch.tensor([-2.0, 2.0]), 8, False)
    # the ONE saved tensor is the side buffer (row bounds + 1-bit/element STE mask: 6 % of x); the input is not saved
    assert len(saved) == 1 and saved[0].dtype == torch.uint8 and saved[0].numel() == 64 * 8 + 64 * 4096 // 8
    assert all(t.data_ptr() != x.data_ptr() for t in saved)
    y.sum().backward()
    ref = torch.where((x >= 2) | (x <= -2), torch.zeros_like(x), torch.ones_like(x))
    assert torch.equal(x.grad, ref)


def test_non_contiguous_and_misaligned_inputs(ops):
    rng = np.random.default_rng(8)
    base = torch.from_numpy(rng.standard_normal((64, 48)).astype(np.float32)).cuda().bfloat16()
    xt = base.t()                                   # non-contiguous
    y = ops.sym_quantize(xt, 8, False)
    assert y.stride() == xt.stride() and not y.is_contiguous()
    want, _, _ = O.sym_fwd(np_from(xt), 48, 64, 8, "bf16")
    assert bits_equal(np_from(y), want, "bf16")
    flat = torch.from_numpy(rng.standard_normal(4096 * 3 + 1).astype(np.float32)).cuda().bfloat16()
    xm = flat[1:].view(3, 4096)                     # contiguous but 2-byte aligned only
    y = ops.sym_quantize(xm, 4, False)
    want, _, _ = O.sym_fwd(np_from(xm), 3, 4096, 4, "bf16")
    assert bits_equal(np_from(y), want, "bf16")
    g = torch.ones_like(xm)
    gx = ops.ste_backward(g, xm, -2.0, 2.0)
    assert bits_equal(np_from(gx), O.ste_bwd(np_from(g), np_from(xm), -2.0, 2.0, "bf16"), "bf16")


def test_error_behaviour_matches_reference(ops):
    from llm_qat_amd.utils_quant import SymQuantizer
    clip = torch.tensor([-2.0, 2.0])
    with pytest.raises(ValueError):      # utils_quant.py:70
        SymQuantizer.apply(torch.zeros(1, 1, 1, 1, 2, device="cuda"), clip, 8, False)
    SymQuantizer.apply(torch.zeros(1, 1, 1, 1, 2, device="cuda"), clip, 8, True)   # layerwise accepts any rank
    with pytest.raises(RuntimeError):    # no CPU fallback
        SymQuantizer.apply(torch.zeros(4, 4), clip, 8, False)
    z64 = SymQuantizer.apply(torch.zeros(4, 4, device="cuda", dtype=torch.float64), clip, 8, False)   # float64 is served since round 3 (tests/test_f64.py)
    assert z64.dtype == torch.float64 and not z64.any()
    with pytest.raises(NotImplementedError):
        SymQuantizer.apply(torch.zeros(4, 4, device="cuda", dtype=torch.int32), clip, 8, False)
    e = SymQuantizer.apply(torch.zeros(0, 8, device="cuda"), clip, 8, False)
    assert e.shape == (0, 8)
    z = SymQuantizer.apply(torch.tensor(1.5, device="cuda"), clip, 8, False)      # 0-dim: one row of one element
    yo, _, _ = O.sym_fwd(np.array([1.5], np.float32), 1, 1, 8, "fp32")
    assert z.shape == () and float(z) == float(yo[0])


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp32"])
def test_one_bit_sym_is_served_like_the_reference(ops, dtype):
    """num_bits = 1 for SymQuantizer: qmax = 2**0 - 1 = 0 (models/utils_quant.py:71), so the scale is 0 and every finite element quantizes
    to a signed zero (NaN / Inf stay NaN) -- useless, but defined, and reachable through the KV hooks' `kv_bits < 32` gate
    (modeling_llama_quant.py:321).  Kernel == oracle (values + bins), drop-in == live eager chain in every backward mode, with and without
    autocast, K+V pair launch included; export gives all-zero bins."""
    import llm_qat_amd
    from llm_qat_amd.utils_quant import SymQuantizer, quantize_kv
    from oracle import eager_chain as E
    clip = torch.tensor([-2.0, 2.0])
    torch.manual_seed(3)
    for shape in [(5, 1000), (3, 4096), (2, 7, 264), (4, 11008)]:
        x = (torch.randn(shape, device="cuda") * 1.7).to(TD[dtype])
        x.view(-1)[3], x.view(-1)[11] = float("inf"), float("nan")
        x.view(-1)[5] = -0.0
        rows, cols = O.rows_cols(tuple(shape), False)
        yo, io, _ = O.sym_fwd(np_from(x), rows, cols, 1, dtype)
        y, idx, _ = ops.sym_quantize_debug(x, 1, False)
        assert (np_from(idx) == io).all() and bits_equal(np_from(y), yo, dtype), f"{dtype} {shape}: {mismatch_report(np_from(y), yo, dtype)}"
        prev = llm_qat_amd.get_backward_mode()
        llm_qat_amd.set_semantics("device_eager")
        try:
            for autocast in ((False, True) if dtype != "fp32" else (False,)):
                xr = x.clone().requires_grad_(True)
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                    want = E.EagerSym.apply(xr, clip, 1, False)
                want.float().sum().backward()
                for mode in ("mask", "bounds", "plain"):
                    llm_qat_amd.set_backward_mode(mode)
                    xg = x.clone().requires_grad_(True)
                    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                        got = SymQuantizer.apply(xg, clip, 1, False)
                    got.float().sum().backward()
                    assert got.dtype == want.dtype and bits_equal(np_from(got), np_from(want), "fp32" if got.dtype == torch.float32 else dtype), (dtype, shape, autocast, mode)
                    assert bits_equal(np_from(xg.grad), np_from(xr.grad), dtype), (dtype, shape, autocast, mode, "grad")
                if len(shape) <= 3:
                    llm_qat_amd.set_backward_mode("mask")
                    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                        kq, vq = quantize_kv(x, x.flip(0).contiguous(), clip, clip, 1)
                        vw = E.EagerSym.apply(x.flip(0).contiguous(), clip, 1, False)
                    assert bits_equal(np_from(kq), np_from(want), "fp32" if kq.dtype == torch.float32 else dtype)
                    assert bits_equal(np_from(vq), np_from(vw), "fp32" if vq.dtype == torch.float32 else dtype)
        finally:
            llm_qat_amd.set_backward_mode(prev)
            llm_qat_amd.set_semantics("cpu_eager")
        ex = ops.sym_export(x, 1, False, container="int4", autocast=False)
        assert int(ex.unpacked().abs().max()) == 0


@pytest.mark.parametrize("autocast", [False, True])
def test_degenerate_shapes_behave_like_the_eager_chain(ops, autocast):
    """Tensors without elements and one-element tensors through SymQuantizer / AsymQuantizer (forward + backward, every backward mode),
    against the reference's op chain run live on this device (oracle/eager_chain.py restates models/utils_quant.py:37-87, :96-162 op for
    op): zero ROWS of a non-empty last dimension give an empty result; a reduction over nothing raises what torch.max raises in the
    reference -- IndexError for an empty reduction dimension, RuntimeError for the layerwise max and for the 4-D branch's ambiguous
    view(d0, d1, -1);
    one-element rows quantize to themselves' bins.  Same result, or the same exception TYPE."""
    import llm_qat_amd
    from llm_qat_amd.utils_quant import AsymQuantizer, SymQuantizer
    from oracle import eager_chain as E
    clip = torch.tensor([-2.0, 2.0])

    def run(q, shape, bits, layerwise):
        torch.manual_seed(0)
        x = torch.randn(shape, device="cuda").bfloat16().requires_grad_(True)
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                y = q.apply(x, clip, bits, layerwise)
            y.float().sum().backward()
        except Exception as e:  # noqa: BLE001
            return type(e)
        return y.detach(), x.grad

    prev = llm_qat_amd.get_backward_mode()
    llm_qat_amd.set_semantics("device_eager")
    try:
        for shape in [(0, 8), (3, 0), (0,), (2, 0, 8), (2, 3, 0), (0, 0), (0, 3, 4, 8), (2, 3, 0, 8), (2, 0, 4, 8), (2, 3, 4, 0), (0, 0, 4, 8),
                      (1, 1), (1,), (5, 1), (), (1, 1, 1, 1)]:
            for layerwise in (False, True):
                for ref_q, q in ((E.EagerSym, SymQuantizer), (E.EagerAsym, AsymQuantizer)):
                    if autocast and q is AsymQuantizer:
                        continue
                    want = run(ref_q, shape, 8, layerwise)
                    for mode in ("mask", "bounds", "plain"):
                        llm_qat_amd.set_backward_mode(mode)
                        got = run(q, shape, 8, layerwise)
                        tag = f"{q.__name__} {shape} layerwise={layerwise} autocast={autocast} mode={mode}"
                        if isinstance(want, type):
                            assert got is want, f"{tag}: reference raises {want.__name__}, drop-in gave {got if isinstance(got, type) else 'a result'}"
                        else:
                            assert not isinstance(got, type), f"{tag}: drop-in raised {got}, the reference returns {tuple(want[0].shape)}"
                            for a, b in zip(got, want):
                                assert a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b), tag
    finally:
        llm_qat_amd.set_backward_mode(prev)
        llm_qat_amd.set_semantics("cpu_eager")


# ------------------------------------------------------------------------------------------
# (3) full-size properties (BASELINE.json sizes; the oracle checks a sample of rows)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,bits,style", [((4096, 11008), 4, "weight"), ((4096, 11008), 8, "act"),
                                              ((2048, 4096), 8, "act"), ((2048, 11008), 8, "act"),
                                              ((2048, 4096), 4, "act"), ((5120, 13824), 4, "weight"),
                                              ((4096, 11008), 8, "weight"),     # config 4 (W8): the +-128 bf16 bin on weight-style rows
                                              ((2048, 5120), 4, "act")])        # config 5 (13B): KV4 without autocast
def test_full_size_properties(ops, shape, bits, style):
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.randn(shape, generator=g, device="cuda", dtype=torch.float32)
    if style == "weight":
        x *= 0.02
    else:
        x[torch.rand(shape, generator=g, device="cuda") < 1e-3] *= 20.0
    x = x.bfloat16()
    y, idx, s = ops.sym_quantize_debug(x, bits, False)
    qmax = 2 ** (bits - 1) - 1
    # bins are integers within the representable range (+1 for the bf16 8-bit overshoot)
    assert int(idx.abs().max()) <= qmax + 1
    # every row reaches the top of the range (the row max maps to qmax-1 .. qmax+1: s is rounded in bf16)
    assert bool((idx.abs().amax(dim=1) >= qmax - 1).all())
    # row-permutation equivariance: scales are per row, so permuting rows permutes outputs
    perm = torch.randperm(shape[0], device="cuda", generator=g)
    assert torch.equal(ops.sym_quantize(x[perm].contiguous(), bits, False), y[perm])
    # within a row, element order is irrelevant to the scale
    cperm = torch.randperm(shape[1], device="cuda", generator=g)
    assert torch.equal(ops.sym_quantize(x[:, cperm].contiguous(), bits, False), y[:, cperm])
    # bins are stable under re-quantization of the output: fq(fq(x)) has the same idx (scale grid is preserved)
    # -- not guaranteed by the algebra in bf16, so only the weaker monotonicity is asserted:
    xs = x[:64].float()
    order = xs.argsort(dim=1)
    assert bool((torch.gather(idx[:64], 1, order).diff(dim=1) >= 0).all())   # idx is monotone in x
    # oracle on a sample of rows (first, last, and 30 random ones), bit-exact
    rows = sorted(set([0, shape[0] - 1] + torch.randint(0, shape[0], (30,), generator=torch.Generator().manual_seed(1)).tolist()))
    xs_np = np_from(x[rows])
    yo, io, so = O.sym_fwd(xs_np, len(rows), shape[1], bits, "bf16")
    assert (np_from(idx[rows]) == io).all()
    assert bits_equal(np_from(y[rows]), yo, "bf16")
    # backward at full size: mask property + checksum against torch.where
    gg = (torch.randn(shape, generator=g, device="cuda") * 1e-3).bfloat16()
    gx = ops.ste_backward(gg, x, -2.0, 2.0)
    ref = torch.where((x >= 2.0) | (x <= -2.0), torch.zeros_like(gg), gg)
    assert torch.equal(gx, ref)
    _, bounds = ops.sym_quantize(x, bits, False, want_bounds=True)
    gx2 = ops.ste_backward(gg, x, -2.0, 2.0, row_bounds=bounds, rows_cols_hint=shape)
    assert torch.equal(gx2, ref)
    y3, b3, m3 = ops.quantize_train("sym", x, bits, False, -2.0, 2.0)
    assert torch.equal(y3, y)
    assert torch.equal(ops.ste_backward_mask(gg, -2.0, 2.0, b3, m3, shape[0], shape[1]), ref)


@pytest.mark.parametrize("shape,bits", [((2048, 4096), 4), ((2048, 5120), 8), ((4096, 11008), 8)])
def test_full_size_autocast_properties(ops, shape, bits):
    """The autocast (fp32-arithmetic) kernels at model sizes: narrow result == fp32 result rounded once; the K/V pair
    launch == two single launches; the fp32-gradient mask backward == where(clipped, 0, round(g)); oracle on sampled rows."""
    g = torch.Generator(device="cuda").manual_seed(4321)
    k = torch.randn(shape, generator=g, device="cuda")
    k[torch.rand(shape, generator=g, device="cuda") < 1e-3] *= 20.0
    k = k.bfloat16()
    v = (torch.randn(shape, generator=g, device="cuda") * 0.7).bfloat16()
    yw, side, rows, cols, got = ops.sym_forward_autocast(k, bits, False, wide=True, train="mask")
    assert got == "mask" and yw.dtype == torch.float32
    yn, _, _, _, _ = ops.sym_forward_autocast(k, bits, False, wide=False)
    assert torch.equal(yn, yw.to(torch.bfloat16))
    sel = sorted(set([0, shape[0] - 1] + torch.randint(0, shape[0], (30,), generator=torch.Generator().manual_seed(2)).tolist()))
    yo, _ = O.sym_fwd_autocast(np_from(k[sel]), len(sel), shape[1], bits, "bf16", wide=True)
    assert bits_equal(np_from(yw[sel]).reshape(yo.shape), yo, "fp32")
    g32 = torch.randn(shape, generator=g, device="cuda") * 1e-3
    ref = torch.where((k >= 2.0) | (k <= -2.0), torch.zeros((), device="cuda", dtype=torch.bfloat16), g32.to(torch.bfloat16))
    gx = ops.train_backward_wide(g32, side, rows, cols, -2.0, 2.0, torch.bfloat16)
    assert torch.equal(gx, ref) and int((ref == 0).sum()) > 0
    with torch.autocast("cuda", dtype=torch.bfloat16):
        res = ops.pair_forward(k, v, bits, bits, -2.0, 2.0, True, True, wide=True)
        vw = ops.sym_forward_autocast(v, bits, False, wide=True)[0]
    assert res is not None and torch.equal(res[0], yw) and torch.equal(res[1], vw)
    gv = torch.randn(shape, generator=g, device="cuda") * 1e-3
    ok, ov = ops.pair_backward_wide(g32, gv, res[2], res[3], res[4], res[5], res[6], -2.0, 2.0, torch.bfloat16)
    assert torch.equal(ok, ref)
    assert torch.equal(ov, torch.where((v >= 2.0) | (v <= -2.0), torch.zeros((), device="cuda", dtype=torch.bfloat16), gv.to(torch.bfloat16)))


# ------------------------------------------------------------------------------------------
# device-eager semantics: the kernels with sem=DEVICE_EAGER against LIVE ATen ops on this GPU
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["bf16", "fp32", "fp16"])
def test_device_eager_semantics_vs_live_aten(ops, dtype):
    """What the reference's eager path computes ON THE DEVICE is given by running its op chain
    (oracle/eager_chain.py) on the GPU.  With set_semantics('device_eager') the kernels must match
    that bit for bit -- including rows with a tiny max, where CPU and device eager differ."""
    import llm_qat_amd
    from oracle import eager_chain as E
    rng = np.random.default_rng(21)
    llm_qat_amd.set_semantics("device_eager")
    try:
        bad = []
        for shape in [(16, 256), (8, 4096), (4, 11008), (5, 33)]:
            for style in ("mixed", "act"):
                _, x = make_input(rng, shape, dtype, style)
                for bits in (4, 8):
                    y, idx, _ = ops.sym_quantize_debug(x, bits, False)
                    ye, ie, _ = E.sym_forward(x, bits, False, want_idx=True)
                    if not bits_equal(np_from(y), np_from(ye), dtype):
                        bad.append(f"sym {dtype} {shape} {style} b{bits}: {mismatch_report(np_from(y), np_from(ye), dtype)}")
                    y, idx, _ = ops.asym_quantize_debug(x, bits, False)
                    ye = E.asym_forward(x, bits, False)
                    if not bits_equal(np_from(y), np_from(ye), dtype):
                        bad.append(f"asym {dtype} {shape} {style} b{bits}: {mismatch_report(np_from(y), np_from(ye), dtype)}")
                gsrc = torch.randn(shape, device="cuda").to(TD[dtype])
                gx = ops.ste_backward(gsrc, x, -2.0, 2.0)
                if not torch.equal(gx, E.ste_backward(gsrc, x, torch.tensor([-2.0, 2.0]))):
                    bad.append(f"ste {dtype} {shape}")
        assert not bad, "\n".join(bad[:20])
    finally:
        llm_qat_amd.set_semantics("cpu_eager")


# ------------------------------------------------------------------------------------------
# randomized stress: shapes, dtypes, bits, clips, NaN/Inf injection -- every data flow vs the oracle
# ------------------------------------------------------------------------------------------
def test_randomized_stress_all_paths(ops):
    import llm_qat_amd
    from llm_qat_amd.utils_quant import AsymQuantizer, SymQuantizer
    rng = np.random.default_rng(20261004)
    col_choices = [1, 3, 8, 24, 64, 100, 256, 264, 512, 1000, 1024, 2048, 4096, 4104, 8192, 11008, 16384, 20000]
    prev = llm_qat_amd.get_backward_mode()
    try:
        import os
        for trial in range(int(os.environ.get("LLMQAT_STRESS_TRIALS", "120"))):
            dtype = ["bf16", "fp32", "fp16"][trial % 3]
            kind = "sym" if rng.random() < 0.6 else "asym"
            rows, cols = int(rng.integers(1, 40)), int(rng.choice(col_choices))
            bits = int(rng.choice([2, 3, 4, 6, 8, 12, 16])) if kind == "sym" else int(rng.choice([1, 2, 4, 8, 16]))
            layerwise = bool(rng.random() < 0.15)
            scales = [1e-4, 0.02, 1.0, 3.0, 100.0]
            if rng.random() < 0.25:   # row magnitudes at the ends of the dtype's range (denormal scales, overflowing alpha, div_exact's guards)
                scales = {"fp32": [1e-42, 1e-38, 1e-30, 1e-19, 1e-10, 1e10, 1e19, 1e30, 1e38], "bf16": [1e-38, 1e-30, 1e-19, 1e-10, 1e10, 1e19, 1e30, 1e38],
                          "fp16": [1e-7, 1e-6, 1e-5, 1e-3, 30.0, 1e3, 2e4, 6e4]}[dtype]
            with np.errstate(over="ignore"):   # 3e38-scaled rows may overflow to inf: wanted
                x = rng.standard_normal((rows, cols)).astype(np.float32) * rng.choice(scales, size=(rows, 1)).astype(np.float32)
            if rng.random() < 0.3:
                x[rng.integers(0, rows), rng.integers(0, cols)] = rng.choice([np.nan, np.inf, -np.inf])
            if rng.random() < 0.3:
                x[rng.integers(0, rows)] = 0.0
            lo, hi = [(-2.0, 2.0), (-0.5, 0.75), (-1.0, 1.0), (-0.3009, 0.3009)][int(rng.integers(0, 4))]
            xt = torch.from_numpy(x).to(TD[dtype])
            x_np, xd = np_from(xt), xt.cuda()
            g_np, gd = make_input(rng, (rows, cols), dtype, "act")
            layout = rng.random()
            if layout < 0.15:      # contiguous but only element-aligned storage (views into a flat buffer, as FSDP hands out)
                flat = torch.empty(rows * cols + 1, dtype=TD[dtype], device="cuda")
                flat[1:].copy_(xd.reshape(-1))
                xd = flat[1:].view(rows, cols)
            elif layout < 0.3:     # non-contiguous (transposed storage)
                xd = xd.t().contiguous().t()
                gd = gd.t().contiguous().t()
            r, c = O.rows_cols((rows, cols), layerwise)
            if kind == "sym":
                yo, io, _ = O.sym_fwd(x_np, r, c, bits, dtype)
                dbg, quant = ops.sym_quantize_debug, SymQuantizer
            else:
                yo, io, _, _ = O.asym_fwd(x_np, r, c, bits, dtype)
                dbg, quant = ops.asym_quantize_debug, AsymQuantizer
            want_g = O.ste_bwd(g_np, x_np, lo, hi, dtype)
            tag = f"trial {trial}: {kind} {dtype} [{rows},{cols}] b{bits} lw={layerwise} clip=({lo},{hi})"
            y, idx, _ = dbg(xd, bits, layerwise)
            assert (np_from(idx) == io).all(), tag + " idx"
            assert bits_equal(np_from(y), yo, dtype), tag + " y: " + mismatch_report(np_from(y), yo, dtype)
            for mode in ("mask", "bounds", "plain"):
                llm_qat_amd.set_backward_mode(mode)
                xr = xd.clone().requires_grad_(True)
                out = quant.apply(xr, torch.tensor([lo, hi]), bits, layerwise)
                out.backward(gd)
                assert bits_equal(np_from(out), yo, dtype), f"{tag} {mode} fwd"
                assert bits_equal(np_from(xr.grad), want_g, dtype), f"{tag} {mode} grad: " + mismatch_report(np_from(xr.grad), want_g, dtype)
            if kind == "sym" and dtype != "fp32":   # the same tensor under autocast: fp32 arithmetic, fp32 result
                ya, _ = O.sym_fwd_autocast(x_np, r, c, bits, dtype, wide=True)
                llm_qat_amd.set_backward_mode(["mask", "bounds", "plain"][trial % 3])
                with torch.autocast("cuda", dtype=TD[dtype]):
                    xr = xd.clone().requires_grad_(True)
                    out = quant.apply(xr, torch.tensor([lo, hi]), bits, layerwise)
                out.backward(gd.float())
                assert out.dtype == torch.float32 and bits_equal(np_from(out).reshape(ya.shape), ya, "fp32"), f"{tag} autocast fwd"
                assert bits_equal(np_from(xr.grad), want_g, dtype), f"{tag} autocast grad"
    finally:
        llm_qat_amd.set_backward_mode(prev)


# ------------------------------------------------------------------------------------------
# autocast: LLM-QAT trains under torch.autocast("cuda", bf16); `reciprocal` is on autocast's fp32 list, so the
# reference's SymQuantizer computes in fp32 behind it and RETURNS fp32 for a 16-bit input.  Truth = the live ATen chain.
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_sym_under_autocast_vs_live_aten(ops, dtype):
    import llm_qat_amd
    from llm_qat_amd.utils_quant import SymQuantizer, _SymQuantizerOperand
    from oracle import eager_chain as E
    rng = np.random.default_rng(31)
    clip = torch.tensor([-2.0, 2.0])
    prev = llm_qat_amd.get_backward_mode()
    # GUARD.  The autocast arithmetic is pinned by live ATen on this GPU, not by reference fixtures
    # (none can exist: CPU autocast does not promote `reciprocal`).  Which kernel the drop-in selects rests on ONE fact
    # about this torch build -- `reciprocal` is on CUDA autocast's fp32 list, so `int / Tensor` (utils_quant.py:71) comes
    # back in fp32 while the `max + 1e-6` in front of it stays in the tensor dtype.  If a torch upgrade changes the
    # autocast lists this fails loudly instead of ops.autocast_active() silently picking the wrong arithmetic.
    with torch.autocast("cuda", dtype=TD[dtype]):
        probe = torch.rand(4, 64, device="cuda").to(TD[dtype]) + 0.5
        top = torch.max(torch.abs(probe), dim=-1, keepdim=True)[0].expand_as(probe)
        t1 = top + 1e-6
        s = 127 / t1
        assert top.dtype == TD[dtype] and t1.dtype == TD[dtype], "abs / max / add are no longer dtype-preserving under autocast"
        assert s.dtype == torch.float32, "`int / Tensor` no longer autocasts to fp32: the autocast op lists of this torch build changed"
        assert (probe * s).dtype == torch.float32 and torch.round(probe * s).div(s + 1e-6).dtype == torch.float32
        assert torch.reciprocal(t1).dtype == torch.float32
    try:
        for shape in [(16, 256), (8, 4096), (3, 11008), (5, 33), (2, 3, 64), (4, 4104), (2, 20000)]:
            for style, bits in (("act", 8), ("weight", 4), ("mixed", 8), ("mixed", 16)):
                x_np, x = make_input(rng, shape, dtype, style)
                g = torch.randn(shape, device="cuda")
                with torch.autocast("cuda", dtype=TD[dtype]):
                    xr = x.clone().requires_grad_(True)
                    ref = E.EagerSym.apply(xr, clip, bits, False)
                ref.backward(g)
                assert ref.dtype == torch.float32 and xr.grad.dtype == TD[dtype]
                rows, cols = O.rows_cols(shape, False)
                yo, _ = O.sym_fwd_autocast(x_np, rows, cols, bits, dtype, wide=True)
                tag = f"{dtype} {shape} {style} b{bits}"
                assert bits_equal(np_from(ref).reshape(yo.shape), yo, "fp32"), tag + " oracle vs live ATen"
                for mode in ("mask", "bounds", "plain"):
                    llm_qat_amd.set_backward_mode(mode)
                    with torch.autocast("cuda", dtype=TD[dtype]):
                        xo = x.clone().requires_grad_(True)
                        y = SymQuantizer.apply(xo, clip, bits, False)
                        xn = x.clone().requires_grad_(True)
                        yn = _SymQuantizerOperand.apply(xn, clip, bits, False)
                    assert y.dtype == torch.float32 and bits_equal(np_from(y), np_from(ref), "fp32"), f"{tag} {mode} wide: {mismatch_report(np_from(y), np_from(ref), 'fp32')}"
                    assert yn.dtype == TD[dtype] and bits_equal(np_from(yn), np_from(ref.to(TD[dtype])), dtype), f"{tag} {mode} narrow"
                    y.backward(g)
                    yn.backward(g.to(TD[dtype]))
                    assert bits_equal(np_from(xo.grad), np_from(xr.grad), dtype), f"{tag} {mode} grad"
                    assert bits_equal(np_from(xn.grad), np_from(xr.grad), dtype), f"{tag} {mode} narrow grad"
        # fp32 tensors are untouched by autocast; Asym has no autocast-listed op
        x32 = torch.randn(8, 512, device="cuda")
        xb = torch.randn(8, 512, device="cuda").to(TD[dtype])
        with torch.autocast("cuda", dtype=TD[dtype]):
            assert torch.equal(SymQuantizer.apply(x32, clip, 8, False), E.sym_forward(x32, 8))
            llm_qat_amd.set_semantics("device_eager")
            from llm_qat_amd.utils_quant import AsymQuantizer
            ya, ye = AsymQuantizer.apply(xb, clip, 8, False), E.asym_forward(xb, 8)
            assert ya.dtype == ye.dtype == TD[dtype] and torch.equal(ya, ye)
    finally:
        llm_qat_amd.set_backward_mode(prev)
        llm_qat_amd.set_semantics("cpu_eager")


def test_quantize_linear_under_autocast_vs_live_aten(ops):
    """the module as kd_trainer runs it (bf16 weights, bf16 autocast): outputs and both gradients bit-identical to the
    eager chain, although the drop-in skips the fp32 materialisation of the quantized operands"""
    import sys
    sys.path.insert(0, __import__("os").path.dirname(__file__))
    import tiny_llama as TL
    from llm_qat_amd.utils_quant import QuantizeLinear
    EQ = TL.EagerQuant()
    torch.manual_seed(3)
    for wb, ab in ((4, 8), (8, 8), (4, 16)):
        ours = QuantizeLinear(1024, 512, w_bits=wb, a_bits=ab).cuda().bfloat16()
        ref = EQ.QuantizeLinear(1024, 512, w_bits=wb, a_bits=ab).cuda().bfloat16()
        with torch.no_grad():
            ref.weight.copy_(ours.weight)
            ours.weight[3, 5] = ref.weight[3, 5] = 2.5
        xs = (torch.randn(4, 64, 1024, device="cuda") * 1.5).bfloat16()
        res = []
        for m in (ours, ref):
            x = xs.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = m(x)
                loss = out.float().square().mean()
            loss.backward()
            res.append((out.detach(), x.grad, m.weight.grad))
        for a, b in zip(res[0], res[1]):
            assert a.dtype == b.dtype and torch.equal(a, b), (wb, ab)


@pytest.mark.parametrize("dtype", ["bf16", "fp32", "fp16"])
def test_device_eager_on_adversarial_rows_vs_live_aten(ops, dtype):
    """NaN / Inf / denormal / tiny-max / tie rows (the reference fixtures' adversarial inputs) through the kernels in
    device-eager mode against the live ATen chain on this GPU -- with and without autocast."""
    import llm_qat_amd
    from llm_qat_amd.utils_quant import AsymQuantizer, SymQuantizer
    from oracle import eager_chain as E
    G = golden("sym_fwd.npz")
    clip = torch.tensor([-2.0, 2.0])
    llm_qat_amd.set_semantics("device_eager")
    try:
        bad = []
        for bits in (4, 8, 16):
            c = next(c for c in G.cases if c["name"] == f"sym_{dtype}_b{bits}_adversarial")
            x = dev_from(G.arr(c, "x"), dtype)
            names = c["row_names"]
            for label, ours, ref in (("sym", SymQuantizer.apply(x, clip, bits, False), E.sym_forward(x, bits)),
                                     ("asym", AsymQuantizer.apply(x, clip, bits, False), E.asym_forward(x, bits))):
                a, b = np_from(ours), np_from(ref)
                for r, nm in enumerate(names):
                    if label == "asym" and nm in ("neg_zero",):
                        continue   # which of -0.0 / +0.0 is "the" minimum is unspecified
                    if not bits_equal(a[r], b[r], dtype):
                        bad.append(f"{label} b{bits} row {nm}: {mismatch_report(a[r], b[r], dtype)}")
            if dtype != "fp32":
                with torch.autocast("cuda", dtype=TD[dtype]):
                    ours, ref = SymQuantizer.apply(x, clip, bits, False), E.sym_forward(x, bits)
                a, b = np_from(ours), np_from(ref)
                for r, nm in enumerate(names):
                    if not bits_equal(a[r], b[r], "fp32"):
                        bad.append(f"autocast sym b{bits} row {nm}: {mismatch_report(a[r], b[r], 'fp32')}")
        assert not bad, "\\n".join(bad[:20])
    finally:
        llm_qat_amd.set_semantics("cpu_eager")


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_layerwise_and_long_rows_under_autocast_vs_live_aten(ops, dtype):
    """layerwise (one row, two-pass kernels) and rows beyond the register kernels under autocast, incl. QuantizeLinear
    with weight_layerwise / act_layerwise -- against the live ATen chain"""
    from llm_qat_amd.utils_quant import QuantizeLinear, SymQuantizer
    from oracle import eager_chain as E
    import sys
    sys.path.insert(0, __import__("os").path.dirname(__file__))
    import tiny_llama as TL
    rng = np.random.default_rng(9)
    clip = torch.tensor([-2.0, 2.0])
    for shape, layerwise in [((256, 1024), True), ((3, 40000), False), ((2, 70001), False), ((2, 3, 4, 5), True), ((64, 4097), True)]:
        _, x = make_input(rng, shape, dtype, "act")
        g = torch.randn(shape, device="cuda")
        with torch.autocast("cuda", dtype=TD[dtype]):
            xr = x.clone().requires_grad_(True)
            ref = E.EagerSym.apply(xr, clip, 8, layerwise)
            xo = x.clone().requires_grad_(True)
            out = SymQuantizer.apply(xo, clip, 8, layerwise)
        ref.backward(g)
        out.backward(g)
        assert out.dtype == ref.dtype == torch.float32 and bits_equal(np_from(out), np_from(ref), "fp32"), (shape, layerwise)
        assert bits_equal(np_from(xo.grad), np_from(xr.grad), dtype), (shape, layerwise)
    EQ = TL.EagerQuant()
    ours = QuantizeLinear(512, 256, w_bits=4, a_bits=8, weight_layerwise=True, act_layerwise=True).cuda().to(TD[dtype])
    refm = EQ.QuantizeLinear(512, 256, w_bits=4, a_bits=8, weight_layerwise=True, act_layerwise=True).cuda().to(TD[dtype])
    with torch.no_grad():
        refm.weight.copy_(ours.weight)
    xs = (torch.randn(8, 64, 512, device="cuda") * 1.5).to(TD[dtype])
    res = []
    for m in (ours, refm):
        x = xs.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=TD[dtype]):
            o = m(x)
        o.float().square().mean().backward()
        res.append((o.detach(), x.grad, m.weight.grad))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_wide_result_mask_path_vs_oracle(ops, dtype):
    """fp32-result (autocast) forward with its own STE mask + fq_ste_bwd_mask_wide (fp32 gradient in, 16-bit gradient
    out, never reading x), single tensors and K/V-style pairs, through the C ABI: forward bit-equal to the oracle's
    autocast chain, gradient bit-equal to oracle-STE applied to the gradient rounded to the input dtype (the autograd
    engine's cast).  Partial 64-half-vector groups, multi-chunk rows, safe / unsafe / NaN rows, asymmetric clips."""
    from llm_qat_amd import _lib
    L = _lib.lib()
    code = {"bf16": _lib.DTYPE_BF16, "fp16": _lib.DTYPE_F16}[dtype]
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(78)

    def side(rows, cols):
        mb = L.fq_ste_mask_bytes(rows, cols, code)
        assert mb
        return torch.zeros(rows * 8 + mb, dtype=torch.uint8, device="cuda"), mb

    def make(shape):
        x_np, x = make_input(rng, shape, dtype, "mixed")
        if shape[0] > 2:
            x[1, shape[1] // 2] = float("nan")
            x[2, 0] = 2.0
            x[2, shape[1] - 1] = -2.0
            x_np = np_from(x)
        g32 = torch.randn(shape, device="cuda") * 1e-3
        return x_np, x, g32

    def want_grad(g32, x_np, lo, hi):
        return O.ste_bwd(np_from(g32.to(TD[dtype])), x_np, lo, hi, dtype)

    for cols in (8, 264, 512, 520, 4096, 11008, 13824, 32768):
        for lo, hi, bits in ((-2.0, 2.0, 8), (-0.5, 0.75, 4)):
            r0, r1 = 5, 3
            x0_np, x0, g0 = make((r0, cols))
            x1_np, x1, g1 = make((r1, cols))
            # single tensor: fq_sym_fwd_autocast(wide_out=1, mask) -> fq_ste_bwd_mask_wide(rows1 = 0)
            y = torch.empty(r0, cols, device="cuda")
            s0, mb0 = side(r0, cols)
            rc = L.fq_sym_fwd_autocast(x0.data_ptr(), y.data_ptr(), r0, cols, bits, code, 1, 1, lo, hi, s0.data_ptr(), s0.data_ptr() + r0 * 8, mb0, None, 0, st)
            _lib.check(rc, "wide fwd")
            yo, _ = O.sym_fwd_autocast(x0_np, r0, cols, bits, dtype, wide=True)
            assert bits_equal(np_from(y).reshape(yo.shape), yo, "fp32"), f"{dtype} cols={cols} wide fwd"
            gx = torch.empty(r0, cols, device="cuda", dtype=TD[dtype])
            rc = L.fq_ste_bwd_mask_wide(g0.data_ptr(), gx.data_ptr(), r0, s0.data_ptr(), s0.data_ptr() + r0 * 8, None, None, 0, None, None,
                                        cols, lo, hi, code, st)
            _lib.check(rc, "wide bwd")
            want = want_grad(g0, x0_np, lo, hi)
            assert bits_equal(np_from(gx), want, dtype), f"{dtype} cols={cols} clip=({lo},{hi}) wide grad: {mismatch_report(np_from(gx), want, dtype)}"
            # pair (K and V): fq_sym_fwd_pair(autocast=2) -> fq_ste_bwd_mask_wide with both tensors
            y0, y1 = torch.empty(r0, cols, device="cuda"), torch.empty(r1, cols, device="cuda")
            s0, mb0 = side(r0, cols)
            s1, mb1 = side(r1, cols)
            rc = L.fq_sym_fwd_pair(x0.data_ptr(), y0.data_ptr(), r0, bits, s0.data_ptr(), s0.data_ptr() + r0 * 8, mb0,
                                   x1.data_ptr(), y1.data_ptr(), r1, bits, s1.data_ptr(), s1.data_ptr() + r1 * 8, mb1,
                                   cols, code, 1, 2, lo, hi, st)
            _lib.check(rc, "wide pair fwd")
            yo1, _ = O.sym_fwd_autocast(x1_np, r1, cols, bits, dtype, wide=True)
            assert bits_equal(np_from(y0).reshape(yo.shape), yo, "fp32") and bits_equal(np_from(y1).reshape(yo1.shape), yo1, "fp32"), f"{dtype} cols={cols} pair fwd"
            gx0, gx1 = torch.empty(r0, cols, device="cuda", dtype=TD[dtype]), torch.empty(r1, cols, device="cuda", dtype=TD[dtype])
            rc = L.fq_ste_bwd_mask_wide(g0.data_ptr(), gx0.data_ptr(), r0, s0.data_ptr(), s0.data_ptr() + r0 * 8,
                                        g1.data_ptr(), gx1.data_ptr(), r1, s1.data_ptr(), s1.data_ptr() + r1 * 8, cols, lo, hi, code, st)
            _lib.check(rc, "wide pair bwd")
            assert bits_equal(np_from(gx0), want, dtype), f"{dtype} cols={cols} pair grad 0"
            want1 = want_grad(g1, x1_np, lo, hi)
            assert bits_equal(np_from(gx1), want1, dtype), f"{dtype} cols={cols} pair grad 1: {mismatch_report(np_from(gx1), want1, dtype)}"
    # shapes the wide mask path does not serve are refused, not mis-served
    x = torch.randn(2, 40000, device="cuda").to(TD[dtype])
    y = torch.empty(2, 40000, device="cuda")
    s, mb = side(2, 40000)
    rc = L.fq_sym_fwd_autocast(x.data_ptr(), y.data_ptr(), 2, 40000, 8, code, 1, 1, -2.0, 2.0, s.data_ptr(), s.data_ptr() + 16, mb, None, 0, st)
    assert rc == _lib.ERR_UNSUPPORTED


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp32"])
def test_ste_mask_is_the_documented_row_bitmap(ops, dtype):
    """ABI 3 (include/llmqat_fakequant.h): the STE mask is a plain bitmap per row -- bit j%8 of byte j/8 of row r (row stride
    8 * ceil(cols/64) bytes) = the reference's predicate `x >= hi || x <= lo` (utils_quant.py:85-86) for element j -- written
    for every row whose recorded bounds do not prove that nothing clips, by EVERY producer (Sym / Asym training forward, the
    autocast forward with a 16-bit or an fp32 result, the pair launch, the scale pre-pass), and every consumer reads it:
    a mask written by the fp32-result forward serves fq_ste_bwd_mask on a cast gradient, a 16-bit forward's serves
    fq_ste_bwd_mask_wide.  The predicate comes from the oracle's STE (gradient of ones == 0)."""
    from llm_qat_amd import _lib
    L = _lib.lib()
    code = {"bf16": _lib.DTYPE_BF16, "fp16": _lib.DTYPE_F16, "fp32": _lib.DTYPE_F32}[dtype]
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(79)
    inf = float("inf")
    clips = ((-2.0, 2.0), (-0.5, 0.75), (-0.3009, 0.3009), (-inf, inf), (-0.0, 0.0), (-1e-3, 1e-3), (-6e-8, 6e-8), (-7e4, 7e4))   # (fp16: subnormal / overflowing clip)
    shapes = ((6, 4), (5, 12), (5, 8), (4, 264), (5, 520), (4, 4096), (4, 11008)) if dtype == "fp32" else ((5, 8), (6, 24), (4, 264), (5, 520), (4, 4096), (4, 11008))

    def expected(x_np, rows, cols, lo, hi):
        ones = np_from(torch.ones(rows, cols, dtype=TD[dtype]))
        gx = O.ste_bwd(ones, x_np, lo, hi, dtype).reshape(rows, cols)
        flags = (to_f32(gx, dtype) == 0).astype(np.uint8)
        return np.packbits(flags, axis=1, bitorder="little")           # [rows, ceil(cols / 8)]

    def check(side, rows, cols, x_np, lo, hi, what):
        stride = 8 * ((cols + 63) // 64)
        assert side.numel() == rows * 8 + rows * stride
        b = side[: rows * 8].view(torch.float32).view(rows, 2).cpu().numpy()
        m = side[rows * 8:].cpu().numpy().reshape(rows, stride)[:, : (cols + 7) // 8]
        want = expected(x_np, rows, cols, lo, hi)
        lo_r, hi_r = (float(torch.tensor(v).to(TD[dtype]).float()) for v in (lo, hi))     # the comparison runs in the tensor dtype
        written = ~((b[:, 0] < hi_r) & (b[:, 1] > lo_r))
        assert written.any(), what
        assert (m[written] == want[written]).all(), f"{what}: bitmap differs in rows {np.argwhere((m != want).any(1) & written).ravel().tolist()}"
        assert (want[~written] == 0).all(), f"{what}: a row the bounds call safe has clipped elements"

    for rows, cols in shapes:
        x_np, x = make_input(rng, (rows, cols), dtype, "mixed")
        x[0] = x[0] * 0.01                                                               # a row the standard clip cannot reach
        x[1, cols // 2] = float("nan")
        x[2, 0], x[2, cols - 1] = 2.0, -2.0
        x[3, cols // 3], x[3, 1] = inf, -inf
        x_np = np_from(x)
        mb = L.fq_ste_mask_bytes(rows, cols, code)
        assert mb == rows * 8 * ((cols + 63) // 64)
        for lo, hi in clips:
            for kind in ("sym", "asym"):
                side = torch.zeros(rows * 8 + mb, dtype=torch.uint8, device="cuda")
                y = torch.empty_like(x)
                fn = L.fq_sym_fwd_train if kind == "sym" else L.fq_asym_fwd_train
                _lib.check(fn(x.data_ptr(), y.data_ptr(), rows, cols, 8, code, 0, lo, hi, side.data_ptr(), side.data_ptr() + rows * 8, mb, st), kind)
                check(side, rows, cols, x_np, lo, hi, f"{kind}_fwd_train {dtype} {(rows, cols)} clip=({lo},{hi})")
            # the scale pre-pass records the same side buffer
            side = torch.zeros(rows * 8 + mb, dtype=torch.uint8, device="cuda")
            sc = torch.empty(rows, 2, device="cuda")
            _lib.check(L.fq_sym_row_scales(x.data_ptr(), sc.data_ptr(), rows, cols, 8, code, 0, 0, lo, hi, side.data_ptr(), side.data_ptr() + rows * 8, mb, st), "scales")
            check(side, rows, cols, x_np, lo, hi, f"fq_sym_row_scales {dtype} {(rows, cols)} clip=({lo},{hi})")
            if dtype == "fp32":
                continue
            g32 = torch.randn(rows, cols, device="cuda")
            g16 = g32.to(TD[dtype])
            want_g = O.ste_bwd(np_from(g16), x_np, lo, hi, dtype)
            sides = {}
            for wide in (0, 1):
                side = torch.zeros(rows * 8 + mb, dtype=torch.uint8, device="cuda")
                y = torch.empty(rows, cols, device="cuda", dtype=torch.float32 if wide else TD[dtype])
                rc = L.fq_sym_fwd_autocast(x.data_ptr(), y.data_ptr(), rows, cols, 8, code, 1, wide, lo, hi, side.data_ptr(), side.data_ptr() + rows * 8, mb, None, 0, st)
                _lib.check(rc, "autocast fwd")
                check(side, rows, cols, x_np, lo, hi, f"fq_sym_fwd_autocast wide={wide} {dtype} {(rows, cols)} clip=({lo},{hi})")
                sides[wide] = side
            # cross-consumption: any consumer reads any producer's mask
            for wide_fwd in (0, 1):
                sd = sides[wide_fwd]
                gx = torch.empty_like(g16)
                _lib.check(L.fq_ste_bwd_mask(g16.data_ptr(), gx.data_ptr(), rows, cols, lo, hi, sd.data_ptr(), sd.data_ptr() + rows * 8, mb, code, st), "bwd")
                assert bits_equal(np_from(gx), want_g, dtype), f"narrow backward on a wide={wide_fwd} forward's mask, {(rows, cols)} clip=({lo},{hi})"
                gx = torch.empty_like(g16)
                _lib.check(L.fq_ste_bwd_mask_wide(g32.data_ptr(), gx.data_ptr(), rows, sd.data_ptr(), sd.data_ptr() + rows * 8, None, None, 0, None, None,
                                                  cols, lo, hi, code, st), "bwd wide")
                assert bits_equal(np_from(gx), want_g, dtype), f"wide backward on a wide={wide_fwd} forward's mask, {(rows, cols)} clip=({lo},{hi})"


def test_tensors_beyond_2_31_elements(ops):
    """64-bit indexing everywhere: a layerwise row of 2^31 + 4096 elements (two-pass kernels, plain STE) and a row-wise
    tensor of 2.1e9 elements (register kernels + mask backward), against ATen on the same GPU."""
    import llm_qat_amd
    from oracle import eager_chain as E
    llm_qat_amd.set_semantics("device_eager")
    try:
        g = torch.Generator(device="cuda").manual_seed(0)
        n = 2 ** 31 + 4096
        x = torch.randn(n, generator=g, device="cuda", dtype=torch.bfloat16)
        x[n - 5] = 7.5   # the global max sits beyond the 2^31 boundary
        y = ops.sym_quantize(x, 8, True)
        assert torch.equal(y, E.sym_forward(x, 8, layerwise=True)) and float(y[n - 5]) > 7.0
        del y
        gr = torch.randn(n, generator=g, device="cuda", dtype=torch.bfloat16)
        assert torch.equal(ops.ste_backward(gr, x, -2.0, 2.0), torch.where((x >= 2.0) | (x <= -2.0), torch.zeros_like(gr), gr))
        del x, gr
        torch.cuda.empty_cache()
        rows, cols = 2 ** 20 + 3, 2048 + 8
        x = torch.randn(rows, cols, generator=g, device="cuda", dtype=torch.bfloat16)
        y, bounds, mask = ops.quantize_train("sym", x, 4, False, -2.0, 2.0)
        sel = torch.tensor([0, 1, rows // 2, rows - 2, rows - 1], device="cuda")
        assert torch.equal(y[sel], E.sym_forward(x[sel], 4))
        del y
        gr = torch.randn(rows, cols, generator=g, device="cuda", dtype=torch.bfloat16)
        gx = ops.ste_backward_mask(gr, -2.0, 2.0, bounds, mask, rows, cols)
        assert torch.equal(gx, torch.where((x >= 2.0) | (x <= -2.0), torch.zeros_like(gr), gr))
    finally:
        llm_qat_amd.set_semantics("cpu_eager")
        torch.cuda.empty_cache()
