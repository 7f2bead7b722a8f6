"""float64 tensors (the reference has no dtype restriction; models/utils_quant.py:37-74, :96-149, :77-87, :202-242).

CPU tier: oracle/fq_oracle_f64.c against tests/golden/f64.npz, the vectors the real reference produced on float64 inputs
(tests/golden/make_golden_f64.py) -- this pins the float64 oracle.
GPU tier: the float64 kernels (llm-qat_amd/csrc/fq_f64.hip, a correctness path in double arithmetic) through the C ABI and through
the drop-in classes against the fixtures, the oracle on seeded inputs, and live ATen on the device.  Bar: bit-exact (any NaN equals
any NaN)."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import oracle as O


def eq64(a, b):
    a, b = np.asarray(a, np.float64).reshape(-1).view(np.uint64), np.asarray(b, np.float64).reshape(-1).view(np.uint64)
    if a.shape != b.shape:
        return False
    na, nb = (a & 0x7FFFFFFFFFFFFFFF) > 0x7FF0000000000000, (b & 0x7FFFFFFFFFFFFFFF) > 0x7FF0000000000000
    return bool((na == nb).all() and ((a == b) | (na & nb)).all())


def test_f64_oracle_matches_reference_fixtures():
    G = golden("f64.npz")
    assert len(G.cases) == 50 and G.meta["dtype"] == "float64"
    n = 0
    for c in G.cases:
        k = c["kind"]
        if k in ("sym", "asym"):
            rows, cols = O.rows_cols(c["shape"], c["layerwise"])
            r = O.sym_fwd(G.arr(c, "x"), rows, cols, c["bits"], "fp64") if k == "sym" else O.asym_fwd(G.arr(c, "x"), rows, cols, c["bits"], "fp64")
            assert (r[1].reshape(-1) == G.arr(c, "idx").reshape(-1)).all(), f"{c['name']}: bin indices"
            assert eq64(r[0], G.arr(c, "y")), c["name"]
        elif k == "ste":
            assert eq64(O.ste_bwd(G.arr(c, "g"), G.arr(c, "x"), c["lo"], c["hi"], "fp64"), G.arr(c, "gx")), c["name"]
        elif k == "w12":
            rows, cols = (1, 240) if c["layerwise"] else (6, 40)
            q, _ = O.w12_fwd(G.arr(c, "w"), rows, cols, c["w_bits"], "fp64", scale_in=G.arr(c, "scale"))
            assert eq64(q, G.arr(c, "wq")), c["name"]
        else:
            continue
        n += 1
    assert n == 49


@pytest.fixture(scope="module")
def pkg():
    import llm_qat_amd
    from llm_qat_amd import _lib
    _lib.lib()
    llm_qat_amd.set_semantics("cpu_eager")
    return llm_qat_amd


@pytest.mark.gpu
def test_f64_kernels_match_reference_fixtures(pkg):
    from llm_qat_amd.utils_quant import AsymQuantizer, QuantizeLinear, SymQuantizer
    G = golden("f64.npz")
    clip = torch.tensor([-2.0, 2.0])
    for c in G.cases:
        k = c["kind"]
        if k in ("sym", "asym"):
            x = torch.from_numpy(G.arr(c, "x")).cuda()
            fn = pkg.ops.sym_quantize_debug if k == "sym" else pkg.ops.asym_quantize_debug
            y, idx, _ = fn(x, c["bits"], c["layerwise"])
            assert y.dtype == torch.float64 and eq64(y.cpu().numpy(), G.arr(c, "y")), c["name"]
            assert (idx.cpu().numpy().reshape(-1) == G.arr(c, "idx").reshape(-1)).all(), c["name"]
            q = (SymQuantizer if k == "sym" else AsymQuantizer).apply(x, clip, c["bits"], c["layerwise"])
            assert eq64(q.cpu().numpy(), G.arr(c, "y")), c["name"] + " (apply)"
        elif k == "ste":
            x = torch.from_numpy(G.arr(c, "x")).cuda().requires_grad_(True)
            g = torch.from_numpy(G.arr(c, "g")).cuda()
            SymQuantizer.apply(x, torch.tensor([c["lo"], c["hi"]]), 8, False).backward(g)
            assert eq64(x.grad.cpu().numpy(), G.arr(c, "gx")), c["name"]
        elif k == "w12":
            w = torch.from_numpy(G.arr(c, "w")).cuda()
            sc = torch.from_numpy(G.arr(c, "scale")).cuda()
            q = pkg.ops.low_bit_weight(w, sc if not c["layerwise"] else sc.reshape(()), c["w_bits"])
            assert eq64(q.cpu().numpy(), G.arr(c, "wq")), c["name"]
        elif k == "qlinear":   # module level: operands exact, the float64 GEMM within a few ulps of the CPU's accumulation order
            lin = QuantizeLinear(64, 16, w_bits=4, a_bits=8).cuda().double()
            with torch.no_grad():
                lin.weight.copy_(torch.from_numpy(G.arr(c, "w")))
            x = torch.from_numpy(G.arr(c, "x")).cuda().requires_grad_(True)
            out = lin(x)
            out.square().sum().backward()
            for got, key in ((out, "out"), (lin.weight.grad, "gw"), (x.grad, "gx")):
                want = torch.from_numpy(G.arr(c, key)).cuda()
                assert got.dtype == torch.float64 and torch.allclose(got, want, rtol=1e-12, atol=1e-12 * float(want.abs().max())), (c["name"], key)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sym", "asym"])
def test_f64_vs_oracle_and_live_aten(pkg, kind):
    """seeded inputs over odd and long rows, 1-D ... 4-D, layerwise, NaN / Inf rows: C ABI vs the oracle, and the drop-in classes
    (values and gradients) vs the reference's op chain executed by ATen in float64 on the device"""
    from llm_qat_amd.utils_quant import AsymQuantizer, SymQuantizer
    rng = np.random.default_rng(640)
    Q = SymQuantizer if kind == "sym" else AsymQuantizer

    def chain(x, bits, lw):   # models/utils_quant.py:50-72 / :110-147, restated for the live comparison
        if kind == "sym":
            m = (torch.max(torch.abs(x)) if lw else torch.max(torch.abs(x.reshape(x.shape[0], x.shape[1], -1) if x.dim() == 4 else x), dim=-1, keepdim=True)[0])
            m = m.unsqueeze(-1).expand_as(x) if (x.dim() == 4 and not lw) else m.expand_as(x)
            s = (2 ** (bits - 1) - 1) / (m + 1e-6)
            return torch.round(x * s).div(s + 1e-6)
        t = x.reshape(x.shape[0], x.shape[1], -1) if (x.dim() == 4 and not lw) else x
        mx, mn = (x.max(), x.min()) if lw else (t.max(dim=-1, keepdim=True)[0], t.min(dim=-1, keepdim=True)[0])
        if x.dim() == 4 and not lw:
            mx, mn = mx.unsqueeze(-1), mn.unsqueeze(-1)
        alpha, beta = (mx - mn).expand_as(x), mn.expand_as(x)
        s = 2 ** bits - 1
        return torch.round((x - beta) / (alpha + 1e-8) * s).div(s) * (alpha + 1e-8) + beta

    for shape in [(7,), (3, 5), (4, 1000), (2, 3, 129), (2, 2, 3, 8), (3, 20000)]:
        for bits in (3, 8, 16):
            for lw in (False, True):
                x = rng.standard_normal(shape) * rng.choice([1e-6, 0.02, 1.0, 50.0, 1e200])
                if len(shape) >= 2 and shape[0] > 2:
                    x[1].reshape(-1)[0] = np.nan
                    x[2].reshape(-1)[-1] = np.inf
                xt = torch.from_numpy(x).cuda()
                rows, cols = O.rows_cols(shape, lw)
                want = (O.sym_fwd(x, rows, cols, bits, "fp64") if kind == "sym" else O.asym_fwd(x, rows, cols, bits, "fp64", sem=O.SEM_CPU))[0]
                got = (pkg.ops.sym_quantize if kind == "sym" else pkg.ops.asym_quantize)(xt, bits, lw)
                assert eq64(got.cpu().numpy(), want), (kind, shape, bits, lw)
                # live ATen on the device (device-eager semantics: `.div(python int)` multiplies by the reciprocal)
                pkg.set_semantics("device_eager")
                try:
                    xr = xt.clone().requires_grad_(True)
                    out = Q.apply(xr, torch.tensor([-0.5, 0.75]), bits, lw)
                    g = torch.from_numpy(rng.standard_normal(shape)).cuda()
                    out.backward(g)
                    assert eq64(out.detach().cpu().numpy(), chain(xt, bits, lw).cpu().numpy()), (kind, shape, bits, lw, "live ATen")
                    ref_g = g.clone()
                    ref_g[xt.ge(0.75)] = 0
                    ref_g[xt.le(-0.5)] = 0
                    assert eq64(xr.grad.cpu().numpy(), ref_g.cpu().numpy()), (kind, shape, bits, lw, "grad")
                finally:
                    pkg.set_semantics("cpu_eager")


@pytest.mark.gpu
def test_f64_low_bit_module_vs_live_aten(pkg):
    from llm_qat_amd.utils_quant import QuantizeLinear
    from test_gpu_features import eager_low_bit
    g = torch.Generator(device="cuda").manual_seed(6)
    for w_bits in (1, 2):
        for lw in (False, True):
            lin = QuantizeLinear(100, 33, w_bits=w_bits, a_bits=32, weight_layerwise=lw).cuda().double()
            with torch.no_grad():
                lin.weight.copy_(torch.randn(33, 100, generator=g, device="cuda", dtype=torch.float64) * 0.05)
            x = torch.randn(3, 100, generator=g, device="cuda", dtype=torch.float64)
            wref = lin.weight.detach().clone().requires_grad_(True)
            out, ref = lin(x), torch.nn.functional.linear(x, eager_low_bit(wref, w_bits, lw))
            assert torch.equal(out, ref)
            out.sum().backward()
            ref.sum().backward()
            assert torch.equal(lin.weight.grad, wref.grad)
    # what float64 does not serve fails loudly, never silently
    with pytest.raises(RuntimeError):
        pkg.ops.sym_export(torch.randn(4, 64, device="cuda", dtype=torch.float64), 8)
