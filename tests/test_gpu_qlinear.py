"""GPU tier: SURVEY §8 f4a -- the quantize-on-load GEMM EXPERIMENT under tools/qlinear/ (not part of the product library since
round 4: measured slower than the unfused product path): QuantizeLinear's no-grad forward with the fake-quant
fused into the GEMM's operand loads.
  * the MFMA path itself on exact integer data (asymmetric operands, tails in every dimension): bit-exact vs fp32 matmul
  * the operand tiles as staged for the MFMAs, dumped: bit-identical to fq_sym_fwd / fq_sym_fwd_autocast(narrow)
  * the product: within the stated fp32-accumulation tolerance of the exact product of those operands, and of
    F.linear(fq(x), fq(W)) as the unfused path computes it
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as O

from conftest import experiment_module

pytestmark = pytest.mark.gpu


class _Lazy:
    """tools/qlinear/qlinear.py, loaded on first use: the tests SKIP when the experiment library cannot be built or loaded"""
    _mod = None

    def __getattr__(self, name):
        if _Lazy._mod is None:
            _Lazy._mod = experiment_module("tools", "qlinear", "qlinear.py")
        return getattr(_Lazy._mod, name)


QX = _Lazy()


def bits(t):
    return t.detach().contiguous().cpu().view(torch.int16).numpy().view(np.uint16)


def oracle_operand(t, nbits, autocast, rows=None):
    """the CPU oracle's fake-quantized operand (models/utils_quant.py:71-72; under autocast the fp32 chain rounded once to bf16,
    which is what F.linear's autocast cast makes of the reference's fp32 result), optionally on a row sample"""
    t = t if rows is None else t[rows]
    r, c = t.shape
    if autocast:
        return O.sym_fwd_autocast(bits(t), r, c, nbits, "bf16", wide=False)[0].reshape(r, c)
    return O.sym_fwd(bits(t), r, c, nbits, "bf16", want_idx=False)[0].reshape(r, c)


@pytest.fixture(scope="module")
def ops():
    import llm_qat_amd
    from llm_qat_amd import _lib
    _lib.lib()
    llm_qat_amd.set_semantics("cpu_eager")
    return llm_qat_amd.ops


SHAPES = [(256, 128, 64), (300, 388, 192), (1, 4, 64), (513, 260, 640), (2048, 512, 1024)]   # (tokens, out, in)


@pytest.mark.parametrize("shape", SHAPES)
def test_mfma_path_exact_on_integer_data(ops, shape):
    """values in {-2..2}: every partial sum is an exact small integer, so ANY accumulation order gives the same bits; a
    swapped row/column map, a wrong k-order inside a fragment or a wrong swizzle would not.  Asymmetric operands."""
    m, n, k = shape
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randint(-2, 3, (m, k), generator=g, device="cuda").to(torch.bfloat16)
    w = torch.randint(-2, 3, (n, k), generator=g, device="cuda").to(torch.bfloat16)
    w[:, 0] = 1
    x[0, :] = 2      # asymmetric: a row of x and a column of W stand out
    out = QX.qlinear_forward(x, w, 8, 8, quantize_x=False, quantize_w=False, autocast=False)
    assert out is not None and out.shape == (m, n)
    want = (x.float() @ w.float().t())
    assert want.abs().max() <= 256 * 16
    assert torch.equal(out.float(), want.to(torch.bfloat16).float())


@pytest.mark.parametrize("autocast", [False, True])
@pytest.mark.parametrize("shape", SHAPES)
def test_staged_tiles_are_bit_identical_to_the_quantizer(ops, shape, autocast):
    m, n, k = shape
    g = torch.Generator(device="cuda").manual_seed(2)
    x = torch.randn(m, k, generator=g, device="cuda")
    x[torch.rand(m, k, generator=g, device="cuda") < 1e-3] *= 20
    x = x.to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    out, sx, sw = QX.qlinear_forward(x, w, 4, 8, autocast=autocast, dump=True)
    if autocast:
        xq = ops.sym_forward_autocast(x, 8, False, wide=False)[0]
        wq = ops.sym_forward_autocast(w, 4, False, wide=False)[0]
    else:
        xq, wq = ops.sym_quantize(x, 8), ops.sym_quantize(w, 4)
    # the tiles as the MFMAs saw them against the ORACLE itself (not only against another HIP kernel)
    assert (bits(sx) == oracle_operand(x, 8, autocast)).all(), "staged x tiles != oracle"
    assert (bits(sw) == oracle_operand(w, 4, autocast)).all(), "staged W tiles != oracle"
    assert torch.equal(sx.view(torch.int16), xq.view(torch.int16)), "staged x tiles != fq_sym_fwd(x)"
    assert torch.equal(sw.view(torch.int16), wq.view(torch.int16)), "staged W tiles != fq_sym_fwd(W)"
    # the product of exactly those operands.  Tolerance: the result is the fp32-accumulated sum rounded once to bf16;
    # against the fp64 product that is <= 2^-9 relative (half a bf16 ulp) + the fp32 accumulation error
    # (<= K * 2^-24 * sum|a_i b_i|)
    ref = xq.double() @ wq.double().t()
    mag = xq.double().abs() @ wq.double().abs().t()
    tol = ref.abs() * 2.0 ** -8 + mag * k * 2.0 ** -24 + 1e-30
    assert ((out.double() - ref).abs() <= tol).all(), float(((out.double() - ref).abs() / tol).max())
    # and the unfused path's own product of the same operands (hipBLASLt, its own accumulation order): <= 1 bf16 ulp apart
    unf = F.linear(xq, wq)
    assert ((out.double() - unf.double()).abs() <= ref.abs() * 2.0 ** -7 + mag * k * 2.0 ** -23 + 1e-30).all()
    # mixed: only W quantized on load, x handed over already quantized == both on load, bit for bit (same staged values)
    out2 = QX.qlinear_forward(xq, w, 4, 8, quantize_x=False, autocast=autocast)
    assert torch.equal(out2.view(torch.int16), out.view(torch.int16))


def test_llama7b_shape_full_size(ops):
    """[2048, 11008] x [4096, 11008]^T, W4 A8 (down_proj): staged operands bit-identical; the product a bf16 ulp from hipBLASLt's"""
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(2048, 11008, generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn(4096, 11008, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    out, sx, sw = QX.qlinear_forward(x, w, 4, 8, autocast=False, dump=True)
    xq, wq = ops.sym_quantize(x, 8), ops.sym_quantize(w, 4)
    assert torch.equal(sx.view(torch.int16), xq.view(torch.int16)) and torch.equal(sw.view(torch.int16), wq.view(torch.int16))
    # sampled rows of both staged operands against the oracle (incl. the first / last row of every 256- / 128-row tile edge)
    rx = torch.unique(torch.cat([torch.arange(0, 2048, 61), torch.tensor([0, 255, 256, 2047])])).cuda()
    rw = torch.unique(torch.cat([torch.arange(0, 4096, 97), torch.tensor([0, 127, 128, 4095])])).cuda()
    assert (bits(sx[rx]) == oracle_operand(x, 8, False, rx)).all(), "staged x rows != oracle"
    assert (bits(sw[rw]) == oracle_operand(w, 4, False, rw)).all(), "staged W rows != oracle"
    out_ac, sx_ac, sw_ac = QX.qlinear_forward(x, w, 4, 8, autocast=True, dump=True)
    assert (bits(sx_ac[rx]) == oracle_operand(x, 8, True, rx)).all() and (bits(sw_ac[rw]) == oracle_operand(w, 4, True, rw)).all(), "autocast staging != oracle"
    unf = F.linear(xq, wq).float()
    err = (out.float() - unf).abs()
    scale = unf.abs().mean()
    assert float(err.max()) <= float(unf.abs().max()) * 2.0 ** -6 and float(err.mean()) <= float(scale) * 2.0 ** -8


def test_unserved_shapes_return_none(ops):
    x = torch.randn(8, 100, device="cuda").to(torch.bfloat16)
    w = torch.randn(16, 100, device="cuda").to(torch.bfloat16)
    assert QX.qlinear_forward(x, w, 4, 8) is None                       # in_features % 64 != 0
    assert QX.qlinear_forward(x.float(), w.float(), 4, 8) is None       # bf16 only


# ---- round 5: the second attempt (tools/qlinear/fq_qlinear_direct.hip): W direct-to-VGPR as the MFMA operand, x by LDS-DMA.  A measured
# no-go like the first (profiles/r05_qlinear_direct_operand.json); what is pinned here is that the experiment is CORRECT.
def _direct(x, w, ws=None, bk=64, mfma=32, ac=0, abl=0):
    out = torch.full((x.shape[0], w.shape[0]), float("nan"), dtype=torch.bfloat16, device=x.device)
    rc = QX.lib().fq_qlinear_direct_fwd(x.data_ptr(), w.data_ptr(), ws.data_ptr() if ws is not None else None, out.data_ptr(), x.shape[0], x.shape[1], w.shape[0],
                                        bk, mfma, ac, abl, torch.cuda.current_stream().cuda_stream)
    QX.check(rc, "fq_qlinear_direct_fwd")
    return out


@pytest.mark.parametrize("mfma", [32, 16])
@pytest.mark.parametrize("bk", [64, 128])
@pytest.mark.parametrize("shape", [(128, 256, 384), (200, 300, 768), (77, 36, 512), (513, 260, 1152)])   # (tokens, out, in): every K remainder mod 3
def test_direct_operand_kernel_exact_on_integer_data(ops, shape, bk, mfma):
    """small integers: every partial sum is exact, so any accumulation order gives the same bits -- a wrong k re-mapping between the W
    loads and the x fragment reads, a wrong source-side swizzle of the LDS-DMA, a swapped fragment map or a lost K-step would not"""
    m, n, k = shape
    if k // bk < 3:
        pytest.skip("the pipeline needs three K-steps")
    g = torch.Generator(device="cuda").manual_seed(m + n + k)
    x = torch.randint(-3, 4, (m, k), generator=g, device="cuda").float()
    w = torch.randint(-2, 3, (n, k), generator=g, device="cuda").float()
    w[:, 0] += torch.arange(n, device="cuda") % 3
    x[:, 1] += torch.arange(m, device="cuda") % 2
    ref = (x @ w.t()).bfloat16()
    for abl in (0, 10, 20):   # plain / `nt` / `sc1` W loads
        got = _direct(x.bfloat16(), w.bfloat16(), bk=bk, mfma=mfma, abl=abl)
        assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), (shape, bk, mfma, abl)


@pytest.mark.parametrize("mfma", [32, 16])
@pytest.mark.parametrize("autocast", [False, True])
def test_direct_operand_kernel_quantizes_w_like_the_product(ops, mfma, autocast):
    """W fake-quantized in registers on its way into the MFMA == the GEMM over the product's own fq_sym_fwd(W): with integer x the only
    difference left is the fp32 accumulation order"""
    m, n, k = 192, 320, 11008
    g = torch.Generator(device="cuda").manual_seed(11)
    w = (torch.randn(n, k, generator=g, device="cuda") * 0.02).bfloat16()
    x = torch.randint(-2, 3, (m, k), generator=g, device="cuda").bfloat16()
    ws = ops.sym_row_scales(w, 4, False, autocast=autocast)
    if autocast:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            wq = ops.sym_forward_autocast(w, 4, False, wide=False)[0]
    else:
        wq = ops.sym_quantize(w, 4)
    got = _direct(x, w, ws=ws, bk=64, mfma=mfma, ac=int(autocast)).double()
    ref = x.double() @ wq.double().t()
    tol = 2.0 ** -8 * ref.abs() + k * 2.0 ** -24 * (x.double().abs() @ wq.double().abs().t())
    assert bool(((got - ref).abs() <= tol).all())
