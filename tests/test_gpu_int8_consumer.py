"""GPU tier: the integer consumer of the export format (tools/int8_linear -- a measurement under tools/, not the product).

What is pinned here:
  * the int8 x int8 -> int32 GEMM over the exported bins is EXACT integer arithmetic (checked against int64 matmul on the CPU);
  * the epilogue is bit-identical to its stated formula bf16(float(acc) * ((1/t2_x[m]) * (1/t2_w[n]))) evaluated by ATen;
  * the result equals the exact real-number product of the fake-quantized operands' UNROUNDED values bin/t2 (fp64) to within one bf16
    rounding of the output, and lies within the stated distance of the reference module's forward (which rounds every operand to bf16
    once more before its bf16 GEMM) -- not bit-identical, and the test says by how much;
  * a bin the int8 container cannot hold (+128 of an 8-bit bf16 row: the reference has no clamp; -128 fits) is saturated and counted.
"""
import pytest
import torch

from conftest import experiment_module

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def I8():
    return experiment_module("tools", "int8_linear", "int8_linear.py")   # skips (never fails) when the experiment cannot be built


@pytest.mark.parametrize("w_bits,a_bits", [(4, 8), (8, 8), (4, 4)])
def test_int8_consumer_of_the_export_format(I8, w_bits, a_bits):
    from llm_qat_amd.utils_quant import QuantizeLinear
    torch.manual_seed(w_bits * 10 + a_bits)
    K, N = 1024, 512
    lin = QuantizeLinear(K, N, w_bits=w_bits, a_bits=a_bits).cuda().bfloat16()
    with torch.no_grad():
        lin.weight.mul_(0.05 / lin.weight.std())
    x = torch.randn(3, 40, K, device="cuda").bfloat16()
    x[0, 3, 7] = 25.0   # an outlier channel
    q = I8.Int8Linear(lin)
    with torch.no_grad():
        out, ex, acc = q(x, return_parts=True)
        ref = lin(x)
    assert out.shape == ref.shape and out.dtype == torch.bfloat16
    bx, bw = ex.bins.cpu().to(torch.int64), q.w_bins.cpu().to(torch.int64)
    assert torch.equal(acc.cpu().to(torch.int64), bx @ bw.t()), "int8 GEMM over the bins is exact"
    # the epilogue == its formula, evaluated by ATen in the same order
    f = (1.0 / ex.scales[:, 1])[:, None] * (1.0 / q.w_scales[:, 1])[None, :]
    want = (acc.to(torch.float32) * f).to(torch.bfloat16).reshape(out.shape)
    assert torch.equal(out, want), "epilogue differs from bf16(float(acc) * (rx * rw))"
    # distance from the exact product of the unrounded fake-quant values bin / t2 (fp64): one output rounding
    yx = bx.double() / ex.scales[:, 1].cpu().double()[:, None]
    yw = bw.double() / q.w_scales[:, 1].cpu().double()[:, None]
    exact = (yx @ yw.t()).reshape(out.shape)
    scale = exact.abs().clamp_min(exact.abs().mean())
    assert ((out.cpu().double() - exact).abs() / scale).max() < 2.0 ** -8, "more than one bf16 rounding from the exact product"
    # distance from the reference module's forward (operands rounded to bf16, bf16 GEMM): small, not zero
    if int(ex.overflow.sum()) == 0 and q.w_overflow == 0:
        d = (out.double() - ref.double()).cpu()
        rel_rms = float(d.pow(2).mean().sqrt() / ref.double().pow(2).mean().sqrt())
        assert rel_rms < 6e-3, rel_rms           # measured 3.3e-3 at LLaMA-7B's shapes (tools/int8_linear/int8_linear_bench.py); bf16 eps = 3.9e-3


def test_bins_beyond_int8_are_saturated_and_counted(I8):
    """8-bit bf16 rows reach bin +-128 when bf16(x * s) rounds up (no clamp in the reference): -128 is an int8, +128 is not --
    the deployment container saturates that element to +127 and says so; everything else in the row is untouched."""
    from llm_qat_amd import ops
    from llm_qat_amd.utils_quant import QuantizeLinear
    torch.manual_seed(5)
    K = 512
    lin = QuantizeLinear(K, 256, w_bits=4, a_bits=8).cuda().bfloat16()
    x = torch.randn(64, K, device="cuda").bfloat16()
    wide = ops.sym_export(x, 8, False, container="int16", autocast=False)
    rows128 = (wide.unpacked() == 128).any(dim=1)
    if not bool(rows128.any()) or not bool((wide.unpacked() == -128).any()):
        pytest.skip("no row of this sample reaches +128 / -128")
    q = I8.Int8Linear(lin)
    ex = q.export_input(x)
    assert torch.equal(ex.overflow > 0, rows128)
    assert torch.equal(ex.unpacked(), wide.unpacked().clamp(-128, 127))
