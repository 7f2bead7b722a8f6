"""GPU tier: SURVEY §8 f4 -- the integer side of the forward through the C ABI:
  fq_sym_export / fq_asym_export   packed int4 / int8 / int16 bins + per-row scales + saturation counts
  fq_sym_row_scales                the scale pre-pass of the fused QuantizeLinear GEMM
Integer work => bit-exact bar: bins == the reference's own `idx` fixtures and == the CPU oracle; scales bit-equal;
dequantising the export reproduces the fake-quant forward bit for bit (up to the sign of zero) where overflow == 0.
"""
import numpy as np
import pytest
import torch

from conftest import bits_equal, golden, to_f32
from oracle import oracle as O
from test_gpu_parity import TD, dev_from, np_from

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import llm_qat_amd
    from llm_qat_amd import _lib
    _lib.lib()
    llm_qat_amd.set_semantics("cpu_eager")
    return llm_qat_amd.ops


def _crange(container, signed):
    cb = {"int4": 4, "int8": 8, "int16": 16}[container]
    return (-(1 << (cb - 1)), (1 << (cb - 1)) - 1) if signed else (0, (1 << cb) - 1)


@pytest.mark.parametrize("kind", ["sym", "asym"])
def test_export_golden(ops, kind):
    """every reference fixture, every container: bins == clip(reference idx), overflow == #(idx outside), scales bit-equal"""
    G = golden(f"{kind}_fwd.npz")
    fn = ops.sym_export if kind == "sym" else ops.asym_export
    failures = []
    for c in G.cases:
        dt, bits = c["dtype"], c["bits"]
        x = dev_from(G.arr(c, "x"), dt)
        rows, cols = O.rows_cols(c["shape"], c["layerwise"])
        idx = G.arr(c, "idx").reshape(rows, cols).astype(np.int64)
        nan = idx == np.iinfo(np.int32).min
        for container in ("int4", "int8", "int16"):
            lo, hi = _crange(container, kind == "sym")
            e = fn(x, bits, c["layerwise"], container=container) if kind == "asym" else fn(x, bits, c["layerwise"], container=container, autocast=False)
            got = e.unpacked().cpu().numpy()
            want = np.where(nan, 0, np.clip(idx, lo, hi))
            if not (got == want).all():
                failures.append(f"{c['name']} {container}: {(got != want).sum()} bins differ")
            bad = (nan | (idx < lo) | (idx > hi)).sum(axis=1)
            if not (e.overflow.cpu().numpy() == bad).all():
                failures.append(f"{c['name']} {container}: overflow {e.overflow.cpu().numpy().tolist()[:4]} vs {bad.tolist()[:4]}")
        sc = e.scales.cpu().numpy()
        if kind == "sym":
            if not bits_equal(sc[:, 0].copy(), to_f32(G.arr(c, "scale"), dt).reshape(-1), "fp32"):
                failures.append(f"{c['name']}: s")
    assert not failures, "\n".join(failures[:20])


SHAPES = [(1, 8), (3, 7), (5, 33), (4, 255), (64, 256), (9, 688), (3, 4096), (2, 11008), (2, 13824), (3, 1000), (1, 65536), (2, 100003)]


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("kind", ["sym", "asym"])
def test_export_vs_oracle(ops, kind, dtype):
    """seeded inputs over register-path, generic-path (odd widths, misaligned) and long rows, bits 3/4/8/16:
    packed bytes, scales and overflow counts bit-identical to the CPU oracle"""
    rng = np.random.default_rng(11)
    fn = ops.sym_export if kind == "sym" else ops.asym_export
    for rows, cols in SHAPES:
        x32 = rng.standard_normal((rows, cols)).astype(np.float32) * rng.choice([0.02, 1.0, 30.0])
        x32[0, 0] = np.abs(x32).max() * 1.0        # make sure a row max exists at a known place
        xt = torch.from_numpy(x32).cuda().to(TD[dtype])
        x_np = np_from(xt)
        for bits in (3, 4, 8, 16):
            for container in ("int4", "int8", "int16"):
                e = fn(xt, bits, False, container=container) if kind == "asym" else fn(xt, bits, False, container=container, autocast=False)
                ob, osc, oov = O.export(kind, x_np, rows, cols, bits, container, dtype)
                raw = e.bins.contiguous().view(torch.uint8).reshape(rows, -1).cpu().numpy()
                assert (raw == ob).all(), f"{kind} {dtype} {(rows, cols)} b{bits} {container}: {(raw != ob).sum()} bytes differ"
                assert (e.overflow.cpu().numpy() == oov).all(), f"{kind} {dtype} {(rows, cols)} b{bits} {container}: overflow"
                assert bits_equal(e.scales.cpu().numpy(), osc, "fp32"), f"{kind} {dtype} {(rows, cols)} b{bits}: scales"


def test_export_misaligned_and_noncontiguous(ops):
    rng = np.random.default_rng(12)
    flat = torch.from_numpy(rng.standard_normal(4096 * 3 + 1).astype(np.float32)).cuda().bfloat16()
    xm = flat[1:].view(3, 4096)                      # contiguous, 2-byte aligned only -> generic kernel
    e = ops.sym_export(xm, 8, container="int8", autocast=False)
    ob, osc, oov = O.export("sym", np_from(xm), 3, 4096, 8, "int8", "bf16")
    assert (e.bins.view(torch.uint8).cpu().numpy() == ob).all() and (e.overflow.cpu().numpy() == oov).all()
    xt = torch.from_numpy(rng.standard_normal((64, 48)).astype(np.float32)).cuda().bfloat16().t()   # non-contiguous
    e = ops.sym_export(xt, 4, container="int4", autocast=False)
    ob, _, _ = O.export("sym", np_from(xt), 48, 64, 4, "int4", "bf16")
    assert (e.bins.cpu().numpy() == ob).all()


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp32"])
def test_export_dequant_reproduces_fq_sym_fwd(ops, dtype):
    """dequantising the export (bins / t2, rounded to the tensor dtype) == fq_sym_fwd, bit for bit, wherever overflow == 0;
    likewise Asym with its three roundings.  W4 / A8 / KV4 model-style tensors."""
    g = torch.Generator(device="cuda").manual_seed(5)
    w = (torch.randn(512, 4096, generator=g, device="cuda") * 0.02).to(TD[dtype])
    a = (torch.randn(256, 11008, generator=g, device="cuda")).to(TD[dtype])
    a[torch.rand(256, 11008, generator=g, device="cuda") < 1e-3] *= 20
    for x, bits in ((w, 4), (w, 8), (a, 8), (a, 4)):
        e = ops.sym_export(x, bits, autocast=False)          # default container: lossless
        assert int(e.overflow.sum()) == 0, (dtype, bits, e.container)
        y = ops.sym_quantize(x, bits)
        d = e.dequantize()
        assert torch.equal(torch.where(d == 0, torch.zeros_like(d), d), torch.where(y == 0, torch.zeros_like(y), y)), (dtype, bits)
        e8 = ops.sym_export(x, bits, container="int8", autocast=False)   # the deployment container: saturates the +128 bin, says so
        ok = e8.overflow == 0
        d8 = e8.dequantize()
        assert torch.equal(torch.where(d8 == 0, torch.zeros_like(d8), d8)[ok], torch.where(y == 0, torch.zeros_like(y), y)[ok])
        if bits == 8 and dtype == "bf16":
            _, idx, _ = ops.sym_quantize_debug(x, bits)
            assert torch.equal(e8.overflow, (idx > 127).sum(dim=1).to(torch.int32))      # exactly the +128 bins, -128 fits
        ea = ops.asym_export(x, bits)
        assert int(ea.overflow.sum()) == 0
        ya = ops.asym_quantize(x, bits)
        da = ea.dequantize()
        assert torch.equal(torch.where(da == 0, torch.zeros_like(da), da), torch.where(ya == 0, torch.zeros_like(ya), ya)), ("asym", dtype, bits)


def test_export_autocast_bins_match_live_aten(ops):
    """under torch.autocast the reference computes the bins in fp32 (reciprocal is on autocast's fp32 list): the export
    follows -- checked against the live ATen chain on this GPU (no reference fixture can exist: CPU autocast differs)"""
    g = torch.Generator(device="cuda").manual_seed(6)
    for dt in (torch.bfloat16, torch.float16):
        x = torch.randn(128, 4096, generator=g, device="cuda").to(dt)
        for bits in (4, 8):
            with torch.autocast("cuda", dtype=dt):
                m = torch.max(torch.abs(x), dim=-1, keepdim=True)[0].expand_as(x)
                s = (2 ** (bits - 1) - 1) / (m + 1e-6)
                assert s.dtype == torch.float32
                want = torch.round(x * s)
                e = ops.sym_export(x, bits, container="int16")      # autocast=None: follows the ambient autocast state
            assert torch.equal(e.unpacked().float().view_as(want), want)
            assert torch.equal(e.scales[:, 0], s[:, 0]) and torch.equal(e.scales[:, 1], (s + 1e-6)[:, 0])
            e0 = ops.sym_export(x, bits, container="int16", autocast=False)
            assert not torch.equal(e0.unpacked(), e.unpacked())       # the two arithmetics really differ on some bins


def test_row_scales_prepass(ops):
    """fq_sym_row_scales == the scale terms of the forward: bit-equal to the `scale` arrays of sym_fwd.npz, to the oracle,
    and (with bounds + mask requested) its side outputs equal fq_sym_fwd_train's"""
    G = golden("sym_fwd.npz")
    for c in G.cases:
        dt = c["dtype"]
        x = dev_from(G.arr(c, "x"), dt)
        sc = ops.sym_row_scales(x, c["bits"], c["layerwise"], autocast=False).cpu().numpy()
        assert bits_equal(sc[:, 0].copy(), to_f32(G.arr(c, "scale"), dt).reshape(-1), "fp32"), c["name"]
    from llm_qat_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(7)
    x = (torch.randn(300, 4096, generator=g, device="cuda") * 1.1).bfloat16()
    rows, cols = x.shape
    mb = L.fq_ste_mask_bytes(rows, cols, _lib.DTYPE_BF16)
    st = torch.cuda.current_stream().cuda_stream
    b0, m0 = torch.empty(rows, 2, device="cuda"), torch.zeros(mb, dtype=torch.uint8, device="cuda")
    b1, m1 = torch.empty(rows, 2, device="cuda"), torch.zeros(mb, dtype=torch.uint8, device="cuda")
    y = torch.empty_like(x)
    sc = torch.empty(rows, 2, device="cuda")
    assert L.fq_sym_fwd_train(x.data_ptr(), y.data_ptr(), rows, cols, 8, _lib.DTYPE_BF16, 0, -2.0, 2.0, b0.data_ptr(), m0.data_ptr(), mb, st) == 0
    assert L.fq_sym_row_scales(x.data_ptr(), sc.data_ptr(), rows, cols, 8, _lib.DTYPE_BF16, 0, 0, -2.0, 2.0, b1.data_ptr(), m1.data_ptr(), mb, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(b0, b1)
    clippable = (b0[:, 0] >= 2.0)
    mrw = mb // rows
    assert torch.equal(m0.view(rows, mrw)[clippable], m1.view(rows, mrw)[clippable])     # mask rows are written only where needed
    _, _, s = ops.sym_quantize_debug(x, 8)
    assert torch.equal(sc[:, 0], s)


def test_export_error_paths(ops):
    from llm_qat_amd import _lib
    L = _lib.lib()
    x = torch.randn(4, 64, device="cuda").bfloat16()
    with pytest.raises(ValueError):
        ops.sym_export(x, 8, container="int3")
    with pytest.raises(RuntimeError):
        ops.sym_export(torch.randn(4, 64), 8)          # CPU tensor: no fallback
    st = torch.cuda.current_stream().cuda_stream
    assert L.fq_sym_export(x.data_ptr(), None, None, None, 4, 64, 8, _lib.BINS_INT8, _lib.DTYPE_BF16, 0, 0, st) != 0   # bins NULL
    assert L.fq_sym_export(x.data_ptr(), x.data_ptr(), None, None, 4, 64, 8, 9, _lib.DTYPE_BF16, 0, 0, st) != 0       # bad container
    assert L.fq_asym_export(x.data_ptr(), x.data_ptr(), None, None, 4, 64, 0, _lib.BINS_INT8, _lib.DTYPE_BF16, 0, st) != 0    # bits
    assert L.fq_sym_export(x.float().data_ptr(), x.data_ptr(), None, None, 4, 64, 8, _lib.BINS_INT8, _lib.DTYPE_F32, 0, 1, st) != 0   # autocast on fp32
    assert L.fq_export_bins_bytes(4, 63, _lib.BINS_INT4) == 4 * 32 and L.fq_export_bins_bytes(4, 64, _lib.BINS_INT16) == 4 * 128


def test_quantize_linear_export_weight(ops):
    from llm_qat_amd.utils_quant import QuantizeLinear
    lin = QuantizeLinear(1024, 256, w_bits=4, a_bits=8).cuda().bfloat16()
    e = lin.export_weight()
    assert e.container == "int4" and e.bins.shape == (256, 512) and e.scales.shape == (256, 2)
    wq = ops.sym_quantize(lin.weight.detach(), 4)
    d = e.dequantize()
    assert torch.equal(torch.where(d == 0, torch.zeros_like(d), d), torch.where(wq == 0, torch.zeros_like(wq), wq))
    with pytest.raises(ValueError):
        QuantizeLinear(64, 64, w_bits=2, a_bits=8).cuda().export_weight()
