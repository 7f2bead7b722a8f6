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


@pytest.mark.parametrize("style,shape,bits,container", [("weight", (4096, 11008), 4, "int4"), ("weight", (4096, 11008), 8, "int8"),
                                                        ("weight", (4096, 11008), 8, "int16"), ("act", (4096, 11008), 8, "int8"),
                                                        ("act", (2048, 4096), 4, "int4"), ("act", (2048, 11008), 8, "int8")])
def test_export_at_metric_size_vs_oracle(ops, style, shape, bits, container):
    """the export kernels at BASELINE.json's sizes (the tensors bench.py times): packed bytes, scales and overflow counts of
    >= 40 sampled rows -- first, last, evenly spaced, and rows the container saturates -- bit-identical to the CPU oracle; plus
    size-independent properties over the whole tensor (every bin inside the container, overflow == #bins the reference's
    rounding puts outside it, dequantisation == the fake-quant forward where nothing overflowed)."""
    rows, cols = shape
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.randn(shape, generator=g, device="cuda")
    if style == "weight":
        x *= 0.02
    else:
        x[torch.rand(shape, generator=g, device="cuda") < 1e-3] *= 20.0
    x = x.bfloat16()
    e = ops.sym_export(x, bits, False, container=container, autocast=False)
    over = e.overflow.cpu().numpy()
    sat_rows = np.flatnonzero(over)[:8].tolist()
    sel = sorted(set(np.linspace(0, rows - 1, 40).astype(int).tolist() + sat_rows))
    ob, osc, oov = O.export("sym", np_from(x[sel]), len(sel), cols, bits, container, "bf16")
    raw = e.bins.contiguous().view(torch.uint8).reshape(rows, -1)[sel].cpu().numpy()
    assert (raw == ob).all(), f"{(raw != ob).sum()} packed bytes differ on the sampled rows"
    assert (over[sel] == oov).all() and bits_equal(e.scales[sel].cpu().numpy(), osc, "fp32")
    # whole tensor: the unclamped bins of the debug forward (oracle-checked elsewhere) vs the container
    y, idx, _ = ops.sym_quantize_debug(x, bits, False)
    lo, hi = _crange(container, True)
    assert torch.equal(e.unpacked(), idx.clamp(lo, hi))
    assert torch.equal(e.overflow.long(), ((idx < lo) | (idx > hi)).sum(dim=1))
    if container == "int8" and bits == 8:
        assert over.sum() > 0, "bf16 8-bit rows reach the bin +128: some rows must saturate an int8 container"
    clean = e.overflow == 0
    assert clean.any()
    deq = e.dequantize()
    assert torch.equal((deq[clean].float() + 0.0), (y[clean].float() + 0.0))       # (+ 0.0: an integer bin has no -0)


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


def test_randomized_stress_round2_entry_points(ops):
    """Randomized sweep over the round-2 entry points with adversarial content (NaN / Inf / zero rows, row magnitudes at the
    ends of each dtype's range, odd widths, misaligned storage): packed export vs the oracle (bytes, scales, overflow
    counts), multi-tensor launches and the in-place backward vs the single-tensor reference path.  LLMQAT_STRESS_TRIALS
    scales it (default 90; 20 000 trials verified on the final kernels)."""
    import os
    rng = np.random.default_rng(20261005)
    col_choices = [1, 2, 3, 8, 24, 64, 100, 256, 264, 512, 1000, 1024, 2048, 4096, 4104, 8192, 11008, 16384, 20000]
    for trial in range(int(os.environ.get("LLMQAT_STRESS_TRIALS", "90"))):
        dtype = ["bf16", "fp32", "fp16"][trial % 3]
        kind = "sym" if rng.random() < 0.6 else "asym"
        rows, cols = int(rng.integers(1, 40)), int(rng.choice(col_choices))
        bits = int(rng.choice([2, 3, 4, 6, 8, 12, 16])) if kind == "sym" else int(rng.choice([1, 2, 4, 8, 16]))
        container = str(rng.choice(["int4", "int8", "int16"]))
        scales = [1e-4, 0.02, 1.0, 3.0, 100.0]
        if rng.random() < 0.25:
            scales = {"fp32": [1e-42, 1e-38, 1e-30, 1e-19, 1e-10, 1e10, 1e19, 1e30, 1e38], "bf16": [1e-38, 1e-30, 1e-19, 1e-10, 1e10, 1e19, 1e30, 1e38],
                      "fp16": [1e-7, 1e-6, 1e-5, 1e-3, 30.0, 1e3, 2e4, 6e4]}[dtype]
        with np.errstate(over="ignore"):
            x = rng.standard_normal((rows, cols)).astype(np.float32) * rng.choice(scales, size=(rows, 1)).astype(np.float32)
        if rng.random() < 0.3:
            x[rng.integers(0, rows), rng.integers(0, cols)] = rng.choice([np.nan, np.inf, -np.inf])
        if rng.random() < 0.3:
            x[rng.integers(0, rows)] = 0.0
        xt = torch.from_numpy(x).to(TD[dtype])
        x_np, xd = np_from(xt), xt.cuda()
        if rng.random() < 0.2:      # contiguous but only element-aligned storage
            flat = torch.empty(rows * cols + 1, dtype=TD[dtype], device="cuda")
            flat[1:].copy_(xd.reshape(-1))
            xd = flat[1:].view(rows, cols)
        tag = f"trial {trial}: {kind} {dtype} [{rows},{cols}] b{bits} {container}"
        e = ops.sym_export(xd, bits, container=container, autocast=False) if kind == "sym" else ops.asym_export(xd, bits, container=container)
        ob, osc, oov = O.export(kind, x_np, rows, cols, bits, container, dtype)
        raw = e.bins.contiguous().view(torch.uint8).reshape(rows, -1).cpu().numpy()
        assert (raw == ob).all(), tag + f": {(raw != ob).sum()} bytes differ"
        assert (e.overflow.cpu().numpy() == oov).all(), tag + " overflow"
        got_sc = e.scales.cpu().numpy()
        if kind == "asym":   # beta may differ in the sign of zero (-0.0 and +0.0 are equal minima)
            got_sc, osc = got_sc.copy(), osc.copy()
            got_sc[:, 1][got_sc[:, 1] == 0] = 0.0
            osc[:, 1][osc[:, 1] == 0] = 0.0
        assert bits_equal(got_sc, osc, "fp32"), tag + " scales"
        if kind == "sym" and dtype != "fp32":
            ea = ops.sym_export(xd, bits, container=container, autocast=True)
            oba, osca, oova = O.export("sym", x_np, rows, cols, bits, container, dtype, autocast=True)
            assert (ea.bins.contiguous().view(torch.uint8).reshape(rows, -1).cpu().numpy() == oba).all(), tag + " autocast bins"
            assert (ea.overflow.cpu().numpy() == oova).all() and bits_equal(ea.scales.cpu().numpy(), osca, "fp32"), tag + " autocast scales"
        # multi-tensor launch + in-place backward against the single-tensor path (Sym, aligned, register-resident shapes only)
        if kind == "sym" and xd.data_ptr() % 16 == 0:
            nt = int(rng.integers(2, 5))
            ts = [xd] + [(torch.randn(int(rng.integers(1, 20)), cols, device="cuda") * float(rng.choice([0.02, 1.0, 5.0]))).to(TD[dtype]) for _ in range(nt - 1)]
            bs = [max(bits, 2)] + [int(rng.choice([3, 4, 8])) for _ in range(nt - 1)]
            res = ops.multi_forward(ts, bs, [True] * nt, -2.0, 2.0)
            if res is not None:
                ys, sides, rws, c = res
                gs = [torch.randn_like(t) for t in ts]
                inpl = [bool(rng.random() < 0.5) for _ in ts]
                outs = ops.multi_backward([g.clone() for g in gs], sides, rws, c, -2.0, 2.0, inplace=inpl)
                for t, b, y, g, o in zip(ts, bs, ys, gs, outs):
                    y1, s1, r1, c1 = ops.train_forward("sym", t, b, False, -2.0, 2.0)
                    assert torch.equal(y.view(torch.uint8), y1.view(torch.uint8)), tag + " multi fwd"
                    assert torch.equal(o.view(torch.uint8), ops.train_backward(g.clone(), s1, r1, c1, -2.0, 2.0).view(torch.uint8)), tag + " multi bwd"
