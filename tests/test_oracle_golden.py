"""CPU tier: the oracle (oracle/fq_oracle.c) against the golden vectors the real reference produced.

This is what pins the oracle: every forward / backward fixture must match bit for bit.
"""
import numpy as np
import pytest

from conftest import bits_equal, golden, mismatch_report, to_f32
from oracle import oracle as O


@pytest.mark.parametrize("kind", ["sym", "asym"])
def test_forward_matches_reference_fixtures(kind):
    G = golden(f"{kind}_fwd.npz")
    assert len(G.cases) >= 100
    for c in G.cases:
        x, dt = G.arr(c, "x"), c["dtype"]
        rows, cols = O.rows_cols(c["shape"], c["layerwise"])
        if kind == "sym":
            y, idx, scale = O.sym_fwd(x, rows, cols, c["bits"], dt)
            want_scale = to_f32(G.arr(c, "scale"), dt)
            assert bits_equal(scale, want_scale, "fp32"), f"{c['name']}: scale"
        else:
            y, idx, alpha, beta = O.asym_fwd(x, rows, cols, c["bits"], dt)
            assert bits_equal(alpha, to_f32(G.arr(c, "alpha"), dt), "fp32"), f"{c['name']}: alpha"
            # beta may legitimately differ in the sign of zero (-0.0 vs +0.0 are equal minima)
            wb = to_f32(G.arr(c, "beta"), dt)
            assert bits_equal(np.where(beta == 0, 0.0, beta).astype(np.float32), np.where(wb == 0, 0.0, wb).astype(np.float32), "fp32"), f"{c['name']}: beta"
        assert (idx == G.arr(c, "idx")).all(), f"{c['name']}: bin indices differ"
        assert bits_equal(y, G.arr(c, "y"), dt), f"{c['name']}: {mismatch_report(y, G.arr(c, 'y'), dt)}"


def test_ste_backward_matches_reference_fixtures():
    G = golden("ste_bwd.npz")
    assert len(G.cases) == 15
    for c in G.cases:
        clip = G.arr(c, "clip")
        gx = O.ste_bwd(G.arr(c, "g"), G.arr(c, "x"), float(clip[0]), float(clip[1]), c["dtype"])
        want = G.arr(c, "gx")
        a, b = (gx.view(np.uint32), want.view(np.uint32)) if gx.dtype == np.float32 else (gx, want)
        assert (a == b).all(), c["name"]  # a clone: even NaN payloads of g survive


def test_no_clamp_8bit_bf16_reaches_128():
    """the Sym forward has no clamp; in bf16 the 8-bit index reaches +-128."""
    G = golden("sym_fwd.npz")
    seen = 0
    for c in G.cases:
        if c["dtype"] == "bf16" and c["bits"] == 8 and not c.get("row_names"):
            idx = G.arr(c, "idx")
            finite = idx[np.abs(idx) < 10**6]
            seen = max(seen, int(np.abs(finite).max()))
    assert seen == 128


def test_fixture_edge_rows():
    """Edge behaviour pinned by the reference: zero rows, NaN rows, +Inf rows."""
    G = golden("sym_fwd.npz")
    c = next(c for c in G.cases if c["name"] == "sym_fp32_b4_adversarial")
    names = c["row_names"]
    y = G.arr(c, "y")
    assert (y[names.index("all_zero")] == 0).all()
    assert np.isnan(y[names.index("nan")]).all()                      # NaN anywhere -> whole row NaN
    r = y[names.index("pos_inf")]
    assert np.isnan(r[0]) and (r[1:] == 0).all()                      # NaN at the Inf, 0 elsewhere
    A = golden("asym_fwd.npz")
    c = next(c for c in A.cases if c["name"] == "asym_fp32_b4_adversarial")
    assert np.isnan(A.arr(c, "y")[names.index("pos_inf")]).all()      # Asym: whole row NaN


def test_oracle_rejects_bad_arguments():
    x = np.zeros(4, np.float32)
    with pytest.raises(ValueError):
        O.sym_fwd(x, 1, 4, 40, "fp32")
    with pytest.raises(TypeError):
        O.sym_fwd(x, 1, 4, 4, "bf16")
    with pytest.raises(ValueError):
        O.rows_cols((1, 2, 3, 4, 5), False)


def test_device_semantics_only_differ_where_documented():
    """sem=1 (device eager) == sem=0 for fp32 Sym always, and for bf16 Sym unless the row max is tiny."""
    rng = np.random.default_rng(5)
    x32 = (rng.standard_normal((64, 96)) * rng.choice([1e-7, 1e-5, 1e-3, 1.0], size=(64, 1))).astype(np.float32)
    y0, i0, _ = O.sym_fwd(x32, 64, 96, 8, "fp32", sem=O.SEM_CPU)
    y1, i1, _ = O.sym_fwd(x32, 64, 96, 8, "fp32", sem=O.SEM_DEVICE)
    assert (i0 == i1).all() and bits_equal(y0, y1, "fp32")
    xb = (x32.view(np.uint32) >> 16).astype(np.uint16)
    y0, i0, _ = O.sym_fwd(xb, 64, 96, 8, "bf16", sem=O.SEM_CPU)
    y1, i1, _ = O.sym_fwd(xb, 64, 96, 8, "bf16", sem=O.SEM_DEVICE)
    big = np.abs(to_f32(xb, "bf16")).max(axis=1) > 1e-3
    assert (i0[big] == i1[big]).all() and bits_equal(y0[big], y1[big], "bf16")


def test_bf16_reciprocal_multiply_equals_divide_exhaustively():
    """The bf16 <=8-bit Sym kernel finishes with  y = rb(idx * (1/t2))  instead of  rb(idx / t2).
    Exhaustive proof over every positive finite bf16 divisor and every |idx| <= 256 that the
    two are bit-identical after the bf16 rounding (DESIGN.md "Numerics")."""
    bits = np.arange(0x0080, 0x7F80, dtype=np.uint32)            # all positive normal bf16 values
    t2 = (bits << 16).view(np.float32)
    t2 = t2[(t2 > 1e-30) & (t2 < 1e30)]                          # t2 = s + 1e-6 lives in [1e-6, 1.3e8]
    idx = np.arange(-256, 257, dtype=np.float32)

    def rb(v):  # fp32 -> bf16 RNE (finite inputs)
        u = v.view(np.uint32)
        return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)

    rinv = (np.float32(1.0) / t2).astype(np.float32)
    for chunk in np.array_split(np.arange(t2.size), 64):
        a = (idx[None, :] / t2[chunk, None]).astype(np.float32)
        b = (idx[None, :] * rinv[chunk, None]).astype(np.float32)
        assert (rb(a) == rb(b)).all()
    # wider quantizers (A16 / KV16): the bin index is rint() of a bf16 value, i.e. m * 2^j with m <= 255 -- the
    # identity is scale invariant, checked here directly up to 2^15
    big = np.unique(np.concatenate([np.arange(128, 256, dtype=np.float32) * np.float32(2.0 ** j) for j in range(1, 9)]))
    big = np.concatenate([big, -big])
    sel = t2[(t2 > 1e-6) & (t2 < 1e9)]
    rsel = (np.float32(1.0) / sel).astype(np.float32)
    for chunk in np.array_split(np.arange(sel.size), 16):
        a = (big[None, :] / sel[chunk, None]).astype(np.float32)
        b = (big[None, :] * rsel[chunk, None]).astype(np.float32)
        assert (rb(a) == rb(b)).all()


def test_w12_weight_branches_match_reference_fixtures():
    """QuantizeLinear's 1-/2-bit weight branches: the value handed to F.linear (detach trick included),
    given the reference's own mean-|w| scale, is reproduced bit for bit."""
    G = golden("w12.npz")
    assert len(G.cases) == 36
    for c in G.cases:
        dt, (rows, cols) = c["dtype"], c["shape"]
        w, sc = G.arr(c, "w"), to_f32(G.arr(c, "scale"), c["dtype"])
        if c["layerwise"]:
            q, _ = O.w12_fwd(w.reshape(1, -1), 1, rows * cols, c["w_bits"], dt, scale_in=sc)
            q = q.reshape(rows, cols)
        else:
            q, _ = O.w12_fwd(w, rows, cols, c["w_bits"], dt, scale_in=sc)
        assert bits_equal(q, G.arr(c, "wq"), dt), f"{c['name']}: {mismatch_report(q, G.arr(c, 'wq'), dt)}"
        # the oracle's own (double-precision) mean agrees with ATen's fp32 mean to within one ulp of the dtype
        _, own = O.w12_fwd(w.reshape(1, -1) if c["layerwise"] else w, 1 if c["layerwise"] else rows, rows * cols if c["layerwise"] else cols, c["w_bits"], dt)
        ok = np.isfinite(sc) & (sc != 0)
        tol = {"fp32": 1e-6, "bf16": 8e-3, "fp16": 1e-3}[dt]
        assert np.allclose(own[ok], sc[ok], rtol=tol), c["name"]


def test_c_oracle_agrees_with_an_independent_numpy_restatement():
    """Second, independently written restatement of the Sym recipe (vectorised numpy, fp32 ops + explicit
    bf16 rounding) against the C oracle on random rows -- guards the oracle against a shared typo."""
    rng = np.random.default_rng(123)

    def rb(v):
        u = np.ascontiguousarray(v, np.float32).view(np.uint32)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
        return r.view(np.float32)

    for bits in (3, 4, 8):
        x32 = (rng.standard_normal((200, 173)) * rng.choice([1e-6, 1e-3, 0.02, 1.0, 40.0], size=(200, 1))).astype(np.float32)
        xb = (rb(x32).view(np.uint32) >> 16).astype(np.uint16)
        x = (xb.astype(np.uint32) << 16).view(np.float32)
        qmax = np.float32(2 ** (bits - 1) - 1)
        c6 = rb(np.float32(1e-6))
        m = np.abs(x).max(axis=1, keepdims=True)
        s = rb(rb(np.float32(1.0) / rb(m + c6)) * qmax)
        idx = np.rint(rb(x * s))
        y = rb(idx / rb(s + c6))
        yo, io, so = O.sym_fwd(xb, 200, 173, bits, "bf16")
        assert (io == idx.astype(np.int32)).all()
        assert ((y.view(np.uint32) >> 16).astype(np.uint16) == yo).all()
        assert (so == s[:, 0]).all()


def test_bf16_asym_reciprocal_multiplies_equal_divides():
    """The bf16 <=8-bit Asym kernel replaces both IEEE divides by reciprocal multiplies:
         n = rb(d * (1/a))  for  rb(d / a)      d = rb(x - beta), a = rb(alpha + 1e-8): 8-bit significands
         w = rb(q * (1/S))  for  rb(q / S)      q in [0, 255], S = 2^bits - 1
    Exhaustive over significand pairs (both operations are exactly scale-invariant in the normal range) and over
    every (q, S)."""
    def rb(v):
        u = np.ascontiguousarray(v, np.float32).view(np.uint32)
        return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)

    def bf16_range(lo_exp, hi_exp):          # every positive bf16 value in [2^lo_exp, 2^hi_exp)
        bits = np.arange((127 + lo_exp) << 7, (127 + hi_exp) << 7, dtype=np.uint32)
        return (bits << 16).view(np.float32)

    d = bf16_range(-24, 1)                   # d <= ~a, down to 2^-24 * a
    for a in (bf16_range(0, 1), bf16_range(-27, -26), bf16_range(100, 101)):
        ra = (np.float32(1.0) / a).astype(np.float32)
        dd = d * np.float32(a[0])            # same relative range at this exponent
        exact = (dd[:, None] / a[None, :]).astype(np.float32)
        fast = (dd[:, None] * ra[None, :]).astype(np.float32)
        assert (rb(exact) == rb(fast)).all()
    q = np.arange(0, 257, dtype=np.float32)
    for bits in range(1, 9):
        S = np.float32(2 ** bits - 1)
        assert (rb(q / S) == rb(q * (np.float32(1.0) / S))).all()


def test_oracle_is_clean_under_asan_ubsan():
    """GPU sanitizers are unavailable on the pool; the CPU oracle (what every kernel is judged against) is run over
    ragged shapes and hostile values under AddressSanitizer + UBSan (oracle/selftest.c, `make -C oracle sanitize`)."""
    import os
    import shutil
    import subprocess
    from conftest import ROOT
    if shutil.which("gcc") is None and shutil.which("cc") is None:
        pytest.skip("no C compiler")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "sanitize"], capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and ("cannot find -lasan" in r.stderr or "libasan" in r.stderr and "No such file" in r.stderr):
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "oracle selftest ok" in r.stdout


# ---- packed integer export: the oracle's container logic against the reference's own bins ----------------
def _container_range(container, signed):
    cb = {"int4": 4, "int8": 8, "int16": 16}[container]
    return (-(1 << (cb - 1)), (1 << (cb - 1)) - 1) if signed else (0, (1 << cb) - 1)


@pytest.mark.parametrize("kind", ["sym", "asym"])
def test_export_bins_are_the_reference_bins(kind):
    """Integer work, bit-exact bar: the exported bins equal the `idx` arrays the real reference produced
    (tests/golden/*_fwd.npz), saturated to the container; overflow counts exactly the elements that did not fit."""
    G = golden(f"{kind}_fwd.npz")
    checked = sat_seen = 0
    for c in G.cases:
        x, dt, bits = G.arr(c, "x"), c["dtype"], c["bits"]
        rows, cols = O.rows_cols(c["shape"], c["layerwise"])
        idx = G.arr(c, "idx").reshape(rows, cols).astype(np.int64)
        nan = idx == np.iinfo(np.int32).min
        for container in ("int4", "int8", "int16"):
            lo, hi = _container_range(container, kind == "sym")
            bins, scales, over = O.export(kind, x, rows, cols, bits, container, dt)
            got = O.unpack_bins(bins, cols, container, kind == "sym")
            want = np.where(nan, 0, np.clip(idx, lo, hi))
            assert (got == want).all(), f"{c['name']} {container}: bins differ"
            bad = nan | (idx < lo) | (idx > hi)
            assert (over == bad.sum(axis=1)).all(), f"{c['name']} {container}: overflow count"
            sat_seen += int(bad.any())
            checked += 1
        if kind == "sym":
            assert bits_equal(scales[:, 0], to_f32(G.arr(c, "scale"), dt).reshape(-1), "fp32"), f"{c['name']}: s"
        else:
            assert bits_equal(scales[:, 1].copy(), np.asarray(O.asym_fwd(x, rows, cols, bits, dt)[3], np.float32), "fp32")
    assert checked >= 300 and sat_seen > 0


def _pos_zero(a, dt):
    a = np.array(a, copy=True)
    if dt == "fp32":
        a[a == 0] = 0.0
    else:
        a[(a & 0x7FFF) == 0] = 0
    return a


def test_export_dequant_reproduces_the_forward():
    """overflow == 0  <=>  bins / t2 (Sym) reproduces the reference's forward output bit for bit"""
    G = golden("sym_fwd.npz")
    n = 0
    for c in G.cases:
        x, dt, bits = G.arr(c, "x"), c["dtype"], c["bits"]
        rows, cols = O.rows_cols(c["shape"], c["layerwise"])
        bins, scales, over = O.export("sym", x, rows, cols, bits, "int16", dt)
        q = O.unpack_bins(bins, cols, "int16", True).astype(np.float32)
        y32 = (q / scales[:, 1:2]).astype(np.float32)
        want = G.arr(c, "y").reshape(rows, cols)
        ok_rows = over == 0
        if dt == "fp32":
            got = y32
        else:
            import torch
            t = torch.from_numpy(y32).to(torch.bfloat16 if dt == "bf16" else torch.float16)
            got = t.view(torch.int16).numpy().view(np.uint16)
        # an integer has no -0: the reference's round(-0.3) = -0.0 dequantises to -0.0, the exported bin 0 to +0.0
        assert bits_equal(_pos_zero(got[ok_rows], dt), _pos_zero(want[ok_rows], dt), dt), c["name"]
        n += int(ok_rows.sum())
    assert n > 400


def test_export_8bit_bf16_plus_128_is_counted_not_hidden():
    """the reference has no clamp: an 8-bit bf16 row whose top bin is +128 saturates int8 and says so; -128 fits"""
    import torch
    x = torch.tensor([[1.0, -1.0, 0.5, 0.25] * 4], dtype=torch.float32)
    x[0, 0] = 1.0
    xb = x.to(torch.bfloat16)
    xb_np = xb.view(torch.int16).numpy().view(np.uint16)
    y, idx, s = O.sym_fwd(xb_np, 1, 16, 8, "bf16")
    bins, scales, over = O.export("sym", xb_np, 1, 16, 8, "int8", "bf16")
    got = O.unpack_bins(bins, 16, "int8", True)
    assert (got == np.clip(idx.reshape(1, 16), -128, 127)).all()
    assert over[0] == int((idx > 127).sum())
    b16, _, o16 = O.export("sym", xb_np, 1, 16, 8, "int16", "bf16")
    assert o16[0] == 0 and (O.unpack_bins(b16, 16, "int16", True) == idx.reshape(1, 16)).all()


def test_module_level_operands_match_the_oracle():
    """quantize_linear.npz (round 4: + `opx` / `opw`, the tensors the real reference's QuantizeLinear hands to F.linear,
    models/utils_quant.py:195-250): the oracle reproduces them bit for bit -- W >= 3 weights and Sym / Asym activations, per row and
    layerwise, fp32 and bf16 -- so the module's OPERANDS are pinned at module level, not only its GEMM outputs (which depend on the
    accumulation order and are compared with a tolerance)."""
    G = golden("quantize_linear.npz")
    checked = 0
    for c in G.cases:
        dt = c["dtype"]
        w, x = G.arr(c, "w"), G.arr(c, "x")
        if 3 <= c["w_bits"] < 32:
            rows, cols = O.rows_cols(w.shape, c.get("weight_layerwise", False))
            y, _, _ = O.sym_fwd(w, rows, cols, c["w_bits"], dt, want_idx=False)
            assert bits_equal(y.reshape(w.shape), G.arr(c, "opw"), dt), f"{c['name']}: weight operand"
            checked += 1
        elif c["w_bits"] >= 32:
            assert bits_equal(w, G.arr(c, "opw"), dt), c["name"]
        if 2 < c["a_bits"] < 32:
            rows, cols = O.rows_cols(x.shape, c.get("act_layerwise", False))
            fn = O.sym_fwd if c["symmetric"] else O.asym_fwd
            y = fn(x, rows, cols, c["a_bits"], dt, want_idx=False)[0]
            assert bits_equal(y.reshape(x.shape), G.arr(c, "opx"), dt), f"{c['name']}: input operand"
            checked += 1
        else:
            assert bits_equal(x, G.arr(c, "opx"), dt), c["name"]   # a_bits <= 2 or >= 32: the input goes in as it is (:184, :244)
    assert checked >= 30
