"""CPU tier: property tests of the oracle (hypothesis) -- size- and value-independent facts of the reference's arithmetic
(models/utils_quant.py:50-72, :83-87, :110-147) that hold for EVERY input, beyond the fixed golden vectors:
rows are independent, element order within a row is irrelevant, SymQuantizer is odd, bins are monotone in x and stay in
range, the STE gradient is exactly `g where lo < x < hi or x is NaN`.  (The GPU tier checks the same properties on the kernels at
BASELINE's full sizes: tests/test_gpu_parity.py::test_full_size_properties.)"""
import numpy as np
from hypothesis import given, settings, strategies as st

from conftest import bits_equal, to_f32
from oracle import oracle as O

DT = ("fp32", "bf16", "fp16")


def make(rng, rows, cols, dtype, scale):
    import torch
    x = (rng.standard_normal((rows, cols)) * scale).astype(np.float32)
    if dtype == "fp32":
        return x
    t = torch.from_numpy(x).to(torch.bfloat16 if dtype == "bf16" else torch.float16)
    return t.view(torch.int16).numpy().view(np.uint16).copy()


def neg(a, dtype):
    return -a if dtype == "fp32" else (a ^ np.uint16(0x8000))


cases = st.tuples(st.integers(1, 6), st.integers(1, 70), st.sampled_from(DT), st.sampled_from([3, 4, 8, 16]),
                  st.sampled_from([1e-4, 0.02, 1.0, 7.0, 300.0]), st.integers(0, 2**31 - 1))


@settings(max_examples=150, deadline=None, derandomize=True)
@given(cases)
def test_sym_forward_properties(c):
    rows, cols, dtype, bits, scale, seed = c
    rng = np.random.default_rng(seed)
    x = make(rng, rows, cols, dtype, scale)
    y, idx, s = O.sym_fwd(x, rows, cols, bits, dtype)
    qmax = 2 ** (bits - 1) - 1
    # rows are independent and their order is irrelevant
    perm = rng.permutation(rows)
    y2, idx2, _ = O.sym_fwd(np.ascontiguousarray(x[perm]), rows, cols, bits, dtype)
    assert bits_equal(y2, y[perm], dtype) and (idx2 == idx[perm]).all()
    # within a row the element order does not matter (the scale is a max)
    cperm = rng.permutation(cols)
    y3, idx3, _ = O.sym_fwd(np.ascontiguousarray(x[:, cperm]), rows, cols, bits, dtype)
    assert bits_equal(y3, y[:, cperm], dtype) and (idx3 == idx[:, cperm]).all()
    # SymQuantizer is odd: fq(-x) == -fq(x), bit for bit (round-half-even is symmetric; signed zeros included)
    yn, idxn, _ = O.sym_fwd(neg(x, dtype), rows, cols, bits, dtype)
    assert bits_equal(yn, neg(y, dtype), dtype) and (idxn == -idx).all()
    # bins: integers, monotone in x, within the range the reference can reach (no clamp: the row maximum may land one bin above qmax
    # when s is rounded up in a 16-bit dtype).  Rows whose scale overflows the dtype (fp16: 1 / max beyond 65504 -- the reference
    # really produces inf / NaN bins there, fixture row "fp16_overflowing_scale") are excluded from the range / order statements.
    xf = to_f32(x, dtype)
    ok = np.isfinite(s) & (s > 0)
    if ok.any():
        order = np.argsort(xf[ok], axis=1, kind="stable")
        assert (np.diff(np.take_along_axis(idx[ok], order, axis=1), axis=1) >= 0).all()
        assert np.abs(idx[ok]).max() <= qmax + max(1, qmax // 128)
        # a row whose maximum is not tiny reaches the top of the range
        big = np.abs(xf[ok]).max(axis=1) > 1e-3
        assert (np.abs(idx[ok])[big].max(axis=1) >= qmax - max(1, qmax // 128)).all() if big.any() else True


@settings(max_examples=150, deadline=None, derandomize=True)
@given(cases, st.sampled_from([(-2.0, 2.0), (-0.5, 0.75), (-0.3009, 0.3009), (0.0, 0.0)]))
def test_ste_backward_is_exactly_the_predicate(c, clip):
    rows, cols, dtype, _, scale, seed = c
    rng = np.random.default_rng(seed)
    x = make(rng, rows, cols, dtype, max(scale, 0.3) if scale < 300 else 2.0)
    g = make(rng, rows, cols, dtype, 1.0)
    lo, hi = clip
    gx = O.ste_bwd(g, x, lo, hi, dtype)
    xf = to_f32(x, dtype)
    # the comparison runs in the tensor dtype: the clip values are rounded to it first (:85-86 compare a dtype tensor with a 0-dim tensor)
    lo_d, hi_d = (to_f32(make_scalar(v, dtype), dtype) for v in (lo, hi))
    keep = ((xf < hi_d) & (xf > lo_d)) | np.isnan(xf)
    want = np.where(keep, g, np.zeros_like(g))
    assert (gx.view(np.uint32) == want.view(np.uint32)).all() if dtype == "fp32" else (gx == want).all()
    # linear in g for a power of two: ste(2 g) == 2 ste(g)
    g2 = make_scaled(g, dtype)
    gx2 = O.ste_bwd(g2, x, lo, hi, dtype)
    assert bits_equal(gx2, make_scaled(gx, dtype), dtype)


def make_scalar(v, dtype):
    import torch
    if dtype == "fp32":
        return np.float32(v)
    t = torch.tensor([v], dtype=torch.float32).to(torch.bfloat16 if dtype == "bf16" else torch.float16)
    return t.view(torch.int16).numpy().view(np.uint16)[0]


def make_scaled(a, dtype):
    """2 * a in the dtype (exact unless it overflows; inputs here are O(1))"""
    import torch
    if dtype == "fp32":
        return (a * np.float32(2.0)).astype(np.float32)
    td = torch.bfloat16 if dtype == "bf16" else torch.float16
    t = torch.from_numpy(a.view(np.int16).copy()).view(td)
    return (t.float() * 2.0).to(td).view(torch.int16).numpy().view(np.uint16).copy()


@settings(max_examples=100, deadline=None, derandomize=True)
@given(cases)
def test_asym_forward_properties(c):
    rows, cols, dtype, bits, scale, seed = c
    rng = np.random.default_rng(seed)
    x = make(rng, rows, cols, dtype, scale)
    y, idx, alpha, beta = O.asym_fwd(x, rows, cols, bits, dtype)
    perm = rng.permutation(rows)
    y2, idx2, _, _ = O.asym_fwd(np.ascontiguousarray(x[perm]), rows, cols, bits, dtype)
    assert bits_equal(y2, y[perm], dtype) and (idx2 == idx[perm]).all()
    cperm = rng.permutation(cols)
    y3, idx3, _, _ = O.asym_fwd(np.ascontiguousarray(x[:, cperm]), rows, cols, bits, dtype)
    assert bits_equal(y3, y[:, cperm], dtype) and (idx3 == idx[:, cperm]).all()
    S = 2 ** bits - 1
    xf = to_f32(x, dtype)
    ok = np.isfinite(alpha) & (alpha > 1e-6 if dtype != "fp32" else alpha > 0)   # (a constant row divides 0 by alpha + 1e-8: 0 / 0 when 1e-8 rounds to 0 in fp16)
    if dtype == "fp16" and bits == 16:
        ok[:] = False   # 1.0 * 65535 overflows fp16: the reference's top bin is inf there
    if ok.any():
        assert idx[ok].min() >= 0 and idx[ok].max() <= S + max(1, S // 128)       # unsigned bins (the top one may round up in a 16-bit dtype)
        order = np.argsort(xf[ok], axis=1, kind="stable")
        assert (np.diff(np.take_along_axis(idx[ok], order, axis=1), axis=1) >= 0).all()
        # the row minimum maps to bin 0
        amin = xf[ok].argmin(axis=1)
        assert (idx[ok][np.arange(int(ok.sum())), amin] == 0).all()
