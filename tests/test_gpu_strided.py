"""GPU tier: tensors whose rows do not follow one another in memory (VERDICT r04 "missing" #4).  The reference accepts any strides
(models/utils_quant.py:37: ATen handles them) and its elementwise results keep the input's layout.  Since round 5 the common "last
dimension contiguous, rows strided" layouts -- slices and chunk() of the last dimension, every other row, transpose(0, 1) of a 3-D tensor,
expanded rows -- are served IN the kernels (fq_rows_view, ABI 5): no .contiguous() in front of the launch, no copy_ behind it.

Every view x {Sym, Asym} x {mask, bounds, plain backward} x {no autocast, autocast}: values, gradients AND strides bit-identical to the
live eager chain on the same view; the views that the kernels serve are counted (ops._views_served) and the training forward / mask backward
stay engaged; views the kernels do not serve (a strided last dimension, 4-D, misaligned rows in the backward) still give the same results
through the copy path.  tools/strided_trace.py + profiles/r05_strided_no_copy_kernels.txt: the kernel trace of the served views holds no
copy kernel."""
import pytest
import torch

from oracle import eager_chain as E

pytestmark = pytest.mark.gpu


def base(shape, dtype, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    t = torch.randn(shape, generator=g, device="cuda") * 1.3
    t.view(-1)[::37] *= 3.0          # some values beyond the STE clip
    return t.to(dtype)


# name -> (builder, served by the kernels in the forward?)
VIEWS = {
    "cols_slice_2d": (lambda dt: base((48, 192), dt)[:, :128], True),
    "cols_offset_slice_2d": (lambda dt: base((48, 192), dt)[:, 64:], True),
    "every_other_row_2d": (lambda dt: base((64, 128), dt)[::2], True),
    "chunk_last_dim_3d": (lambda dt: base((3, 16, 384), dt).chunk(3, dim=-1)[1], True),
    "transpose01_3d": (lambda dt: base((16, 5, 256), dt).transpose(0, 1), True),
    "every_other_token_3d": (lambda dt: base((2, 32, 128), dt)[:, ::2, :], True),
    "narrow_batch_and_cols_3d": (lambda dt: base((4, 12, 320), dt)[1:3, :, 64:320], True),
    "expanded_rows_2d": (lambda dt: base((1, 256), dt).expand(24, 256), True),
    "model_width_slice": (lambda dt: base((8, 3 * 4096), dt)[:, 4096:8192], True),
    "misaligned_slice_2d": (lambda dt: base((40, 200), dt)[:, 1:129], True),      # rows start 2 / 4 bytes off a 16-byte boundary: element-wise kernel
    "odd_width_slice_2d": (lambda dt: base((40, 200), dt)[:, :63], True),
    "strided_last_dim": (lambda dt: base((24, 128), dt)[:, ::2], False),          # copy path
    "transpose12_3d": (lambda dt: base((3, 64, 24), dt).transpose(1, 2), False),  # copy path
    "transpose01_4d": (lambda dt: base((3, 2, 8, 96), dt).transpose(0, 1), False),  # `view(d0, d1, -1)` exists: copy path
    "slice_4d": (lambda dt: base((2, 3, 8, 96), dt)[..., :64], None),             # no such view: the reference raises, and so does the drop-in
}


def same(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and torch.equal(a.nan_to_num(nan=3.0), b.nan_to_num(nan=3.0))


@pytest.mark.parametrize("mode", ["mask", "bounds", "plain"])
@pytest.mark.parametrize("autocast", [False, True])
@pytest.mark.parametrize("kind", ["sym", "asym"])
@pytest.mark.parametrize("name", list(VIEWS))
def test_strided_views_match_the_eager_chain(name, kind, autocast, mode):
    import llm_qat_amd
    from llm_qat_amd import ops
    from llm_qat_amd.utils_quant import AsymQuantizer, SymQuantizer
    build, served = VIEWS[name]
    clip = torch.tensor([-2.0, 2.0])
    llm_qat_amd.set_semantics("device_eager")
    llm_qat_amd.set_backward_mode(mode)
    try:
        for dt in (torch.bfloat16, torch.float32):
            for bits in (4, 8):
                xe = build(dt).detach().requires_grad_(True)
                xq = build(dt).detach().requires_grad_(True)
                assert not xe.is_contiguous() and xe.stride() == xq.stride()
                if served is None:   # a layout the reference itself refuses (its 4-D branch needs `input.view(d0, d1, -1)`): same exception type
                    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                        with pytest.raises(RuntimeError):
                            (E.EagerSym if kind == "sym" else E.EagerAsym).apply(xe, clip, bits, False)
                        with pytest.raises(RuntimeError):
                            (SymQuantizer if kind == "sym" else AsymQuantizer).apply(xq, clip, bits, False)
                    continue
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                    ye = (E.EagerSym if kind == "sym" else E.EagerAsym).apply(xe, clip, bits, False)
                    before = ops._views_served
                    yq = (SymQuantizer if kind == "sym" else AsymQuantizer).apply(xq, clip, bits, False)
                tag = f"{name} {kind} {dt} b{bits} ac={autocast} {mode}"
                assert same(ye, yq), tag + ": values"
                assert ye.stride() == yq.stride(), f"{tag}: result strides {yq.stride()} vs the reference's {ye.stride()}"
                # (under autocast the Sym forward of rows that are not whole aligned 16-byte vectors has no register kernel to go to: copy path)
                if served and not (autocast and kind == "sym" and dt != torch.float32 and name in ("misaligned_slice_2d", "odd_width_slice_2d")):
                    assert ops._views_served > before, tag + ": took a copy instead of the view"
                # gradients: one with the result's own layout, one contiguous, one transposed where the shape allows
                gs = [torch.randn_like(ye), torch.randn(ye.shape, device="cuda", dtype=ye.dtype)]
                if ye.dim() == 3:
                    gs.append(torch.randn(ye.shape[1], ye.shape[0], ye.shape[2], device="cuda", dtype=ye.dtype).transpose(0, 1))
                for i, g in enumerate(gs):
                    xe.grad = xq.grad = None
                    ye.backward(g, retain_graph=True)
                    yq.backward(g.clone() if g.is_contiguous() else g, retain_graph=True)
                    assert same(xe.grad, xq.grad), f"{tag}: gradient {i}"
                    assert xe.grad.stride() == xq.grad.stride(), f"{tag}: gradient {i} strides"
    finally:
        llm_qat_amd.set_backward_mode("mask")
        llm_qat_amd.set_semantics("cpu_eager")


@pytest.mark.parametrize("autocast", [False, True])
def test_quantize_linear_on_strided_operands(autocast):
    """a QuantizeLinear whose input is a slice of a fused projection's output (rows strided): the pair launch and the mask backward stay
    engaged (stats), no copy is taken, results == the eager chain"""
    import sys
    import os
    import llm_qat_amd
    from llm_qat_amd import ops
    import llm_qat_amd.utils_quant as UQ
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import tiny_llama as TL
    llm_qat_amd.set_semantics("device_eager")
    try:
        fused = base((2, 24, 3 * 256), torch.bfloat16, seed=5)

        def run(Q):
            m = Q.QuantizeLinear(256, 192, w_bits=4, a_bits=8).cuda().bfloat16()
            with torch.no_grad():
                m.weight.copy_(base((192, 256), torch.bfloat16, seed=6) * 0.3)
            f = fused.clone().requires_grad_(True)
            x = f.chunk(3, dim=-1)[1]
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                out = m(x)
            out.float().square().mean().backward()
            return out.detach(), f.grad, m.weight.grad

        want = run(TL.EagerQuant())
        llm_qat_amd.reset_learned_state()
        llm_qat_amd.stats(reset=True)
        before = ops._views_served
        got = run(UQ)
        st = llm_qat_amd.stats()
        for a, b in zip(want, got):
            assert same(a, b)
        assert st.get("pair_launch") == 1 and ops._views_served > before, (st, ops._views_served - before)
    finally:
        llm_qat_amd.set_semantics("cpu_eager")
        llm_qat_amd.reset_learned_state()


def test_c_abi_views_against_the_contiguous_entry_points():
    """fq_rowwise_fwd_v / fq_ste_bwd_mask_multi_v / fq_ste_bwd_v on gathered rows == the contiguous entry points on a contiguous copy"""
    from llm_qat_amd import _lib, ops
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    big = base((6, 20, 512), torch.bfloat16, seed=9)
    x = big[1:5, ::2, 128:384]                      # [4, 10, 256], strides (10240, 1024, 1)
    xc = x.contiguous()
    rows, cols = 40, 256
    xv = _lib.RowsView(10, x.stride(0), x.stride(1))
    mb = L.fq_ste_mask_bytes(rows, cols, _lib.DTYPE_BF16)
    y = torch.empty(4, 10, 256, device="cuda", dtype=torch.bfloat16)
    side = torch.zeros(rows * 8 + mb, dtype=torch.uint8, device="cuda")
    _lib.check(L.fq_rowwise_fwd_v(0, x.data_ptr(), xv, y.data_ptr(), None, rows, cols, 8, _lib.DTYPE_BF16, 0, -2.0, 2.0, side.data_ptr(),
                                  side.data_ptr() + rows * 8, mb, st), "fwd_v")
    yc = torch.empty_like(xc)
    sidec = torch.zeros_like(side)
    _lib.check(L.fq_sym_fwd_train(xc.data_ptr(), yc.data_ptr(), rows, cols, 8, _lib.DTYPE_BF16, 0, -2.0, 2.0, sidec.data_ptr(), sidec.data_ptr() + rows * 8,
                                  mb, st), "fwd")
    assert torch.equal(y, yc) and torch.equal(side[: rows * 8], sidec[: rows * 8])
    # mask backward into a strided destination
    g = torch.randn(4, 10, 256, device="cuda").bfloat16()
    dst = torch.zeros(4, 20, 512, device="cuda", dtype=torch.bfloat16)
    gxv = dst[:, ::2, 64:320]
    t = (_lib.BwdTensorV * 1)(_lib.BwdTensorV(g.data_ptr(), gxv.data_ptr(), rows, side.data_ptr(), side.data_ptr() + rows * 8, _lib.RowsView(0, 0, 0),
                                              _lib.RowsView(10, gxv.stride(0), gxv.stride(1))))
    _lib.check(L.fq_ste_bwd_mask_multi_v(1, t, cols, -2.0, 2.0, _lib.DTYPE_BF16, 0, st), "bwd_mask_v")
    want = ops.ste_backward(g, xc, -2.0, 2.0)
    assert torch.equal(gxv, want)
    untouched = dst.clone()
    untouched[:, ::2, 64:320] = 0
    assert not untouched.any(), "the strided store wrote outside its rows"
    # x-re-reading backward with three different layouts
    gx2 = torch.empty(10, 4, 256, device="cuda", dtype=torch.bfloat16).transpose(0, 1)
    _lib.check(L.fq_ste_bwd_v(g.data_ptr(), None, x.data_ptr(), xv, gx2.data_ptr(), _lib.RowsView(10, gx2.stride(0), gx2.stride(1)), rows, cols, -2.0, 2.0, None,
                              _lib.DTYPE_BF16, st), "bwd_v")
    assert torch.equal(gx2, want)
