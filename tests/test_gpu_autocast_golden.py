"""GPU tier: the arithmetic LLM-QAT trains with (run_train.sh:17-18 --bf16 -> utils/kd_trainer.py:106: the student runs under
torch.autocast("cuda", bf16)) against tests/golden/autocast.npz -- fixtures the REAL reference produced (its own SymQuantizer /
QuantizeLinear, models/utils_quant.py:31-87,:165-254, on CPU tensors under CUDA autocast's cast policy;
tests/golden/make_golden_autocast.py, tests/autocast_policy.py).

  (i)  the HIP kernels through the C ABI and through the drop-in classes == the fixtures, bit for bit;
  (ii) the live `torch.autocast("cuda")` ATen chain on this GPU == the same fixtures: that validates the emulated cast policy
       the fixtures were generated under (the only thing standing between them and a real CUDA-autocast run of the reference).
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import bits_equal, golden, mismatch_report, to_f32
from oracle import eager_chain as E
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TD = {"bf16": torch.bfloat16, "fp16": torch.float16}
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def dev16(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).cuda().view(TD[dtype])


def np16(t):
    return t.detach().contiguous().cpu().view(torch.int16).numpy().view(np.uint16)


def np32(t):
    return t.detach().contiguous().cpu().numpy()


@pytest.fixture(scope="module")
def lib():
    import llm_qat_amd
    from llm_qat_amd import _lib
    llm_qat_amd.set_semantics("cpu_eager")
    return _lib


def sems(case, lib):
    return {"cpu": (lib.SEM_CPU_EAGER,), "device": (lib.SEM_DEVICE_EAGER,), "both": (lib.SEM_CPU_EAGER, lib.SEM_DEVICE_EAGER)}[case["scalars"]]


def served_by_mask(rows, cols):
    return cols % 8 == 0


def test_kernels_match_the_reference_autocast_fixture(lib):
    """fq_sym_fwd_autocast (fp32 result / rounded once to the tensor dtype, with and without the training-mode side outputs),
    the pair launch in both autocast modes, fq_ste_bwd_mask_wide and fq_sym_export(autocast=1): every output == the fixture"""
    G = golden("autocast.npz")
    L = lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    bad = []
    n_mask = n_pair = 0
    for c in G.cases:
        dt, bits, code = c["dtype"], c["bits"], {"bf16": lib.DTYPE_BF16, "fp16": lib.DTYPE_F16}[c["dtype"]]
        rows, cols = O.rows_cols(tuple(c["shape"]), c["layerwise"])
        x = dev16(G.arr(c, "x"), dt).reshape(rows, cols)
        y_want, yn_want, idx_want = G.arr(c, "y").reshape(rows, cols), G.arr(c, "y_narrow").reshape(rows, cols), G.arr(c, "idx").reshape(rows, cols)
        clip = G.arr(c, "clip")
        lo, hi = float(clip[0]), float(clip[1])
        g = torch.from_numpy(G.arr(c, "g")).cuda().reshape(rows, cols)
        gx_want = G.arr(c, "gx").reshape(rows, cols)
        wsb = L.fq_rowwise_workspace_bytes(rows, cols, code)
        ws = torch.empty(max(wsb, 8), dtype=torch.uint8, device="cuda")
        for sem in sems(c, lib):
            tag = f"{c['name']} sem={sem}"
            y = torch.empty(rows, cols, device="cuda")
            yn = torch.empty(rows, cols, device="cuda", dtype=TD[dt])
            lib.check(L.fq_sym_fwd_autocast(x.data_ptr(), y.data_ptr(), rows, cols, bits, code, sem, 1, lo, hi, None, None, 0, ws.data_ptr(), wsb, st), tag)
            lib.check(L.fq_sym_fwd_autocast(x.data_ptr(), yn.data_ptr(), rows, cols, bits, code, sem, 0, lo, hi, None, None, 0, ws.data_ptr(), wsb, st), tag)
            if not bits_equal(np32(y), y_want, "fp32"):
                bad.append(f"{tag}: wide {mismatch_report(np32(y), y_want, 'fp32')}")
            if not bits_equal(np16(yn), yn_want, dt):
                bad.append(f"{tag}: narrow {mismatch_report(np16(yn), yn_want, dt)}")
            # export: the bins are the reference's torch.round output wherever the container holds them
            bins = torch.empty(L.fq_export_bins_bytes(rows, cols, lib.BINS_INT16), dtype=torch.uint8, device="cuda")
            scales = torch.empty(rows, 2, device="cuda")
            over = torch.empty(rows, dtype=torch.int32, device="cuda")
            lib.check(L.fq_sym_export(x.data_ptr(), bins.data_ptr(), scales.data_ptr(), over.data_ptr(), rows, cols, bits, lib.BINS_INT16, code, sem, 1, st), tag)
            got = bins.view(torch.int16).reshape(rows, cols).cpu().numpy().astype(np.int32)
            fits = (idx_want >= -32768) & (idx_want <= 32767)
            if not (got[fits] == idx_want[fits]).all() or not (over.cpu().numpy() == (~fits).sum(axis=1)).all():
                bad.append(f"{tag}: export bins")
            if not bits_equal(np32(scales[:, 0]), G.arr(c, "scale"), "fp32"):
                bad.append(f"{tag}: export scale")
            mb = L.fq_ste_mask_bytes(rows, cols, code)
            if not mb or cols % 4 or cols > 32768:
                continue
            # training mode: fp32 result + bounds + mask -> fq_ste_bwd_mask_wide (fp32 gradient in, 16-bit gradient out)
            n_mask += 1
            side = torch.zeros(rows * 8 + mb, dtype=torch.uint8, device="cuda")
            y2 = torch.empty(rows, cols, device="cuda")
            rc = L.fq_sym_fwd_autocast(x.data_ptr(), y2.data_ptr(), rows, cols, bits, code, sem, 1, lo, hi, side.data_ptr(), side.data_ptr() + rows * 8, mb, None, 0, st)
            lib.check(rc, tag + " wide+mask")
            gx = torch.empty(rows, cols, device="cuda", dtype=TD[dt])
            lib.check(L.fq_ste_bwd_mask_wide(g.data_ptr(), gx.data_ptr(), rows, side.data_ptr(), side.data_ptr() + rows * 8, None, None, 0, None, None,
                                             cols, lo, hi, code, st), tag + " wide bwd")
            if not bits_equal(np32(y2), y_want, "fp32") or not bits_equal(np16(gx), gx_want, dt):
                bad.append(f"{tag}: wide+mask fwd / bwd: {mismatch_report(np16(gx), gx_want, dt)}")
            # narrow result + mask -> fq_ste_bwd_mask on the gradient already cast (what QuantizeLinear's operands take)
            side.zero_()
            rc = L.fq_sym_fwd_autocast(x.data_ptr(), yn.data_ptr(), rows, cols, bits, code, sem, 0, lo, hi, side.data_ptr(), side.data_ptr() + rows * 8, mb, None, 0, st)
            lib.check(rc, tag + " narrow+mask")
            g16 = g.to(TD[dt])
            gxn = torch.empty_like(g16)
            lib.check(L.fq_ste_bwd_mask(g16.data_ptr(), gxn.data_ptr(), rows, cols, lo, hi, side.data_ptr(), side.data_ptr() + rows * 8, mb, code, st), tag)
            if not bits_equal(np16(yn), yn_want, dt) or not bits_equal(np16(gxn), gx_want, dt):
                bad.append(f"{tag}: narrow+mask fwd / bwd")
            # the pair launch (this tensor twice, as weight + input / K + V), autocast = 1 and 2
            n_pair += 1
            for ac, odt in ((1, TD[dt]), (2, torch.float32)):
                y0, y1 = torch.empty(rows, cols, device="cuda", dtype=odt), torch.empty(rows, cols, device="cuda", dtype=odt)
                s0, s1 = torch.zeros_like(side), torch.zeros_like(side)
                rc = L.fq_sym_fwd_pair(x.data_ptr(), y0.data_ptr(), rows, bits, s0.data_ptr(), s0.data_ptr() + rows * 8, mb,
                                       x.data_ptr(), y1.data_ptr(), rows, bits, s1.data_ptr(), s1.data_ptr() + rows * 8, mb, cols, code, sem, ac, lo, hi, st)
                lib.check(rc, tag + f" pair ac={ac}")
                for yy in (y0, y1):
                    ok = bits_equal(np32(yy), y_want, "fp32") if ac == 2 else bits_equal(np16(yy), yn_want, dt)
                    if not ok:
                        bad.append(f"{tag}: pair autocast={ac}")
    torch.cuda.synchronize()
    assert not bad, "\n".join(bad[:20])
    assert n_mask >= 40 and n_pair >= 40, (n_mask, n_pair)


def test_dropin_classes_under_torch_autocast_match_the_fixture(lib):
    """SymQuantizer.apply under the real torch.autocast("cuda") (every backward data flow) == the reference fixture's fp32 result
    and gradient; the operand form (_SymQuantizerOperand) == the fixture's result rounded once"""
    import llm_qat_amd
    from llm_qat_amd.utils_quant import SymQuantizer, _SymQuantizerOperand
    G = golden("autocast.npz")
    prev = llm_qat_amd.get_backward_mode()
    try:
        for mode in ("mask", "bounds", "plain"):
            llm_qat_amd.set_backward_mode(mode)
            for c in G.cases:
                if c["scalars"] == "cpu":
                    continue   # a real autocast run is on the device: device scalars
                dt = c["dtype"]
                x = dev16(G.arr(c, "x"), dt).reshape(c["shape"])
                clip = torch.from_numpy(G.arr(c, "clip"))
                g = torch.from_numpy(G.arr(c, "g")).cuda().reshape(c["shape"])
                with torch.autocast("cuda", dtype=TD[dt]):
                    xo = x.clone().requires_grad_(True)
                    y = SymQuantizer.apply(xo, clip, c["bits"], c["layerwise"])
                    xn = x.clone().requires_grad_(True)
                    yn = _SymQuantizerOperand.apply(xn, clip, c["bits"], c["layerwise"])
                y.backward(g)
                yn.backward(g.to(TD[dt]))
                tag = f"{c['name']} {mode}"
                assert y.dtype == torch.float32 and bits_equal(np32(y).reshape(-1), G.arr(c, "y").reshape(-1), "fp32"), tag
                assert yn.dtype == TD[dt] and bits_equal(np16(yn).reshape(-1), G.arr(c, "y_narrow").reshape(-1), dt), tag + " narrow"
                assert bits_equal(np16(xo.grad).reshape(-1), G.arr(c, "gx").reshape(-1), dt), tag + " grad"
                assert bits_equal(np16(xn.grad).reshape(-1), G.arr(c, "gx").reshape(-1), dt), tag + " narrow grad"
    finally:
        llm_qat_amd.set_backward_mode(prev)


def test_live_cuda_autocast_reproduces_the_fixture():
    """(ii) the live ATen op chain (oracle/eager_chain.py: the reference's ops in the reference's order) under the REAL
    torch.autocast("cuda") == the fixtures generated under the EMULATED policy: values, dtypes, gradients"""
    G = golden("autocast.npz")
    n = 0
    for c in G.cases:
        if c["scalars"] == "cpu":
            continue
        dt = c["dtype"]
        x = dev16(G.arr(c, "x"), dt).reshape(c["shape"])
        clip = torch.from_numpy(G.arr(c, "clip"))
        with torch.autocast("cuda", dtype=TD[dt]):
            xr = x.clone().requires_grad_(True)
            y, idx, s = E.sym_forward(xr.detach(), c["bits"], c["layerwise"], want_idx=True)
            yr = E.EagerSym.apply(xr, clip, c["bits"], c["layerwise"])
        yr.backward(torch.from_numpy(G.arr(c, "g")).cuda().reshape(c["shape"]))
        assert yr.dtype == torch.float32 and idx.dtype == torch.float32 and s.dtype == torch.float32, c["name"]
        assert bits_equal(np32(yr).reshape(-1), G.arr(c, "y").reshape(-1), "fp32"), f"{c['name']}: {mismatch_report(np32(yr).reshape(-1), G.arr(c, 'y').reshape(-1), 'fp32')}"
        fin = torch.isfinite(idx)
        assert (idx[fin].to(torch.int64).cpu().numpy() == G.arr(c, "idx").reshape(c["shape"])[fin.cpu().numpy()]).all(), c["name"] + " idx"
        assert bits_equal(np16(xr.grad).reshape(-1), G.arr(c, "gx").reshape(-1), dt), c["name"] + " grad"
        n += 1
    assert n >= 50


def linear_cases(G):
    return json.loads(bytes(G.z["manifest"]).decode())["linear_cases"]


@pytest.mark.parametrize("which", ["dropin", "live_aten"])
def test_quantize_linear_under_autocast_matches_the_fixture(lib, which):
    """QuantizeLinear.forward + backward under torch.autocast("cuda") against the module-level fixture: the operands handed to
    the GEMM BIT-EXACT (reference: fp32 fake-quant results cast by F.linear's autocast; drop-in: the narrow kernels), out /
    gradients within the GEMM's tolerance (rocBLAS on the device vs the CPU's accumulation order).  `live_aten` runs the eager
    chain module instead of the drop-in: the same fixture validates the emulated policy at module level."""
    import torch.nn.functional as F
    import llm_qat_amd
    import tiny_llama as TL
    from llm_qat_amd.utils_quant import QuantizeLinear
    G = golden("autocast.npz")
    QL = QuantizeLinear if which == "dropin" else TL.EagerQuant().QuantizeLinear
    real = F.linear
    seen = []

    def spy(inp, weight, bias=None):
        adt = torch.get_autocast_dtype("cuda")
        seen.append((inp.detach().to(adt), weight.detach().to(adt)))   # what the GEMM sees after autocast's own cast (a no-op for the drop-in's narrow operands)
        return real(inp, weight, bias)

    n = 0
    try:
        for c in linear_cases(G):
            if c["scalars"] == "cpu" and c["symmetric"]:
                continue
            if which == "live_aten" and c["w_bits"] < 3:
                continue   # the eager-chain harness module does not restate the 1-/2-bit weight branches
            dt, adt = c["dtype"], c["autocast_dtype"]
            llm_qat_amd.set_semantics("cpu_eager" if c["scalars"] == "cpu" else "device_eager")   # (governs the Asym activation case only)
            kw = {k: c[k] for k in ("w_bits", "a_bits", "symmetric", "act_layerwise", "weight_layerwise") if k in c}
            lin = QL(c["in_features"], c["out_features"], **kw).cuda()
            lin.weight.data = dev16(G.z[f"{c['name']}/w"], dt)
            x = dev16(G.z[f"{c['name']}/x"], dt).requires_grad_(True)
            del seen[:]
            F.linear = torch.nn.functional.linear = spy
            try:
                with torch.autocast("cuda", dtype=TD[adt]):
                    out = lin(x)
            finally:
                F.linear = torch.nn.functional.linear = real
            out.backward(dev16(G.z[f"{c['name']}/go"], adt))
            (opx, opw), = seen
            tag = f"{which} {c['name']}"
            assert out.dtype == TD[adt], tag
            assert bits_equal(np16(opw), G.z[f"{c['name']}/opw"], adt), f"{tag}: weight operand {mismatch_report(np16(opw), G.z[c['name'] + '/opw'], adt)}"
            assert bits_equal(np16(opx), G.z[f"{c['name']}/opx"], adt), f"{tag}: input operand {mismatch_report(np16(opx), G.z[c['name'] + '/opx'], adt)}"
            tol = dict(rtol=2e-2, atol=2e-2) if adt == "bf16" else dict(rtol=4e-3, atol=4e-3)
            for name, got, d in (("out", out, adt), ("gw", lin.weight.grad, dt), ("gx", x.grad, dt)):
                assert got.dtype == TD[d], f"{tag} {name} dtype"
                want = torch.from_numpy(to_f32(G.z[f"{c['name']}/{name}"], d))
                torch.testing.assert_close(got.float().cpu(), want, msg=lambda m: f"{tag} {name}: {m}", **tol)
            if 3 <= c["w_bits"] < 32:   # the STE mask on the weight gradient is exact
                gw = lin.weight.grad.float().cpu()
                assert gw[3, 5] == 0 and gw[4, 6] == 0, tag
            n += 1
    finally:
        llm_qat_amd.set_semantics("cpu_eager")
    assert n >= (18 if which == "dropin" else 10), n
