"""GPU tier: what the reference computes on THIS device outside autocast (`sem = FQ_SEM_DEVICE_EAGER`, the default for CUDA tensors since
round 5) against tests/golden/device_scalars.npz -- fixtures the REAL reference produced (models/utils_quant.py:31-254 on CPU tensors
under ATen's GPU scalar rules imposed from outside; tests/golden/make_golden_device_scalars.py, tests/device_scalar_policy.py).

  (i)  the HIP kernels through the C ABI and through the drop-in classes, sem = 1 == the fixtures, bit for bit (bins, values, gradients);
  (ii) the LIVE non-autocast ATen op chain on this GPU == the same fixtures: that validates the imposed scalar policy -- the only thing
       standing between the fixtures and a real GPU run of the reference.
"""
import numpy as np
import pytest
import torch

from conftest import bits_equal, golden, mismatch_report
from oracle import eager_chain as E
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TD = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def to_dev(a, dtype):
    if dtype == "fp32":
        return torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).cuda().view(TD[dtype])


def to_np(t):
    t = t.detach().contiguous().cpu()
    return t.numpy() if t.dtype == torch.float32 else t.view(torch.int16).numpy().view(np.uint16)


@pytest.fixture()
def device_eager():
    import llm_qat_amd
    from llm_qat_amd import _lib
    _lib.lib()
    prev = llm_qat_amd.get_semantics()
    llm_qat_amd.set_semantics("device_eager")
    yield llm_qat_amd
    llm_qat_amd.set_semantics(prev)


def quantizer_cases():
    return [c for c in golden("device_scalars.npz").cases if c["op"] == "quantizer"]


def test_live_aten_reproduces_the_fixture():
    """no policy imposed here: ATen's own GPU kernels, outside autocast, run the reference's op chain"""
    G = golden("device_scalars.npz")
    bad = []
    for c in quantizer_cases():
        x = to_dev(G.arr(c, "x"), c["dtype"])
        y, idx = (E.sym_forward if c["kind"] == "sym" else E.asym_forward)(x, c["bits"], c["layerwise"], want_idx=True)[:2]
        if not bits_equal(to_np(y), G.arr(c, "y"), c["dtype"]):
            bad.append(f"{c['name']}: {mismatch_report(to_np(y), G.arr(c, 'y'), c['dtype'])}")
        want_idx = G.arr(c, "idx")
        fin = np.abs(want_idx.astype(np.int64)) < 2 ** 30   # (NaN / Inf bins are coded INT32_MIN / +-(2^31 - 1): compared through y)
        if not (idx.float().cpu().numpy()[fin] == want_idx[fin]).all():
            bad.append(f"{c['name']}: live bin indices differ")
    assert not bad, "\n".join(bad[:20])


def test_kernels_match_the_fixture(device_eager):
    """debug forward (values + bin indices), training forward + mask backward, x-re-reading backward: all == the reference's"""
    ops = device_eager.ops
    G = golden("device_scalars.npz")
    bad = []
    for c in quantizer_cases():
        dt = c["dtype"]
        x = to_dev(G.arr(c, "x"), dt)
        fn = ops.sym_quantize_debug if c["kind"] == "sym" else ops.asym_quantize_debug
        y, idx, _ = fn(x, c["bits"], c["layerwise"])
        want, want_idx = G.arr(c, "y"), G.arr(c, "idx")
        if not bits_equal(to_np(y), want, dt):
            bad.append(f"{c['name']}: y {mismatch_report(to_np(y), want, dt)}")
        if not (idx.cpu().numpy().reshape(want_idx.shape) == want_idx).all():
            bad.append(f"{c['name']}: bin indices")
        g = to_dev(G.arr(c, "g"), dt)
        res = ops.train_forward(c["kind"], x, c["bits"], c["layerwise"], -2.0, 2.0)
        if res is not None:
            yt, side, rows, cols = res
            gx = ops.train_backward(g, side, rows, cols, -2.0, 2.0)
            if not bits_equal(to_np(yt), want, dt) or not bits_equal(to_np(gx), G.arr(c, "gx"), dt):
                bad.append(f"{c['name']}: training forward / mask backward")
        gx = ops.ste_backward(g, x, -2.0, 2.0)
        if not bits_equal(to_np(gx), G.arr(c, "gx"), dt):
            bad.append(f"{c['name']}: STE backward")
    assert not bad, "\n".join(bad[:20])


@pytest.mark.parametrize("mode", ["mask", "bounds", "plain"])
def test_dropin_classes_match_the_fixture(device_eager, mode):
    from llm_qat_amd.utils_quant import AsymQuantizer, SymQuantizer
    G = golden("device_scalars.npz")
    clip = torch.tensor([-2.0, 2.0])
    device_eager.set_backward_mode(mode)
    try:
        for c in quantizer_cases():
            dt = c["dtype"]
            x = to_dev(G.arr(c, "x"), dt).requires_grad_(True)
            y = (SymQuantizer if c["kind"] == "sym" else AsymQuantizer).apply(x, clip, c["bits"], c["layerwise"])
            y.backward(to_dev(G.arr(c, "g"), dt))
            assert bits_equal(to_np(y), G.arr(c, "y"), dt), f"{c['name']} [{mode}]: {mismatch_report(to_np(y), G.arr(c, 'y'), dt)}"
            assert bits_equal(to_np(x.grad), G.arr(c, "gx"), dt), f"{c['name']} [{mode}]: gradient"
    finally:
        device_eager.set_backward_mode("mask")


def test_quantize_linear_operands_and_results(device_eager):
    """module level: the operands QuantizeLinear hands to its GEMM are the oracle's (sem = 1: pinned to the fixture in the CPU tier) bit for
    bit, hence the reference's; output and gradients == the live eager chain on this GPU, and within GEMM rounding of the fixture (whose GEMM
    ran in ATen's CPU order)"""
    import torch.nn.functional as F

    from llm_qat_amd.utils_quant import QuantizeLinear
    G = golden("device_scalars.npz")
    clip = torch.tensor([-2.0, 2.0])
    for c in [c for c in G.cases if c["op"] == "quantize_linear"]:
        dt = c["dtype"]
        w, x, go = G.arr(c, "w"), G.arr(c, "x"), G.arr(c, "go")
        lin = QuantizeLinear(w.shape[1], w.shape[0], symmetric=c["symmetric"], w_bits=c["w_bits"], a_bits=c["a_bits"]).cuda().to(TD[dt])
        with torch.no_grad():
            lin.weight.copy_(to_dev(w, dt))
        xd = to_dev(x, dt).requires_grad_(True)
        seen = []
        real = F.linear
        F.linear = torch.nn.functional.linear = lambda a, b, bias=None: (seen.append((a.detach().clone(), b.detach().clone())), real(a, b, bias))[1]
        try:
            out = lin(xd)
        finally:
            F.linear = torch.nn.functional.linear = real
        out.backward(to_dev(go, dt))
        (opx, opw), = seen
        rows = x.shape[0] * x.shape[1]
        wq = O.sym_fwd(w, w.shape[0], w.shape[1], c["w_bits"], dt, sem=O.SEM_DEVICE)[0].reshape(w.shape)
        xq = (O.sym_fwd if c["symmetric"] else O.asym_fwd)(x, rows, x.shape[2], c["a_bits"], dt, sem=O.SEM_DEVICE)[0].reshape(x.shape)
        assert bits_equal(to_np(opw), wq, dt) and bits_equal(to_np(opx), xq, dt), f"{c['name']}: operands"
        # the live eager chain through the same GEMM
        wl = to_dev(w, dt).requires_grad_(True)
        xl = to_dev(x, dt).requires_grad_(True)
        QA = E.EagerSym if c["symmetric"] else E.EagerAsym
        ol = F.linear(QA.apply(xl, clip, c["a_bits"], False), E.EagerSym.apply(wl, clip, c["w_bits"], False))
        ol.backward(to_dev(go, dt))
        for a, b in ((out, ol), (xd.grad, xl.grad), (lin.weight.grad, wl.grad)):   # (fp16 rows with a tiny |max| overflow to an infinite scale: NaN == NaN here)
            assert a.dtype == b.dtype and torch.equal(a.nan_to_num(nan=7.0), b.nan_to_num(nan=7.0)), c["name"]
        ref = torch.from_numpy(G.arr(c, "out").astype(np.float32)) if dt == "fp32" else to_dev(G.arr(c, "out"), dt).float().cpu()
        got = out.detach().float().cpu()
        assert torch.equal(torch.isnan(got), torch.isnan(ref)), c["name"]
        fin = torch.isfinite(ref)
        tol = (2.0 ** -7 if dt == "bf16" else 2.0 ** -10 if dt == "fp16" else 2.0 ** -20) * ref[fin].abs().max().item() + 1e-30
        assert (got[fin] - ref[fin]).abs().max().item() <= 4 * tol, c["name"]


def test_device_eager_is_the_default_for_cuda_tensors():
    """a fresh interpreter: no environment override -> device_eager (round 5; cpu_eager remains the switch and what CPU tensors get from ATen)"""
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k != "LLMQAT_AMD_SEMANTICS"}
    out = subprocess.run([sys.executable, "-c", "import llm_qat_amd; print(llm_qat_amd.get_semantics())"], capture_output=True, text=True, env=env,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.stdout.strip().splitlines()[-1] == "device_eager", (out.stdout, out.stderr)
