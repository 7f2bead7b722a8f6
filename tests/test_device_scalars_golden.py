"""CPU tier: the device-eager arithmetic (`sem = FQ_SEM_DEVICE_EAGER`: what the reference computes on a GPU OUTSIDE autocast) pinned to the
reference's own code.  tests/golden/device_scalars.npz = the real SymQuantizer / AsymQuantizer / QuantizeLinear run on CPU tensors under
ATen's GPU scalar rules imposed from outside (tests/device_scalar_policy.py; generator tests/golden/make_golden_device_scalars.py).
Here: the oracle with sem = 1 == those fixtures bit for bit (bins, values, STE gradient), the eager chain under the same policy too, and
the fixtures really are a different arithmetic from the CPU's where the generator says so.  GPU tier: tests/test_gpu_device_scalars.py.
"""
import numpy as np
import pytest
import torch

from conftest import bits_equal, golden, mismatch_report
from oracle import eager_chain as E
from oracle import oracle as O
from device_scalar_policy import DeviceScalars

TD = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def t_from(a, dtype):
    if dtype == "fp32":
        return torch.from_numpy(a.copy())
    return torch.from_numpy(a.view(np.int16).copy()).view(TD[dtype])


def np_from(t):
    t = t.detach().contiguous()
    return t.numpy() if t.dtype == torch.float32 else t.view(torch.int16).numpy().view(np.uint16)


def quantizer_cases():
    return [c for c in golden("device_scalars.npz").cases if c["op"] == "quantizer"]


def test_fixture_inventory():
    G = golden("device_scalars.npz")
    q = quantizer_cases()
    assert len(q) >= 50 and sum(c["differs_from_cpu"] for c in q) >= 12
    assert {c["dtype"] for c in q} == {"bf16", "fp16", "fp32"} and {c["kind"] for c in q} == {"sym", "asym"}
    # every fp32 AsymQuantizer case differs (the `.div(s)`), no fp32 SymQuantizer case does (nothing to rewrite there)
    for c in q:
        if c["dtype"] == "fp32":
            assert c["differs_from_cpu"] == (c["kind"] == "asym"), c["name"]
    assert G.meta["policy"] == "tests/device_scalar_policy.py"


@pytest.mark.parametrize("kind", ["sym", "asym"])
def test_oracle_device_semantics_match_the_reference_under_the_device_policy(kind):
    G = golden("device_scalars.npz")
    n = 0
    for c in quantizer_cases():
        if c["kind"] != kind:
            continue
        x, dt = G.arr(c, "x"), c["dtype"]
        rows, cols = O.rows_cols(c["shape"], c["layerwise"])
        if kind == "sym":
            y, idx, _ = O.sym_fwd(x, rows, cols, c["bits"], dt, sem=O.SEM_DEVICE)
            y0 = O.sym_fwd(x, rows, cols, c["bits"], dt, sem=O.SEM_CPU)[0]
        else:
            y, idx = O.asym_fwd(x, rows, cols, c["bits"], dt, sem=O.SEM_DEVICE)[:2]
            y0 = O.asym_fwd(x, rows, cols, c["bits"], dt, sem=O.SEM_CPU)[0]
        want = G.arr(c, "y")
        assert (idx.reshape(G.arr(c, "idx").shape) == G.arr(c, "idx")).all(), f"{c['name']}: bin indices differ"
        assert bits_equal(y.reshape(want.shape), want, dt), f"{c['name']}: {mismatch_report(y.reshape(want.shape), want, dt)}"
        # ... and the CPU policy gives other bits exactly where the generator (which ran the reference both ways) says so
        assert bits_equal(y0.reshape(want.shape), want, dt) == (not c["differs_from_cpu"]), c["name"]
        gx = O.ste_bwd(G.arr(c, "g"), x, -2.0, 2.0, dt)
        assert bits_equal(gx.reshape(want.shape), G.arr(c, "gx"), dt), f"{c['name']}: STE gradient"
        n += 1
    assert n >= 25


def test_eager_chain_under_the_device_policy_matches_the_fixture():
    """oracle/eager_chain.py executes ATen's own ops: under the same imposed policy it must give the fixture too (on the GPU, with no policy
    imposed, the same chain is the live truth: tests/test_gpu_device_scalars.py)"""
    G = golden("device_scalars.npz")
    for c in quantizer_cases():
        x = t_from(G.arr(c, "x"), c["dtype"])
        with DeviceScalars():
            y = (E.sym_forward if c["kind"] == "sym" else E.asym_forward)(x, c["bits"], c["layerwise"])
        assert bits_equal(np_from(y), G.arr(c, "y"), c["dtype"]), c["name"]


def test_module_level_cases_follow_from_the_oracle():
    """QuantizeLinear under the device policy: the operands the oracle (sem = 1) makes, multiplied by ATen's CPU GEMM, reproduce the
    reference's output and gradients (the GEMM and the STE are not touched by the scalar policy)"""
    G = golden("device_scalars.npz")
    for c in [c for c in G.cases if c["op"] == "quantize_linear"]:
        dt = c["dtype"]
        w, x, go = G.arr(c, "w"), G.arr(c, "x"), G.arr(c, "go")
        wq = O.sym_fwd(w, w.shape[0], w.shape[1], c["w_bits"], dt, sem=O.SEM_DEVICE)[0].reshape(w.shape)
        rows = x.shape[0] * x.shape[1]
        fn = O.sym_fwd if c["symmetric"] else O.asym_fwd
        xq = fn(x, rows, x.shape[2], c["a_bits"], dt, sem=O.SEM_DEVICE)[0].reshape(x.shape)
        tw, tx, tgo = t_from(wq, dt).requires_grad_(True), t_from(xq, dt).requires_grad_(True), t_from(go, dt)
        out = torch.nn.functional.linear(tx, tw)
        assert bits_equal(np_from(out), G.arr(c, "out"), dt), c["name"]
        out.backward(tgo)                                   # ATen's own dgrad / wgrad GEMMs, as in the reference's backward
        gx = O.ste_bwd(np_from(tx.grad), x, -2.0, 2.0, dt)   # then the STE masks of :83-87 on the ORIGINAL operands
        gw = O.ste_bwd(np_from(tw.grad), w, -2.0, 2.0, dt)
        assert bits_equal(gx.reshape(x.shape), G.arr(c, "gx"), dt), f"{c['name']}: input gradient"
        assert bits_equal(gw.reshape(w.shape), G.arr(c, "gw"), dt), f"{c['name']}: weight gradient"
