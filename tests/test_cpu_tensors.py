"""CPU tier: the OPT-IN CPU-tensor path of the drop-in (llm-qat_amd/cpu_tensors.py: plain torch ops in the reference's op order,
selected by `device.type == "cpu"` after llm_qat_amd.allow_cpu_tensors(True) -- the reference accepts any device,
models/utils_quant.py:37, and BASELINE configs[0] is a tiny-LLaMA QAT step on CPU) against the fixtures the real reference produced
(tests/golden/*.npz).  The default keeps raising for CPU tensors; a CUDA tensor never reaches this path (no fallback)."""
import numpy as np
import pytest
import torch

from conftest import bits_equal, golden, mismatch_report, to_f32
import tiny_llama as TL

TD = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def t_from(a, dtype):
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a) if dtype == "fp32" else torch.from_numpy(a.view(np.int16)).view(TD[dtype])


def n_from(t):
    t = t.detach().contiguous()
    return t.numpy() if t.dtype == torch.float32 else t.view(torch.int16).numpy().view(np.uint16)


@pytest.fixture
def pkg():
    import llm_qat_amd
    llm_qat_amd.allow_cpu_tensors(True)
    yield llm_qat_amd
    llm_qat_amd.allow_cpu_tensors(False)


def test_off_by_default_and_never_a_fallback():
    import llm_qat_amd
    from llm_qat_amd import cpu_tensors
    from llm_qat_amd.utils_quant import QuantizeLinear, SymQuantizer
    assert cpu_tensors.ENABLED is False
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SymQuantizer.apply(torch.randn(4, 8), torch.tensor([-2.0, 2.0]), 8, False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        QuantizeLinear(8, 4, w_bits=2, a_bits=8)(torch.randn(2, 8))
    # the functional front-end over the C ABI (ops.*) has no CPU path at all, switch or no switch
    llm_qat_amd.allow_cpu_tensors(True)
    try:
        with pytest.raises(RuntimeError, match="no CPU"):
            llm_qat_amd.ops.sym_quantize(torch.randn(4, 8), 8)
    finally:
        llm_qat_amd.allow_cpu_tensors(False)


@pytest.mark.parametrize("kind", ["sym", "asym"])
def test_forward_matches_reference_fixture(pkg, kind):
    from llm_qat_amd.utils_quant import AsymQuantizer, SymQuantizer
    G = golden(f"{kind}_fwd.npz")
    q = SymQuantizer if kind == "sym" else AsymQuantizer
    for c in G.cases:
        x = t_from(G.arr(c, "x"), c["dtype"])
        y = q.apply(x, torch.tensor([-2.0, 2.0]), c["bits"], c["layerwise"])
        assert y.dtype == x.dtype and bits_equal(n_from(y), G.arr(c, "y"), c["dtype"]), f"{c['name']}: {mismatch_report(n_from(y), G.arr(c, 'y'), c['dtype'])}"
    with pytest.raises(ValueError):
        q.apply(torch.zeros(1, 1, 1, 1, 2), torch.tensor([-2.0, 2.0]), 8, False)   # :70


def test_backward_matches_reference_fixture(pkg):
    from llm_qat_amd.utils_quant import AsymQuantizer, SymQuantizer
    G = golden("ste_bwd.npz")
    for c in G.cases:
        q = SymQuantizer if c["quant"] == "SymQuantizer" else AsymQuantizer
        x = t_from(G.arr(c, "x"), c["dtype"]).requires_grad_(True)
        q.apply(x, torch.from_numpy(G.arr(c, "clip")), c["bits"], False).backward(t_from(G.arr(c, "g"), c["dtype"]))
        a, b = n_from(x.grad), G.arr(c, "gx")
        a, b = (a.view(np.uint32), b.view(np.uint32)) if a.dtype == np.float32 else (a, b)
        assert (a == b).all(), c["name"]


def test_quantize_linear_matches_reference_fixture(pkg):
    from llm_qat_amd.utils_quant import QuantizeLinear
    G = golden("quantize_linear.npz")
    torch.set_num_threads(1)   # the fixtures' GEMMs ran single-threaded
    for c in G.cases:
        dt = c["dtype"]
        kw = {k: c[k] for k in ("w_bits", "a_bits", "symmetric", "act_layerwise", "weight_layerwise") if k in c}
        lin = QuantizeLinear(c["in_features"], c["out_features"], bias=True, **kw)
        lin.weight.data = t_from(G.arr(c, "w"), dt)
        x = t_from(G.arr(c, "x"), dt).requires_grad_(True)
        out = lin(x)
        out.backward(t_from(G.arr(c, "go"), dt))
        for name, got in (("out", out), ("gw", lin.weight.grad), ("gx", x.grad)):
            assert bits_equal(n_from(got), G.arr(c, name), dt), f"{c['name']} {name}: {mismatch_report(n_from(got), G.arr(c, name), dt)}"


def test_low_bit_weight_matches_reference_fixture(pkg):
    import torch.nn.functional as F
    from llm_qat_amd.utils_quant import QuantizeLinear
    G = golden("w12.npz")
    seen = {}
    real = F.linear

    def spy(inp, weight, bias=None):
        seen["w"] = weight.detach().clone()
        return real(inp, weight, bias)

    for c in G.cases:
        dt = c["dtype"]
        w = t_from(G.arr(c, "w"), dt)
        lin = QuantizeLinear(w.shape[1], w.shape[0], w_bits=c["w_bits"], a_bits=32, weight_layerwise=c["layerwise"])
        lin.weight.data = w
        F.linear = torch.nn.functional.linear = spy
        try:
            lin(torch.zeros(1, w.shape[1], dtype=TD[dt]))
        finally:
            F.linear = torch.nn.functional.linear = real
        assert bits_equal(n_from(seen["w"]), G.arr(c, "wq"), dt), c["name"]


@pytest.mark.parametrize("tag", ["w8a8kv8", "w4a8kv4"])
def test_tiny_llama_qat_step_on_cpu_through_the_dropin(pkg, tag):
    """BASELINE.json configs[0] as written -- tiny-LLaMA (2 layers, d_model 256) QAT step ON CPU -- through the drop-in classes,
    against the real reference model's fixture"""
    import llm_qat_amd.utils_quant as UQ
    G = golden("tiny_llama.npz")
    w, a, kv = {"w8a8kv8": (8, 8, 8), "w4a8kv4": (4, 8, 4)}[tag]
    model = TL.load_deterministic(TL.TinyLlama(UQ, w_bits=w, a_bits=a, kv_bits=kv).float())
    ids = TL.deterministic_batch()
    loss, logits = model(ids, labels=ids)
    loss.backward()
    assert abs(loss.item() - float(G.z[f"{tag}/loss"][0])) < 2e-5
    np.testing.assert_allclose(logits[:, :6, :16].detach().numpy(), G.z[f"{tag}/logits_slice"], rtol=2e-4, atol=2e-5)
    norms = np.array([p.grad.double().norm().item() for _, p in model.named_parameters()])
    np.testing.assert_allclose(norms, G.z[f"{tag}/grad_norms"], rtol=2e-3)
