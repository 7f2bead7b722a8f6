"""GPU tier: the shared activation fake-quant (sibling projections that quantize the SAME input with the same settings get one launch and
one autograd node) under call sequences that could fool it -- against the same calls on the live eager chain (tiny_llama.EagerQuant):
requires_grad switched on the input between two siblings, the same module twice, a sibling under no_grad then one with grad, autocast
toggled between siblings, a tensor hook on the shared input, the siblings' losses backwarded separately.  Bit-identical.

One documented limit (llm-qat_amd/utils_quant.py, point 1): if ANOTHER consumer of the same input is created after the sharing siblings,
the input's gradient is the same sum in a different association order -- bit-identical again with share_activation_quant(False)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tiny_llama as TL  # noqa: E402


def same(a, b):
    return len(a) == len(b) and all((x is None and y is None) or (x is not None and y is not None and x.dtype == y.dtype and x.shape == y.shape
                                                                    and torch.equal(x.nan_to_num(), y.nan_to_num())) for x, y in zip(a, b))


def mk(Q, ab=8, sym=True, alw=False, d=64, seed=0):
    m = Q.QuantizeLinear(d, d, w_bits=4, a_bits=ab, symmetric=sym, act_layerwise=alw).cuda().bfloat16()
    with torch.no_grad():
        m.weight.copy_((torch.randn(d, d, generator=torch.Generator().manual_seed(30 + seed)) * 0.4).cuda().bfloat16())
    return m


def X(grad=True):
    return (torch.randn(2, 9, 64, generator=torch.Generator().manual_seed(3)) * 1.5).cuda().bfloat16().requires_grad_(grad)


def _ctx(ac):
    return torch.autocast("cuda", dtype=torch.bfloat16, enabled=ac)


def grad_switched_on(Q, ac):
    m0, m1, x = mk(Q, seed=0), mk(Q, seed=1), X(False)
    with _ctx(ac):
        a = m0(x)
        x.requires_grad_(True)     # (no version bump: the remembered output has no graph and must not be handed to the second sibling)
        b = m1(x)
    (a.float() + 2 * b.float()).sum().backward()
    return [a.detach(), b.detach(), x.grad, m0.weight.grad, m1.weight.grad]


def grad_switched_off(Q, ac):
    m0, m1, x = mk(Q, seed=0), mk(Q, seed=1), X(True)
    with _ctx(ac):
        a = m0(x)
        x.requires_grad_(False)
        b = m1(x)
    (a.float() + 2 * b.float()).sum().backward()
    return [a.detach(), b.detach(), x.grad, m0.weight.grad, m1.weight.grad]


def same_module_twice(Q, ac):
    m0, x = mk(Q), X()
    with _ctx(ac):
        a, b = m0(x), m0(x)
    (a.float() + 2 * b.float()).sum().backward()
    return [a.detach(), b.detach(), x.grad, m0.weight.grad]


def nograd_then_grad(Q, ac):
    m0, m1, x = mk(Q, seed=0), mk(Q, seed=1), X()
    with _ctx(ac):
        with torch.no_grad():
            a = m0(x)
        b = m1(x)
    b.float().sum().backward()
    return [a, b.detach(), x.grad, m1.weight.grad]


def autocast_toggled(Q, ac):
    m0, m1, x = mk(Q, seed=0), mk(Q, seed=1), X()
    with _ctx(ac):
        a = m0(x)
    with _ctx(not ac):
        b = m1(x)
    (a.float() + 2 * b.float()).sum().backward()
    return [a.detach(), b.detach(), x.grad, m0.weight.grad, m1.weight.grad]


def hook_on_input(Q, ac):
    m0, m1, x = mk(Q, seed=0), mk(Q, seed=1), X()
    h = x * 1.0
    seen = []
    h.register_hook(lambda g: seen.append(g.clone()))
    with _ctx(ac):
        a, b = m0(h), m1(h)
    (a.float() + 2 * b.float()).sum().backward()
    return [a.detach(), b.detach(), x.grad] + seen


def separate_backwards(Q, ac):
    m0, m1, x = mk(Q, seed=0), mk(Q, seed=1), X()
    with _ctx(ac):
        a, b = m0(x), m1(x)
    a.float().sum().backward(retain_graph=True)
    g1 = x.grad.clone()
    b.float().sum().backward()
    return [g1, x.grad, m0.weight.grad, m1.weight.grad]


def asym_and_layerwise_siblings(Q, ac):
    ms, x = [mk(Q, ab=8, sym=False, seed=3), mk(Q, ab=8, sym=False, seed=5), mk(Q, ab=8, alw=True, seed=4), mk(Q, ab=8, alw=True, seed=6)], X()
    with _ctx(ac):
        outs = [m(x) for m in ms]
    sum(o.float().sum() * (i + 1) for i, o in enumerate(outs)).backward()
    return [o.detach() for o in outs] + [m.weight.grad for m in ms]     # (x.grad: see the association-order test)


SCENARIOS = {f.__name__: f for f in (grad_switched_on, grad_switched_off, same_module_twice, nograd_then_grad, autocast_toggled, hook_on_input,
                                     separate_backwards, asym_and_layerwise_siblings)}


@pytest.mark.parametrize("autocast", [False, True])
@pytest.mark.parametrize("name", list(SCENARIOS))
def test_shared_activation_sequences_match_the_eager_chain(name, autocast):
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    llm_qat_amd.set_semantics("device_eager")
    try:
        llm_qat_amd.reset_learned_state()
        want = SCENARIOS[name](TL.EagerQuant(), autocast)
        llm_qat_amd.reset_learned_state()
        assert same(want, SCENARIOS[name](UQ, autocast)), name
    finally:
        llm_qat_amd.set_semantics("cpu_eager")
        llm_qat_amd.reset_learned_state()


@pytest.mark.parametrize("autocast", [False, True])
def test_a_later_consumer_changes_only_the_association_order_of_the_input_gradient(autocast):
    """A8, A4, A8 again, on one input: the two A8 siblings share one node, the A4 module in between is another consumer of x.  Outputs and
    weight gradients bit-identical; x.grad is the same three-term sum in another association order (close, not equal);
    with sharing off: bit-identical."""
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ

    def run(Q):
        ms, x = [mk(Q, ab=8, seed=0), mk(Q, ab=4, seed=1), mk(Q, ab=8, seed=2)], X()
        with _ctx(autocast):
            outs = [m(x) for m in ms]
        sum(o.float().sum() * (i + 1) for i, o in enumerate(outs)).backward()
        return [o.detach() for o in outs] + [m.weight.grad for m in ms], x.grad

    llm_qat_amd.set_semantics("device_eager")
    try:
        llm_qat_amd.reset_learned_state()
        want, want_gx = run(TL.EagerQuant())
        llm_qat_amd.reset_learned_state()
        got, got_gx = run(UQ)
        assert same(want, got)
        # a bf16 rounding or two of the partial sums, measured against the size of the terms (small sums are differences of large ones)
        assert float((got_gx.float() - want_gx.float()).abs().max()) <= 2 ** -6 * float(want_gx.float().abs().max())
        llm_qat_amd.share_activation_quant(False)
        llm_qat_amd.reset_learned_state()
        got, got_gx = run(UQ)
        assert same(want, got) and torch.equal(got_gx, want_gx)
    finally:
        llm_qat_amd.share_activation_quant(True)
        llm_qat_amd.set_semantics("cpu_eager")
        llm_qat_amd.reset_learned_state()


def test_clip_val_requiring_grad_while_the_input_does_not():
    """the reference's backward returns None for clip_val (:87) whatever needs a gradient; with an input that needs none the drop-in records
    nothing in the forward and must still answer the engine"""
    import llm_qat_amd.utils_quant as UQ
    from oracle import eager_chain as E
    x = torch.randn(4, 64, device="cuda").bfloat16()
    for q in (E.EagerSym, UQ.SymQuantizer, UQ.AsymQuantizer):
        clip = torch.tensor([-2.0, 2.0], requires_grad=True)
        y = q.apply(x, clip, 8, False)
        if y.requires_grad:
            y.float().sum().backward()
        assert clip.grad is None
    xg = x.clone().requires_grad_(True)
    clip = torch.tensor([-2.0, 2.0], requires_grad=True)
    UQ.SymQuantizer.apply(xg, clip, 8, False).float().sum().backward()
    assert clip.grad is None and xg.grad is not None
