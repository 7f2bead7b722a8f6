"""GPU tier: ways a training script can drive autograd around the drop-in, each against the same model built on the live eager chain
(oracle/eager_chain.py via tiny_llama.EagerQuant): three sibling QuantizeLinear layers on one input (pairing + shared activation), the two
KV hooks (speculation), STE clip hit in one weight row.  Bit-identical results -- or the same exception type -- with and without autocast.

One documented difference: the default backward does not keep a quantizer's INPUT alive, so modifying that input in
place after the forward, which makes the reference's backward raise ("modified by an inplace operation": it saved the input, :45),
goes unnoticed here and the gradient is the correct one for the forward that ran; backward mode "plain" restores the reference's error.
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tiny_llama as TL  # noqa: E402


def same(a, b):
    return len(a) == len(b) and all((x is None and y is None) or (x is not None and y is not None and x.dtype == y.dtype and x.shape == y.shape
                                                                    and torch.equal(x.nan_to_num(), y.nan_to_num()) and torch.equal(x.isnan(), y.isnan()))
                                    for x, y in zip(a, b))


def build(Q):
    mods = [Q.QuantizeLinear(64, 64, w_bits=4, a_bits=8).cuda().bfloat16() for _ in range(3)]
    with torch.no_grad():
        for k, m in enumerate(mods):
            m.weight.copy_((torch.randn(64, 64, generator=torch.Generator().manual_seed(10 + k)) * 0.4).cuda().bfloat16())
            m.weight[1, 2] = 2.5
    x = (torch.randn(5, 9, 64, generator=torch.Generator().manual_seed(3)) * 1.5).cuda().bfloat16().requires_grad_(True)
    return mods, x


def fwd(Q, mods, x, ac):
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=ac):
        q, k, v = mods[0](x), mods[1](x), mods[2](x)
        clip = torch.tensor([-2.0, 2.0])
        k = Q.SymQuantizer.apply(k, clip, 4, False)
        v = Q.SymQuantizer.apply(v, clip, 4, False)
        return torch.tanh(q.float()) * k.float() + v.float()


def retain_twice(Q, ac):
    mods, x = build(Q)
    y = fwd(Q, mods, x, ac)
    y.sum().backward(retain_graph=True)
    g1 = [x.grad.clone()] + [m.weight.grad.clone() for m in mods]
    y.sum().backward()
    return g1 + [x.grad] + [m.weight.grad for m in mods]


def grad_api(Q, ac):
    mods, x = build(Q)
    y = fwd(Q, mods, x, ac)
    go = torch.randn(9, 5, 64, generator=torch.Generator().manual_seed(7)).cuda().transpose(0, 1)   # non-contiguous grad_outputs
    return list(torch.autograd.grad(y, [x, mods[0].weight, mods[2].weight], go, allow_unused=True))


def accumulate(Q, ac):
    mods, x = build(Q)
    for _ in range(3):
        fwd(Q, mods, x, ac).sum().backward()
    return [x.grad] + [m.weight.grad for m in mods]


def nan_grads(Q, ac):
    mods, x = build(Q)
    y = fwd(Q, mods, x, ac)
    go = torch.ones_like(y)
    go[0, 0, :8], go[1, 2, 3] = float("nan"), float("inf")
    y.backward(go)
    return [x.grad] + [m.weight.grad for m in mods]


def k_without_v(Q, ac):
    mods, x = build(Q)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=ac):
        q, k = mods[0](x), mods[1](x)          # v_proj never runs: K's speculation finds no V
        k = Q.SymQuantizer.apply(k, torch.tensor([-2.0, 2.0]), 4, False)
    (q.float() * k.float()).sum().backward()
    return [x.grad] + [m.weight.grad for m in mods[:2]]


def eval_then_train(Q, ac):
    mods, x = build(Q)
    with torch.no_grad():
        y0 = fwd(Q, mods, x, ac)
    y = fwd(Q, mods, x, ac)
    y.sum().backward()
    return [y0, y.detach(), x.grad] + [m.weight.grad for m in mods]


SCENARIOS = {"backward twice with retain_graph": retain_twice, "autograd.grad, non-contiguous grad_outputs, allow_unused": grad_api,
             "three accumulated micro-steps": accumulate, "NaN / Inf in grad_output": nan_grads, "K hook without a V": k_without_v,
             "no_grad forward, then a training forward on the same input": eval_then_train}


@pytest.mark.parametrize("autocast", [False, True])
@pytest.mark.parametrize("name", list(SCENARIOS))
def test_autograd_usage_patterns_match_the_eager_chain(name, autocast):
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    llm_qat_amd.set_semantics("device_eager")
    try:
        out = []
        for Q in (TL.EagerQuant(), UQ):
            llm_qat_amd.reset_learned_state()
            try:
                out.append(SCENARIOS[name](Q, autocast))
            except Exception as e:  # noqa: BLE001
                out.append(type(e))
        want, got = out
        if isinstance(want, type) or isinstance(got, type):
            assert want is got, f"{name}: eager chain -> {want}, drop-in -> {got}"
        else:
            assert same(want, got), name
    finally:
        llm_qat_amd.set_semantics("cpu_eager")
        llm_qat_amd.reset_learned_state()


@pytest.mark.parametrize("autocast", [False, True])
def test_input_modified_in_place_after_the_forward(autocast):
    """the one documented difference (module docstring): silent and correct by default, the reference's error in backward mode "plain" """
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ

    def run(Q):
        mods, x = build(Q)
        h = x * 1.0
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            a = mods[0](h)
            h.mul_(0.5)                 # (a shared activation fake-quant must not be reused after this: the version changed)
            b = mods[1](h)
        (a.float() + b.float()).sum().backward()
        return [a.detach(), b.detach(), x.grad] + [m.weight.grad for m in mods[:2]]

    llm_qat_amd.set_semantics("device_eager")
    prev = llm_qat_amd.get_backward_mode()
    try:
        with pytest.raises(RuntimeError, match="modified by an inplace operation"):
            run(TL.EagerQuant())
        llm_qat_amd.set_backward_mode("plain")
        with pytest.raises(RuntimeError, match="modified by an inplace operation"):
            run(UQ)
        llm_qat_amd.set_backward_mode("mask")
        got = run(UQ)
        # the gradient of the forward that ran: the same graph with the in-place op replaced by an out-of-place one
        def clean(Q):
            mods, x = build(Q)
            h = x * 1.0
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                a = mods[0](h)
                b = mods[1](h * 0.5)
            (a.float() + b.float()).sum().backward()
            return [a.detach(), b.detach(), x.grad] + [m.weight.grad for m in mods[:2]]
        assert same(clean(TL.EagerQuant()), got)
    finally:
        llm_qat_amd.set_backward_mode(prev)
        llm_qat_amd.set_semantics("cpu_eager")
        llm_qat_amd.reset_learned_state()
