"""GPU tier: the KV-cache hooks called in every order a model could call them, with the K+V speculation of the drop-in at its default
(on): V before K, K twice, different clips or bit widths for K and V, two attention blocks interleaved, hooks on views of the projections'
outputs, a detached K, a V that is recomputed or modified in place between the two hooks, hooks under no_grad -- against the same calls on
the live eager chain (tiny_llama.EagerQuant).  Every result and gradient bit-identical; a wrong guess costs one discarded launch and is
learned (llm_qat_amd.stats()), never a wrong value.  Reference call site: models/modeling_llama_quant.py:317-327."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tiny_llama as TL  # noqa: E402


@pytest.fixture(autouse=True, params=["c++", "python"])
def node(request):
    """every test of this file with either kind of autograd node behind K and V (csrc/fq_autograd_node.cpp::FqOneNode / _PrecomputedAct)"""
    import llm_qat_amd
    assert llm_qat_amd.cpp_node(request.param == "c++") == (request.param == "c++"), llm_qat_amd.host_node()
    yield request.param
    llm_qat_amd.cpp_node(True)


def same(a, b):
    return len(a) == len(b) and all((x is None and y is None) or (x is not None and y is not None and x.dtype == y.dtype and x.shape == y.shape
                                                                    and torch.equal(x.nan_to_num(), y.nan_to_num())) for x, y in zip(a, b))


def build(Q, n=4, d=64):
    mods = [Q.QuantizeLinear(d, d, w_bits=4, a_bits=8).cuda().bfloat16() for _ in range(n)]
    with torch.no_grad():
        for k, m in enumerate(mods):
            m.weight.copy_((torch.randn(d, d, generator=torch.Generator().manual_seed(20 + k)) * 0.4).cuda().bfloat16())
    x = (torch.randn(2, 9, d, generator=torch.Generator().manual_seed(3)) * 1.5).cuda().bfloat16().requires_grad_(True)
    return mods, x


def C(lo=-2.0, hi=2.0):
    return torch.tensor([lo, hi])


def finish(outs, mods, x):
    sum(o.float().sum() * (i + 1) for i, o in enumerate(outs)).backward()
    return [o.detach() for o in outs] + [x.grad] + [m.weight.grad for m in mods]


def _ctx(ac):
    return torch.autocast("cuda", dtype=torch.bfloat16, enabled=ac)


def v_before_k(Q, ac):
    mods, x = build(Q)
    with _ctx(ac):
        k, v = mods[0](x), mods[1](x)
        v = Q.SymQuantizer.apply(v, C(), 4, False)
        k = Q.SymQuantizer.apply(k, C(), 4, False)
    return finish([k, v], mods[:2], x)


def k_twice(Q, ac):
    mods, x = build(Q)
    with _ctx(ac):
        k, v = mods[0](x), mods[1](x)
        k1 = Q.SymQuantizer.apply(k, C(), 4, False)
        k2 = Q.SymQuantizer.apply(k, C(), 4, False)
        v = Q.SymQuantizer.apply(v, C(), 4, False)
    return finish([k1, k2, v], mods[:2], x)


def diff_clip(Q, ac):
    mods, x = build(Q)
    with _ctx(ac):
        k, v = mods[0](x), mods[1](x)
        k = Q.SymQuantizer.apply(k, C(-2.0, 2.0), 4, False)
        v = Q.SymQuantizer.apply(v, C(-1.0, 1.5), 4, False)
    return finish([k, v], mods[:2], x)


def diff_bits(Q, ac):
    mods, x = build(Q)
    with _ctx(ac):
        k, v = mods[0](x), mods[1](x)
        k = Q.SymQuantizer.apply(k, C(), 4, False)
        v = Q.SymQuantizer.apply(v, C(), 8, False)
    return finish([k, v], mods[:2], x)


def interleaved(Q, ac):
    mods, x = build(Q)
    with _ctx(ac):
        k1, v1, k2, v2 = mods[0](x), mods[1](x), mods[2](x), mods[3](x)
        k1 = Q.SymQuantizer.apply(k1, C(), 4, False)
        k2 = Q.SymQuantizer.apply(k2, C(), 4, False)
        v1 = Q.SymQuantizer.apply(v1, C(), 4, False)
        v2 = Q.SymQuantizer.apply(v2, C(), 4, False)
    return finish([k1, v1, k2, v2], mods, x)


def views(Q, ac):
    mods, x = build(Q)
    with _ctx(ac):
        k, v = mods[0](x), mods[1](x)
        k = Q.SymQuantizer.apply(k.view(2, 9, 4, 16), C(), 4, False)
        v = Q.SymQuantizer.apply(v.transpose(0, 1), C(), 4, False)
    return finish([k, v], mods[:2], x)


def detached_k(Q, ac):
    mods, x = build(Q)
    with _ctx(ac):
        k, v = mods[0](x), mods[1](x)
        k = Q.SymQuantizer.apply(k.detach(), C(), 4, False)
        v = Q.SymQuantizer.apply(v, C(), 4, False)
    return finish([v], mods[1:2], x) + [k]


def recomputed_v(Q, ac):
    mods, x = build(Q)
    with _ctx(ac):
        k, v = mods[0](x), mods[1](x)
        k = Q.SymQuantizer.apply(k, C(), 4, False)
        v = Q.SymQuantizer.apply(v * 2.0, C(), 4, False)
    return finish([k, v], mods[:2], x)


def v_inplace(Q, ac):
    mods, x = build(Q)
    with _ctx(ac):
        k, v = mods[0](x), mods[1](x)
        k = Q.SymQuantizer.apply(k, C(), 4, False)
        v.add_(1.0)
        v = Q.SymQuantizer.apply(v, C(), 4, False)
    return finish([k, v], mods[:2], x)


def results_modified_in_place(Q, ac):
    """the reference's hook results are fresh tensors of their Functions: in-place arithmetic on them (here a scale and a shift) is ordinary
    autograd -- the drop-in's K and V results are the launch's own tensors too, not views handed out by a node (round 5)"""
    mods, x = build(Q)
    with _ctx(ac):
        k, v = mods[0](x), mods[1](x)
        k = Q.SymQuantizer.apply(k, C(), 4, False)
        v = Q.SymQuantizer.apply(v, C(), 4, False)
        k.mul_(0.5)
        v.add_(0.25)
    return finish([k, v], mods[:2], x)


def nograd_hooks(Q, ac):
    mods, x = build(Q)
    with _ctx(ac), torch.no_grad():
        k, v = mods[0](x), mods[1](x)
        k = Q.SymQuantizer.apply(k, C(), 4, False)
        v = Q.SymQuantizer.apply(v, C(), 4, False)
    return [k, v]


# name -> (scenario, what stats() must show for the drop-in: launched, hit, discarded)
SCENARIOS = {"V before K": (v_before_k, (0, 0, 0)), "K twice, then V": (k_twice, (1, 0, 1)), "different clips": (diff_clip, (1, 0, 1)),
             "K 4-bit, V 8-bit": (diff_bits, (1, 0, 1)), "two blocks interleaved": (interleaved, (1, 0, 1)), "hooks on views": (views, (0, 0, 0)),
             "K detached": (detached_k, (0, 0, 0)), "V recomputed between the hooks": (recomputed_v, (1, 0, 1)),
             "V modified in place between the hooks": (v_inplace, (1, 0, 1)), "hooks under no_grad": (nograd_hooks, (1, 1, 0)),
             "results modified in place": (results_modified_in_place, (1, 1, 0))}


@pytest.mark.parametrize("autocast", [False, True])
@pytest.mark.parametrize("name", list(SCENARIOS))
def test_kv_hook_sequences_match_the_eager_chain(name, autocast):
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    fn, (launched, hit, discarded) = SCENARIOS[name]
    llm_qat_amd.set_semantics("device_eager")
    prev_mode = llm_qat_amd.get_backward_mode()
    llm_qat_amd.set_backward_mode("mask")      # the speculation (and the counters asserted below) belong to the default data flow
    try:
        llm_qat_amd.reset_learned_state()
        want = fn(TL.EagerQuant(), autocast)
        for repeat in range(2):       # the second time the call signature may have been learned off: same values either way
            llm_qat_amd.stats(reset=True)
            got = fn(UQ, autocast)
            assert same(want, got), f"{name} (repeat {repeat})"
            st = llm_qat_amd.stats()
            if repeat == 0:
                assert (st.get("kv_pair_launch", 0), st.get("kv_pair_hit", 0), st.get("kv_pair_discarded", 0)) == (launched, hit, discarded), st
            elif discarded:
                assert not st.get("kv_pair_launch"), st     # a signature that was wrong once stops speculating
    finally:
        llm_qat_amd.set_backward_mode(prev_mode)
        llm_qat_amd.set_semantics("cpu_eager")
        llm_qat_amd.reset_learned_state()


def _g(*shape, dt=torch.bfloat16, grad=True, seed=0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * 1.5).cuda().to(dt).requires_grad_(grad)


def _noncontig():
    a, b = _g(9, 2, 64), _g(9, 2, 64, seed=1)
    return a.transpose(0, 1), b.transpose(0, 1)


QKV_CASES = {
    "plain [2,9,64]": (lambda: (_g(2, 9, 64), _g(2, 9, 64, seed=1)), {}),
    "different shapes": (lambda: (_g(2, 9, 64), _g(2, 5, 64, seed=1)), {}),
    "different widths": (lambda: (_g(2, 9, 64), _g(2, 9, 32, seed=1)), {}),
    "different dtypes": (lambda: (_g(2, 9, 64), _g(2, 9, 64, dt=torch.float16, seed=1)), {}),
    "fp32 pair": (lambda: (_g(2, 9, 64, dt=torch.float32), _g(2, 9, 64, dt=torch.float32, seed=1)), {}),
    "K grad, V no grad": (lambda: (_g(2, 9, 64), _g(2, 9, 64, grad=False, seed=1)), {}),
    "neither needs grad": (lambda: (_g(2, 9, 64, grad=False), _g(2, 9, 64, grad=False, seed=1)), {}),
    "different clips": (lambda: (_g(2, 9, 64), _g(2, 9, 64, seed=1)), {"cv": (-1.0, 1.0)}),
    "4-D inputs": (lambda: (_g(2, 3, 4, 16), _g(2, 3, 4, 16, seed=1)), {}),
    "1-D inputs": (lambda: (_g(64), _g(64, seed=1)), {}),
    "no rows": (lambda: (_g(0, 64), _g(0, 64, seed=1)), {}),
    "1 bit": (lambda: (_g(2, 9, 64), _g(2, 9, 64, seed=1)), {"bits": 1}),
    "16 bits": (lambda: (_g(2, 9, 64), _g(2, 9, 64, seed=1)), {"bits": 16}),
    "the same tensor as K and V": (lambda: (lambda t: (t, t))(_g(2, 9, 64)), {}),
    "non-contiguous views": (_noncontig, {}),
    "odd width 100": (lambda: (_g(3, 100), _g(3, 100, seed=1)), {}),
}


@pytest.mark.parametrize("autocast", [False, True])
@pytest.mark.parametrize("name", list(QKV_CASES))
def test_quantize_kv_equals_the_two_reference_calls(name, autocast):
    """llm_qat_amd.quantize_kv(K, V, clip_k, clip_v, bits) -- the two-line call-site change of INTEGRATION.md -- against the two
    SymQuantizer.apply calls it stands for (modeling_llama_quant.py:320-327) on the live eager chain, whatever it is handed: it pairs what
    it can and falls back to two calls for the rest; values, gradients, dtypes (fp32 under autocast) bit-identical, or the same exception."""
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    from oracle import eager_chain as E
    mk, kw = QKV_CASES[name]
    ck, cv, bits = kw.get("ck", (-2.0, 2.0)), kw.get("cv", (-2.0, 2.0)), kw.get("bits", 4)
    llm_qat_amd.set_semantics("device_eager")
    try:
        res = []
        for which in ("ref", "got"):
            k, v = mk()
            try:
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                    if which == "ref":
                        kq, vq = E.EagerSym.apply(k, torch.tensor(ck), bits, False), E.EagerSym.apply(v, torch.tensor(cv), bits, False)
                    else:
                        kq, vq = UQ.quantize_kv(k, v, torch.tensor(ck), torch.tensor(cv), bits)
                if kq.requires_grad or vq.requires_grad:
                    (kq.float().sum() + 2 * vq.float().sum()).backward()
                res.append([kq.detach(), vq.detach(), k.grad if k.is_leaf else None, v.grad if v.is_leaf else None])
            except Exception as e:  # noqa: BLE001
                res.append(type(e))
        want, got = res
        if isinstance(want, type) or isinstance(got, type):
            assert want is got, (want, got)
        else:
            assert same(want, got), name
    finally:
        llm_qat_amd.set_semantics("cpu_eager")
        llm_qat_amd.reset_learned_state()
