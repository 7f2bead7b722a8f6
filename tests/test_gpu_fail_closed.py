"""GPU tier: the stateful host logic FAILS CLOSED when a private torch API it leans on is missing (VERDICT r04 "weak" #5).

  torch._C._autograd._top_saved_tensors_default_hooks   which saved-tensor-hooks region a call runs in: without it, remembering anything
                                                        across calls could hand a non-reentrant checkpoint's first pass something its
                                                        recompute cannot repeat -> activation sharing and K/V pairing switch themselves off
  torch._C._are_functorch_transforms_active             -> K/V pairing off
  torch._C._storage_Use_Count / Tensor._use_count       -> no weight gradient is ever masked in place

Each is monkeypatched away in turn; the programs that used to break (a checkpoint around ONE sibling projection; a hook that stashes a
weight's gradient) and a whole decoder layer must give the eager chain's results bit for bit, raise nothing, and stats() must say what
was switched off."""
import os
import sys

import pytest
import torch
from torch.utils.checkpoint import checkpoint

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tiny_llama as TL  # noqa: E402


def mk(Q, seed, d=64):
    m = Q.QuantizeLinear(d, d, w_bits=4, a_bits=8).cuda().bfloat16()
    with torch.no_grad():
        m.weight.copy_((torch.randn(d, d, generator=torch.Generator().manual_seed(40 + seed)) * 0.4).cuda().bfloat16())
    return m


def X():
    return (torch.randn(2, 9, 64, generator=torch.Generator().manual_seed(5)) * 1.5).cuda().bfloat16().requires_grad_(True)


def checkpoint_around_one_sibling(Q, ac):
    """q runs outside, k inside a non-reentrant checkpoint, v outside again: with something remembered across the region boundary the
    recompute of k would save other tensors than its first pass (CheckpointError, DESIGN round-4 table)"""
    q, k, v, x = mk(Q, 0), mk(Q, 1), mk(Q, 2), X()
    h = x * 1.0
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=ac):
        a = q(h)
        b = checkpoint(k, h, use_reentrant=False)
        c = v(h)
    (a.float() + 2 * b.float() + 3 * c.float()).sum().backward()
    return [a.detach(), b.detach(), c.detach(), x.grad, q.weight.grad, k.weight.grad, v.weight.grad]


def stashing_hook(Q, ac):
    """a tensor hook keeps a reference to a weight's incoming gradient: the in-place path must not touch what the hook holds"""
    m, x = mk(Q, 3), X()
    stash = []
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=ac):
        out = m(x)
    m.weight.register_hook(lambda g: stash.append(g))
    out.float().sum().backward()
    return [out.detach(), x.grad, m.weight.grad] + [s.clone() for s in stash]


def kv_layer(Q, ac):
    """k_proj, v_proj and the two unchanged KV hooks (modeling_llama_quant.py:317-327)"""
    kp, vp, x = mk(Q, 4), mk(Q, 5), X()
    clip = torch.tensor([-2.0, 2.0])
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=ac):
        k, v = kp(x), vp(x)
        k = Q.SymQuantizer.apply(k, clip, 4, False)
        v = Q.SymQuantizer.apply(v, clip, 4, False)
    (k.float().sum() + 2 * v.float().sum()).backward()
    return [k.detach(), v.detach(), x.grad, kp.weight.grad, vp.weight.grad]


PROGRAMS = [checkpoint_around_one_sibling, stashing_hook, kv_layer]


def same(a, b):
    return len(a) == len(b) and all(x.dtype == y.dtype and x.shape == y.shape and torch.equal(x.nan_to_num(), y.nan_to_num()) for x, y in zip(a, b))


MISSING = {
    "no_region_api": ("_top_hooks", ("share_disabled:no_region_api",)),
    "no_functorch_api": ("_functorch_active", ("kv_pair_disabled:no_functorch_api",)),
    "no_storage_use_count": ("_storage_use_count", ("inplace_refused:no_refcount_api",)),
}


@pytest.mark.parametrize("node", ["c++", "python"])
@pytest.mark.parametrize("autocast", [False, True])
@pytest.mark.parametrize("missing", list(MISSING))
def test_missing_private_api_switches_the_feature_off(monkeypatch, missing, autocast, node):
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    attr, counters = MISSING[missing]
    llm_qat_amd.set_semantics("device_eager")
    assert llm_qat_amd.cpp_node(node == "c++") == (node == "c++"), llm_qat_amd.host_node()
    try:
        want = [p(TL.EagerQuant(), autocast) for p in PROGRAMS]
        monkeypatch.setattr(UQ, attr, None)
        llm_qat_amd.reset_learned_state()
        llm_qat_amd.stats(reset=True)
        got = [p(UQ, autocast) for p in PROGRAMS]     # no CheckpointError, no AttributeError / TypeError on the missing API
        st = llm_qat_amd.stats()
        for p, w, g in zip(PROGRAMS, want, got):
            assert same(w, g), f"{missing}: {p.__name__}"
        if missing == "no_storage_use_count" and node == "c++":
            # the C++ node's guard reads the reference counts through the public C++ API (Tensor::use_count, Storage::use_count): it does
            # not lean on this private Python API at all (what it refuses: test_gpu_features.py::test_inplace_weight_gradient_is_guarded)
            assert st.get("cpp_pair_backward", 0) > 0 and st.get("inplace_taken", 0) > 0 and not st.get("inplace_refused:no_refcount_api"), st
            return
        for c in counters:
            assert st.get(c, 0) > 0, (c, st)
        if missing == "no_region_api":     # nothing remembered: no sharing, no K/V speculation
            assert not st.get("act_share_hit") and not st.get("kv_pair_launch"), st
        if missing == "no_functorch_api":
            assert not st.get("kv_pair_launch"), st
        if missing == "no_storage_use_count":
            assert not st.get("inplace_taken"), st
    finally:
        llm_qat_amd.cpp_node(True)
        llm_qat_amd.set_semantics("cpu_eager")
        llm_qat_amd.reset_learned_state()


@pytest.mark.parametrize("autocast", [False, True])
def test_the_same_programs_with_every_api_present(autocast):
    """the control: with the APIs there the features engage and the results are the same"""
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    llm_qat_amd.set_semantics("device_eager")
    try:
        want = [p(TL.EagerQuant(), autocast) for p in PROGRAMS]
        llm_qat_amd.reset_learned_state()
        llm_qat_amd.stats(reset=True)
        got = [p(UQ, autocast) for p in PROGRAMS]
        st = llm_qat_amd.stats()
        for p, w, g in zip(PROGRAMS, want, got):
            assert same(w, g), p.__name__
        assert st.get("kv_pair_launch", 0) >= 1 and st.get("inplace_taken", 0) >= 1, st
    finally:
        llm_qat_amd.set_semantics("cpu_eager")
        llm_qat_amd.reset_learned_state()
