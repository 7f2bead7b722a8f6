"""BASELINE.json configs[0]: tiny-LLaMA (2 layers, d_model=256) QAT step -- the plumbing test.

CPU tier : this repo's harness model (tests/tiny_llama.py) driven by the eager-chain quantizers reproduces
           the REAL reference model's loss / logits / gradients (tests/golden/tiny_llama.npz), i.e. the
           harness has the reference's call sites.
GPU tier : the same harness driven by the HIP-backed drop-in (llm_qat_amd.utils_quant)
           (a) matches the reference fixture within GEMM-order tolerance, and
           (b) is BIT-IDENTICAL (loss, logits, every gradient) to the harness driven by the reference's eager
               op chain on the same GPU -- the drop-in changes nothing but speed.
"""
import json

import numpy as np
import pytest
import torch

from conftest import golden
import tiny_llama as TL

CASES = {"w8a8kv8": (8, 8, 8), "w4a8kv4": (4, 8, 4)}


def run(model, ids):
    model.zero_grad(set_to_none=True)
    loss, logits = model(ids, labels=ids)
    loss.backward()
    return loss, logits


@pytest.mark.parametrize("tag", list(CASES))
def test_harness_with_eager_chain_matches_reference_model_on_cpu(tag):
    G = golden("tiny_llama.npz")
    case = next(c for c in G.cases if c["tag"] == tag)
    w, a, kv = CASES[tag]
    torch.manual_seed(0)
    model = TL.load_deterministic(TL.TinyLlama(TL.EagerQuant(), w_bits=w, a_bits=a, kv_bits=kv).float())
    assert [n for n, _ in model.named_parameters()] == case["param_names"]
    ids = TL.deterministic_batch()
    loss, logits = run(model, ids)
    assert abs(loss.item() - float(G.z[f"{tag}/loss"][0])) < 2e-5
    np.testing.assert_allclose(logits[:, :6, :16].detach().numpy(), G.z[f"{tag}/logits_slice"], rtol=2e-4, atol=2e-5)
    norms = np.array([p.grad.double().norm().item() for _, p in model.named_parameters()])
    np.testing.assert_allclose(norms, G.z[f"{tag}/grad_norms"], rtol=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(CASES))
def test_dropin_on_gpu_matches_reference_fixture(tag):
    import llm_qat_amd.utils_quant as UQ
    G = golden("tiny_llama.npz")
    w, a, kv = CASES[tag]
    model = TL.load_deterministic(TL.TinyLlama(UQ, w_bits=w, a_bits=a, kv_bits=kv).float()).cuda()
    ids = TL.deterministic_batch().cuda()
    loss, logits = run(model, ids)
    # different GEMM accumulation order on the device moves a few values across bin edges: loose tolerance
    assert abs(loss.item() - float(G.z[f"{tag}/loss"][0])) < 5e-3
    np.testing.assert_allclose(logits[:, :6, :16].detach().cpu().numpy(), G.z[f"{tag}/logits_slice"], rtol=0.1, atol=0.02)
    norms = np.array([p.grad.double().norm().item() for _, p in model.named_parameters()])
    np.testing.assert_allclose(norms, G.z[f"{tag}/grad_norms"], rtol=0.05)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tag", list(CASES))
def test_dropin_is_bit_identical_to_eager_chain_on_gpu(tag, dtype):
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    w, a, kv = CASES[tag]
    ids = TL.deterministic_batch().cuda()
    llm_qat_amd.set_semantics("device_eager")   # compare with the op chain as ATen executes it on this device
    try:
        ours = TL.load_deterministic(TL.TinyLlama(UQ, w_bits=w, a_bits=a, kv_bits=kv).to(dtype)).cuda()
        ref = TL.load_deterministic(TL.TinyLlama(TL.EagerQuant(), w_bits=w, a_bits=a, kv_bits=kv).to(dtype)).cuda()
        l1, g1 = run(ours, ids)
        l2, g2 = run(ref, ids)
        assert torch.equal(l1, l2), (l1.item(), l2.item())
        assert torch.equal(g1, g2)
        for (n, p), (_, q) in zip(ours.named_parameters(), ref.named_parameters()):
            assert torch.equal(p.grad, q.grad), n
    finally:
        llm_qat_amd.set_semantics("cpu_eager")


@pytest.mark.gpu
def test_dropin_is_bit_identical_under_autocast():
    """kd_trainer.py:106 runs the step under HF's bf16 autocast: the quantizers are not autocast ops, so they must run
    in the dtype they are handed (fp32 master weights here) -- exactly as the eager chain does."""
    import llm_qat_amd.utils_quant as UQ
    ids = TL.deterministic_batch().cuda()
    ours = TL.load_deterministic(TL.TinyLlama(UQ, w_bits=4, a_bits=8, kv_bits=4).float()).cuda()
    ref = TL.load_deterministic(TL.TinyLlama(TL.EagerQuant(), w_bits=4, a_bits=8, kv_bits=4).float()).cuda()
    outs = []
    for m in (ours, ref):
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss, logits = m(ids, labels=ids)
        loss.backward()
        outs.append((loss.detach(), logits.detach()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for (n, p), (_, q) in zip(ours.named_parameters(), ref.named_parameters()):
        assert p.grad.dtype == torch.float32 and torch.equal(p.grad, q.grad), n


@pytest.mark.gpu
@pytest.mark.parametrize("reentrant", [True, False])
def test_dropin_under_checkpointing_and_autocast(reentrant):
    """run_train.sh's combination: bf16 weights, bf16 autocast, per-layer activation checkpointing (HF 4.30 used the
    reentrant flavour).  Drop-in == eager chain, bit for bit, with the weight cache on as well."""
    from torch.utils.checkpoint import checkpoint
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    ids = TL.deterministic_batch().cuda()

    def step(model):
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            h = model.model.embed_tokens(ids)
            h.requires_grad_(True)
            for layer in model.model.layers:
                h = checkpoint(layer, h, use_reentrant=reentrant)
            logits = model.lm_head(model.model.norm(h))
            loss = torch.nn.functional.cross_entropy(logits[..., :-1, :].reshape(-1, logits.shape[-1]).float(), ids[..., 1:].reshape(-1))
        loss.backward()
        return loss.detach(), [p.grad.clone() for p in model.parameters()]

    ref = TL.load_deterministic(TL.TinyLlama(TL.EagerQuant(), w_bits=4, a_bits=8, kv_bits=4).bfloat16()).cuda()
    ours = TL.load_deterministic(TL.TinyLlama(UQ, w_bits=4, a_bits=8, kv_bits=4).bfloat16()).cuda()
    l_ref, g_ref = step(ref)
    for cache in (False, True):
        llm_qat_amd.enable_weight_quant_cache(cache)
        try:
            l, g = step(ours)
        finally:
            llm_qat_amd.enable_weight_quant_cache(False)
        assert torch.equal(l, l_ref), (cache, l.item(), l_ref.item())
        for a, b, (n, _) in zip(g, g_ref, ours.named_parameters()):
            assert torch.equal(a, b), (cache, n)


@pytest.mark.gpu
@pytest.mark.parametrize("autocast", [False, True])
def test_dropin_with_kv_hooks_in_one_launch(autocast):
    """The optional call-site change of INTEGRATION.md (K and V through quantize_kv): the whole step stays bit-identical
    to the eager chain, plain bf16 and under bf16 autocast (where K / V come back in fp32, as in the reference)."""
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    ids = TL.deterministic_batch().cuda()
    llm_qat_amd.set_semantics("device_eager")
    try:
        ref = TL.load_deterministic(TL.TinyLlama(TL.EagerQuant(), w_bits=4, a_bits=8, kv_bits=4).bfloat16()).cuda()
        ours = TL.load_deterministic(TL.TinyLlama(UQ, w_bits=4, a_bits=8, kv_bits=4).bfloat16()).cuda()
        outs = []
        for m, one in ((ref, False), (ours, True)):
            TL.KV_ONE_LAUNCH = one
            m.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                loss, logits = m(ids, labels=ids)
            loss.backward()
            outs.append((loss.detach(), logits.detach()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        for (n, p), (_, q) in zip(ours.named_parameters(), ref.named_parameters()):
            assert torch.equal(p.grad, q.grad), n
    finally:
        TL.KV_ONE_LAUNCH = False
        llm_qat_amd.set_semantics("cpu_eager")


@pytest.mark.gpu
@pytest.mark.parametrize("reentrant", [True, False])
def test_training_steps_match_eager_chain(reentrant):
    """Several optimizer steps of run_train.sh's combination (bf16 autocast + per-layer activation checkpointing): every
    step's loss and every parameter after every step stay bit-identical to the eager chain -- a checkpointed forward and its
    recompute always build the same graph.  (The learned sibling-group dispatch of rounds 2-3 is gone: no multi-tensor launch
    ever happens at module level.)"""
    from torch.utils.checkpoint import checkpoint
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    ids = TL.deterministic_batch().cuda()

    def train(model, steps=3):
        opt = torch.optim.SGD(model.parameters(), lr=0.05)
        losses = []
        for _ in range(steps):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                h = model.model.embed_tokens(ids)
                h.requires_grad_(True)
                for layer in model.model.layers:
                    h = checkpoint(layer, h, use_reentrant=reentrant)
                logits = model.lm_head(model.model.norm(h))
                loss = torch.nn.functional.cross_entropy(logits[..., :-1, :].reshape(-1, logits.shape[-1]).float(), ids[..., 1:].reshape(-1))
            loss.backward()
            opt.step()
            losses.append(loss.detach().clone())
        return losses, [p.detach().clone() for p in model.parameters()]

    ref = TL.load_deterministic(TL.TinyLlama(TL.EagerQuant(), w_bits=4, a_bits=8, kv_bits=4).bfloat16()).cuda()
    ours = TL.load_deterministic(TL.TinyLlama(UQ, w_bits=4, a_bits=8, kv_bits=4).bfloat16()).cuda()
    calls = []
    orig = llm_qat_amd.ops.multi_forward
    llm_qat_amd.ops.multi_forward = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        l_ref, p_ref = train(ref)
        l, p = train(ours)
    finally:
        llm_qat_amd.ops.multi_forward = orig
    assert not calls
    for s, (a, b) in enumerate(zip(l, l_ref)):
        assert torch.equal(a, b), (s, a.item(), b.item())
    for a, b, (n, _) in zip(p, p_ref, ours.named_parameters()):
        assert torch.equal(a, b), n


@pytest.mark.gpu
@pytest.mark.parametrize("autocast", [False, True])
def test_conservative_mode_is_bit_identical(autocast):
    """`llm_qat_amd.conservative(True)` (LLMQAT_AMD_CONSERVATIVE=1): one launch and one autograd node per reference call, nothing
    remembered between calls, every gradient out of place.  Same loss, logits and gradients, bit for bit, as the default settings
    and as the eager chain; and it really launches per call (no pair / multi launch, 2 x 7 forward launches per layer + 2 KV hooks)."""
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    ids = TL.deterministic_batch().cuda()
    llm_qat_amd.set_semantics("device_eager")
    counts = {"pair": 0, "multi": 0}
    orig_pair, orig_multi = llm_qat_amd.ops.pair_forward, llm_qat_amd.ops.multi_forward
    llm_qat_amd.ops.pair_forward = lambda *a, **k: (counts.__setitem__("pair", counts["pair"] + 1), orig_pair(*a, **k))[1]
    llm_qat_amd.ops.multi_forward = lambda *a, **k: (counts.__setitem__("multi", counts["multi"] + 1), orig_multi(*a, **k))[1]
    try:
        outs = []
        for which in ("eager", "default", "conservative"):
            quant = TL.EagerQuant() if which == "eager" else UQ
            llm_qat_amd.conservative(which == "conservative")
            m = TL.load_deterministic(TL.TinyLlama(quant, w_bits=4, a_bits=8, kv_bits=4).bfloat16()).cuda()
            before = dict(counts)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                loss, logits = m(ids, labels=ids)
            loss.backward()
            outs.append((loss.detach(), logits.detach(), [p.grad.clone() for p in m.parameters()], [n for n, _ in m.named_parameters()]))
            if which == "default":
                assert counts["pair"] > before["pair"], "the default settings pair operands"
            if which == "conservative":
                assert counts == before, "conservative mode must not pair or group anything"
        for other in outs[1:]:
            assert torch.equal(outs[0][0], other[0]) and torch.equal(outs[0][1], other[1])
            for a, b, n in zip(outs[0][2], other[2], outs[0][3]):
                assert torch.equal(a, b), n
    finally:
        llm_qat_amd.ops.pair_forward, llm_qat_amd.ops.multi_forward = orig_pair, orig_multi
        llm_qat_amd.conservative(False)
        llm_qat_amd.set_semantics("cpu_eager")
