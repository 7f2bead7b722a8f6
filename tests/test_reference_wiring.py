"""CPU tier, build container only (skipped where /root/reference is absent, e.g. on the GPU box):
the REAL reference model code (models/modeling_llama_quant.py) is constructed on top of this package's
drop-in by swapping `models.utils_quant` for `llm_qat_amd.utils_quant` -- the one-file switch of INTEGRATION.md.
Construction, parameter names and the state_dict must be exactly the reference's; the forward must reach the
HIP-backed quantizers (which refuse CPU tensors loudly -- there is no fallback to hide behind)."""
import importlib
import os
import sys
import warnings

import pytest
import torch

REF = os.environ.get("LLMQAT_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "models")), reason="reference checkout not present")


def _fresh_import(swap):
    for name in [m for m in sys.modules if m == "models" or m.startswith("models.")]:
        del sys.modules[name]
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    try:
        if swap:
            import llm_qat_amd.utils_quant as dropin
            importlib.import_module("models")            # the package itself
            sys.modules["models.utils_quant"] = dropin   # what INTEGRATION.md's re-export amounts to
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            cfg_mod = importlib.import_module("models.configuration_llama")
            mdl_mod = importlib.import_module("models.modeling_llama_quant")
        return cfg_mod, mdl_mod
    finally:
        sys.path.remove(REF)


def _build(cfg_mod, mdl_mod):
    cfg = cfg_mod.LlamaConfig(vocab_size=128, hidden_size=64, intermediate_size=176, num_hidden_layers=2, num_attention_heads=4,
                              max_position_embeddings=32, w_bits=4, a_bits=8, pad_token_id=0, bos_token_id=1, eos_token_id=2)
    cfg.kv_bits = 4
    cfg.use_cache = False
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return mdl_mod.LlamaForCausalLM(cfg)


def test_reference_model_builds_on_the_dropin_with_identical_parameters():
    ref_model = _build(*_fresh_import(swap=False))
    ref_keys = [(k, tuple(v.shape)) for k, v in ref_model.state_dict().items()]
    cfg_mod, mdl_mod = _fresh_import(swap=True)
    import llm_qat_amd.utils_quant as dropin
    assert mdl_mod.QuantizeLinear is dropin.QuantizeLinear and mdl_mod.SymQuantizer is dropin.SymQuantizer
    model = _build(cfg_mod, mdl_mod)
    assert [(k, tuple(v.shape)) for k, v in model.state_dict().items()] == ref_keys
    lin = model.model.layers[0].self_attn.q_proj
    assert isinstance(lin, dropin.QuantizeLinear) and (lin.w_bits, lin.a_bits) == (4, 8) and lin.act_quantizer is dropin.SymQuantizer
    assert model.model.layers[0].self_attn.act_quantizer_k is dropin.SymQuantizer        # KV hooks (:253-254)
    model.load_state_dict(ref_model.state_dict())                                         # checkpoints interchange
    # the forward reaches the HIP-backed quantizer and fails loudly on CPU tensors: no silent eager fallback
    with pytest.raises(RuntimeError, match="no CPU"):
        model(input_ids=torch.randint(2, 128, (1, 8)), use_cache=False)
    for name in [m for m in sys.modules if m == "models" or m.startswith("models.")]:
        del sys.modules[name]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("w_bits,a_bits,kv_bits", [(4, 8, 4), (8, 8, 8), (2, 8, 16), (1, 8, 32)])
def test_real_reference_model_on_the_dropin_cpu_path_equals_the_real_reference(w_bits, a_bits, kv_bits, dtype):
    """The REAL model code (models/modeling_llama_quant.py: 7 QuantizeLinear per layer, the KV hooks, gradient flow through all of it) run
    twice on CPU tensors, live: on the reference's own models/utils_quant.py, and on the drop-in with its opt-in CPU-tensor path
    (llm_qat_amd.allow_cpu_tensors: the product's own torch ops, with the shared activation fake-quant of the host logic active).
    Loss, logits and every parameter gradient bit-identical -- W4-A8-KV4, W8-A8-KV8, and the 1-/2-bit weight branches."""
    import llm_qat_amd

    def step(cfg_mod, mdl_mod, state=None):
        cfg = cfg_mod.LlamaConfig(vocab_size=128, hidden_size=64, intermediate_size=176, num_hidden_layers=2, num_attention_heads=4,
                                  max_position_embeddings=32, w_bits=w_bits, a_bits=a_bits, pad_token_id=0, bos_token_id=1, eos_token_id=2)
        cfg.kv_bits = kv_bits
        cfg.use_cache = False
        torch.manual_seed(0)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = mdl_mod.LlamaForCausalLM(cfg)
        if state is not None:
            model.load_state_dict(state)
        model = model.to(dtype)
        ids = torch.randint(2, 128, (2, 12), generator=torch.Generator().manual_seed(1))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = model(input_ids=ids, labels=ids, use_cache=False)
        out.loss.backward()
        return model, out.loss.detach(), out.logits.detach(), {n: p.grad for n, p in model.named_parameters()}

    ref_model, ref_loss, ref_logits, ref_grads = step(*_fresh_import(swap=False))
    state = {k: v.float() for k, v in ref_model.state_dict().items()}
    llm_qat_amd.allow_cpu_tensors(True)
    llm_qat_amd.stats(reset=True)
    try:
        cfg_mod, mdl_mod = _fresh_import(swap=True)
        _, loss, logits, grads = step(cfg_mod, mdl_mod, state)    # the reference model's own weights (exact in fp32 for either dtype)
        assert torch.equal(loss, ref_loss), (float(loss), float(ref_loss))
        assert torch.equal(logits, ref_logits)
        assert set(grads) == set(ref_grads)
        for n in ref_grads:
            assert (grads[n] is None) == (ref_grads[n] is None) and (grads[n] is None or torch.equal(grads[n], ref_grads[n])), n
        st = llm_qat_amd.stats()
        if 2 < a_bits < 32:
            assert st.get("act_share_hit", 0) >= 2 * 3, st     # q/k/v and gate/up found their input already fake-quantized, per layer
    finally:
        llm_qat_amd.allow_cpu_tensors(False)
        llm_qat_amd.reset_learned_state()
        for name in [m for m in sys.modules if m == "models" or m.startswith("models.")]:
            del sys.modules[name]


def test_random_programs_on_the_cpu_path_equal_the_live_reference():
    """tests/test_gpu_random_programs.py's generator (graphs of QuantizeLinear layers incl. the 1-/2-bit branches, hook-style SymQuantizer.apply
    calls, glue, no_grad regions, checkpointed steps, tensor hooks, a second backward), run on CPU tensors against the REAL reference's own
    classes, live: the drop-in's opt-in CPU-tensor path with its host logic at random settings.  Every output and gradient bit
    for bit, with the shared activation fake-quant off and on (every sibling has its own autograd node: utils_quant.py point 1)."""
    import random
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_gpu_random_programs as RP
    for name in [m for m in sys.modules if m == "models" or m.startswith("models.")]:
        del sys.modules[name]
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    try:
        import models.utils_quant as R
    finally:
        sys.path.remove(REF)
    R.REAL_REFERENCE = True
    n, seed0 = int(os.environ.get("LLMQAT_RANDOM_PROGRAMS", "80")), int(os.environ.get("LLMQAT_RANDOM_SEED0", "0"))
    llm_qat_amd.allow_cpu_tensors(True)
    try:
        for seed in range(seed0, seed0 + n):
            for share in (False, True):
                prog = RP.gen_program(random.Random(seed))
                cfg = prog[-1]
                llm_qat_amd.reset_learned_state()
                want_o, want_g = RP.run_program(R, True, prog, device="cpu")
                prev = llm_qat_amd.get_backward_mode()
                try:
                    llm_qat_amd.conservative(cfg["conservative"])
                    if not cfg["conservative"]:
                        llm_qat_amd.share_activation_quant(share)
                        llm_qat_amd.enable_weight_quant_cache(cfg["weight_cache"] is not None, persistent=cfg["weight_cache"] == "persistent")
                    llm_qat_amd.set_backward_mode(cfg["backward_mode"])
                    llm_qat_amd.cpp_node(cfg["cpp_node"])
                    llm_qat_amd.reset_learned_state()
                    got_o, got_g = RP.run_program(UQ, False, prog, device="cpu")
                finally:
                    llm_qat_amd.cpp_node(True)
                    llm_qat_amd.set_backward_mode(prev)
                    llm_qat_amd.conservative(False)
                    llm_qat_amd.enable_weight_quant_cache(False)
                tag = f"program seed={seed} share={share}: {prog}"
                assert len(want_o) == len(got_o)
                for i, (a, b) in enumerate(zip(want_o, got_o)):
                    assert RP.eq(a, b), f"output {i} of {tag}"
                for i, (a, b) in enumerate(zip(want_g, got_g)):
                    assert (a is None) == (b is None), f"gradient {i} present in one run only, {tag}"
                    if a is None:
                        continue
                    assert RP.eq(a, b), f"gradient {i} of {tag}"
    finally:
        llm_qat_amd.allow_cpu_tensors(False)
        llm_qat_amd.reset_learned_state()
        for name in [m for m in sys.modules if m == "models" or m.startswith("models.")]:
            del sys.modules[name]
