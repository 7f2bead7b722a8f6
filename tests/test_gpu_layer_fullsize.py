"""GPU tier: ONE decoder layer at the real model dimensions of BASELINE configs 1-3 (LLaMA-7B: hidden 4096, MLP 11008, 32 heads,
seq 2048, bs 1) and config 5 (13B: 5120 / 13824 / 40 heads), as LLM-QAT runs it -- bf16 weights, torch.autocast("cuda", bf16),
activation checkpointing (run_train.sh:17-18,:36) -- through three implementations of the reference's call sites
(models/modeling_llama_quant.py:313-327 attention incl. the KV hooks, :235 MLP, :732-747 checkpointing):

    eager chain                 the reference's ATen op chain (oracle/eager_chain.py), live on this GPU
    drop-in, default settings   operand pairing + shared activation quant + K/V speculation + in-place weight gradients, all at once,
                                at the launch shapes the metric is quoted on (row_reg_kernel<..., 512, 3>, 2-chunk backward)
    drop-in, conservative(True) one launch and one autograd node per reference call

Bar: the layer's output, the input gradient and EVERY parameter gradient bit-identical across the three.  While at it: the number of
fq_* launches per step (so a silent fall-back to unpaired calls shows) and the llm_qat_amd.stats() counters of the stateful host logic.
"""
import os
import sys

import pytest
import torch
from torch.utils.checkpoint import checkpoint

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tiny_llama as TL  # noqa: E402

DIMS = {"7b": dict(hidden_size=4096, intermediate_size=11008, num_attention_heads=32),
        "13b": dict(hidden_size=5120, intermediate_size=13824, num_attention_heads=40)}
LAUNCHING = ("fq_sym_fwd", "fq_asym_fwd", "fq_sym_fwd_train", "fq_asym_fwd_train", "fq_sym_fwd_autocast", "fq_sym_fwd_pair", "fq_sym_fwd_multi",
             "fq_ste_bwd", "fq_ste_bwd_rows", "fq_ste_bwd_mask", "fq_ste_bwd_mask_pair", "fq_ste_bwd_mask_multi", "fq_ste_bwd_mask_wide", "fq_w12_fwd",
             "fq_w12_fwd_rows", "fq_rowwise_fwd_v", "fq_sym_fwd_multi_v", "fq_ste_bwd_mask_multi_v", "fq_ste_bwd_v")


class LaunchCounter:
    """counts calls of the C ABI's launching entry points (instance attributes shadow the CDLL's lazily bound functions)"""

    def __init__(self):
        from llm_qat_amd import _lib
        self.L, self.n = _lib.lib(), {}

    @staticmethod
    def _cpp():
        """launches the C++ autograd node made itself (it calls the same two entry points through their addresses)"""
        from llm_qat_amd import utils_quant as U
        c = U._cnode.counters() if U._cnode is not None else {}
        return {"fq_sym_fwd_pair": c.get("cpp_pair_forward", 0), "fq_ste_bwd_mask_pair": c.get("cpp_pair_backward", 0),
                "fq_sym_fwd_multi": c.get("cpp_weight_forward", 0), "fq_ste_bwd_mask": c.get("cpp_one_backward", 0),
                "fq_ste_bwd_mask_wide": c.get("cpp_one_backward_wide", 0)}

    def __enter__(self):
        self.orig = {}
        self.cpp0 = self._cpp()
        for name in LAUNCHING:
            f = getattr(self.L, name)
            self.orig[name] = f

            def g(*a, _f=f, _n=name):
                self.n[_n] = self.n.get(_n, 0) + 1
                return _f(*a)
            setattr(self.L, name, g)
        return self

    def __exit__(self, *exc):
        for name, f in self.orig.items():
            setattr(self.L, name, f)
        for name, v in self._cpp().items():
            if v - self.cpp0[name]:
                self.n[name] = self.n.get(name, 0) + v - self.cpp0[name]

    @property
    def forward(self):
        return sum(v for k, v in self.n.items() if "fwd" in k)

    @property
    def backward(self):
        return sum(v for k, v in self.n.items() if "bwd" in k)


def build_layer(quant, dims, w_bits, a_bits, kv_bits, state=None):
    cfg = dict(TL.TINY, max_position_embeddings=2048, **DIMS[dims])
    layer = TL.Layer(cfg, quant, w_bits, a_bits, kv_bits).to(device="cuda", dtype=torch.bfloat16)
    if state is None:
        g = torch.Generator(device="cuda").manual_seed(1234)
        with torch.no_grad():
            for name, p in layer.named_parameters():
                if "layernorm" in name:
                    p.copy_(1.0 + 0.05 * torch.randn(p.shape, generator=g, device="cuda"))
                else:
                    p.copy_(torch.randn(p.shape, generator=g, device="cuda") * 0.02)
                    # a few weights at / beyond the STE clip: their rows take the masked path of the (in-place) weight backward
                    p[3, 5], p[4, 6], p[p.shape[0] - 1, p.shape[1] - 1] = 2.5, -2.0, 2.25
    else:
        layer.load_state_dict(state)
    return layer


def run_step(layer, h0, go, ckpt):
    for p in layer.parameters():
        p.grad = None
    h = h0.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = layer(h) if ckpt is None else checkpoint(layer, h, use_reentrant=(ckpt == "reentrant"))
    out.backward(go)
    return out.detach(), h.grad.detach(), {n: p.grad.detach() for n, p in layer.named_parameters()}


CONFIGS = [("7b", 4, 8, 4), ("7b", 4, 8, 8), ("7b", 8, 8, 8), ("13b", 4, 8, 4)]   # BASELINE configs 1, 2, 3 and 5's per-GPU work


@pytest.mark.parametrize("ckpt", ["reentrant", "nonreentrant"])
@pytest.mark.parametrize("dims,w_bits,a_bits,kv_bits", CONFIGS)
def test_full_size_layer_is_bit_identical_to_the_eager_chain(dims, w_bits, a_bits, kv_bits, ckpt):
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    hidden = DIMS[dims]["hidden_size"]
    g = torch.Generator(device="cuda").manual_seed(7)
    h0 = torch.randn(1, 2048, hidden, generator=g, device="cuda").bfloat16()
    h0[0, 5, 17], h0[0, 9, 3] = 30.0, -25.0          # outlier channels, as activations have
    go = (torch.randn(1, 2048, hidden, generator=g, device="cuda") * 1e-2).bfloat16()
    llm_qat_amd.set_semantics("device_eager")      # the live ATen chain is device-eager by definition
    llm_qat_amd.reset_learned_state()
    try:
        ref_layer = build_layer(TL.EagerQuant(), dims, w_bits, a_bits, kv_bits)
        state = ref_layer.state_dict()
        want = run_step(ref_layer, h0, go, ckpt)
        del ref_layer
        results = {}
        for mode in ("default", "conservative"):
            llm_qat_amd.conservative(mode == "conservative")
            layer = build_layer(UQ, dims, w_bits, a_bits, kv_bits, state)
            run_step(layer, h0, go, ckpt)           # warm-up step (allocator, first-use paths)
            llm_qat_amd.stats(reset=True)
            with LaunchCounter() as lc:
                got = run_step(layer, h0, go, ckpt)
            results[mode] = (got, lc, llm_qat_amd.stats())
            del layer
        for mode, (got, lc, st) in results.items():
            tag = f"{dims} W{w_bits}A{a_bits}KV{kv_bits} {ckpt} {mode}"
            assert got[0].dtype == want[0].dtype and torch.equal(got[0], want[0]), tag + ": layer output"
            assert torch.equal(got[1], want[1]), tag + ": input gradient"
            for n in want[2]:
                assert torch.equal(got[2][n], want[2][n]), f"{tag}: gradient of {n}"
                if n.endswith("proj.weight"):
                    assert got[2][n][3, 5] == 0 and got[2][n][4, 6] == 0, f"{tag}: STE mask on {n}"
        # launches per step.  Forward passes per step: 2 (the checkpointed pass + its recompute).
        (_, lc, st), (_, lcc, stc) = results["default"], results["conservative"]
        # reference structure: 7 weights + 7 inputs + K + V = 16 quantizer calls per forward, 16 STE backwards
        assert lcc.forward == 2 * 16 and lcc.backward == 16, (lcc.n,)
        assert not any(k.startswith(("pair_", "kv_pair", "act_share", "inplace_taken")) for k in stc), stc
        # default: q / o / gate / down pair their weight with their input (4), k / v / up find the input shared and quantize the weight
        # alone (3), K + V hooks are one launch (1): 8 per forward; backward one launch per autograd node: every module has ONE node over its
        # weight and its (own or shared) activation data = 7 two-tensor launches (round 5: each sibling masks its own input gradient, as in
        # the reference's graph; rounds 1-4: 4 pairs + 3 weight-only launches behind one shared activation node), + K + V
        # (the speculated K and V share a forward launch but not a node: a V nobody asks for must leave no trace in the graph) = 9
        assert lc.forward == 2 * 8 and lc.backward == 9, (lc.n,)
        assert lc.n.get("fq_sym_fwd_pair") == 2 * 5 and lc.n.get("fq_ste_bwd_mask_pair", 0) == 7 and lc.n.get("fq_ste_bwd_mask_wide", 0) == 2, lc.n
        assert st.get("pair_launch") == 2 * 4 and st.get("single_launch") == 2 * 3 and st.get("act_share_hit") == 2 * 3, st
        assert st.get("kv_pair_launch") == 2 and st.get("kv_pair_hit") == 2 and not st.get("kv_pair_discarded"), st
        assert st.get("inplace_taken") == 7 and not any(k.startswith("inplace_refused") for k in st), st   # all seven weight gradients by reference
    finally:
        llm_qat_amd.conservative(False)
        llm_qat_amd.set_semantics("cpu_eager")
        llm_qat_amd.reset_learned_state()
        torch.cuda.empty_cache()
