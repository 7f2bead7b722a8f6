"""GPU tier: a block of QuantizeLinear layers + the KV hooks, forward AND backward, captured into ONE HIP graph at the product's default
settings (operand pairing, shared activation fake-quant, K/V speculation at the unchanged hooks, weight gradients masked in place) and
replayed with fresh data: every output and gradient bit-identical to the same block run eagerly.

Why it matters: the small launches of a layer (`[tokens, 4096]`-sized, 6 us) and all of BASELINE configs[0] are host-bound;
a captured step has no host cost at all.  The library allocates only through PyTorch's graph-aware allocator, never synchronises, reads
its clip values from CPU tensors and decides everything stateful from the call sequence alone, so the capture holds exactly the launches
the eager step makes.  (The reference MODEL is not capturable as written -- its attention builds a device scalar from a Python float,
modeling_llama_quant.py:72 -- which is why this block has the projections, the hooks and the MLP but no softmax.)
"""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


class Block(nn.Module):
    """q/k/v projections on one input (shared activation fake-quant), K and V through the unchanged hooks (modeling_llama_quant.py:320-327),
    o_proj, then gate/up/down (:235)"""

    def __init__(self, quant, d, m, w_bits=4, a_bits=8, kv_bits=4):
        super().__init__()
        mk = lambda i, o: quant.QuantizeLinear(i, o, bias=False, w_bits=w_bits, a_bits=a_bits)  # noqa: E731
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = mk(d, d), mk(d, d), mk(d, d), mk(d, d)
        self.gate_proj, self.up_proj, self.down_proj = mk(d, m), mk(d, m), mk(m, d)
        self.quant, self.kv_bits, self.clip = quant, kv_bits, torch.tensor([-2.0, 2.0])

    def forward(self, h):
        q = self.q_proj(h)
        k = self.k_proj(h)
        v = self.v_proj(h)
        k = self.quant.SymQuantizer.apply(k, self.clip, self.kv_bits, False)
        v = self.quant.SymQuantizer.apply(v, self.clip, self.kv_bits, False)
        h = h + self.o_proj(torch.tanh(q) * k.to(q.dtype) + v.to(q.dtype))
        return h + self.down_proj(F.silu(self.gate_proj(h)) * self.up_proj(h))


def _step(block, x, go, autocast):
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        out = block(x)
    out.backward(go)
    return out


@pytest.mark.parametrize("autocast", [True, False])
def test_block_step_captured_into_one_graph_matches_eager(autocast):
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    torch.manual_seed(11)
    d, m, tokens = 512, 1408, 192
    llm_qat_amd.reset_learned_state()
    block = Block(UQ, d, m).cuda().bfloat16()
    with torch.no_grad():
        for p in block.parameters():
            p.mul_(0.6)
            p[1, 2] = 2.5     # beyond the STE clip: that weight row takes the masked path of the in-place backward
    x = torch.randn(2, tokens // 2, d, device="cuda").bfloat16().requires_grad_(True)
    go = (torch.randn(2, tokens // 2, d, device="cuda") * 1e-2).bfloat16()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):      # warm-up outside the capture (allocator, lazy inits), as torch's whole-network capture recipe has it
        for _ in range(3):
            block.zero_grad(set_to_none=True)
            x.grad = None
            _step(block, x, go, autocast)
    torch.cuda.current_stream().wait_stream(s)
    block.zero_grad(set_to_none=True)   # the captured backward allocates the .grad tensors from the graph's pool
    x.grad = None
    llm_qat_amd.stats(reset=True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = _step(block, x, go, autocast)
    st = llm_qat_amd.stats()
    # the capture ran the default data flow: 4 paired launches (q, o, gate, down), K+V in one, every weight gradient in place
    assert st.get("pair_launch") == 4 and st.get("act_share_hit") == 3 and st.get("kv_pair_hit") == 1 and st.get("inplace_taken") == 7, st
    for trial in range(3):
        with torch.no_grad():
            x.copy_((torch.randn_like(x, dtype=torch.float32) * (0.5 + trial)).bfloat16())
            go.copy_((torch.randn_like(go, dtype=torch.float32) * 1e-2).bfloat16())
        graph.replay()
        torch.cuda.synchronize()
        got = (out.clone(), x.grad.clone(), {n: p.grad.clone() for n, p in block.named_parameters()})
        # the same step, eagerly, on copies
        ref = Block(UQ, d, m).cuda().bfloat16()
        ref.load_state_dict(block.state_dict())
        xr = x.detach().clone().requires_grad_(True)
        want_out = _step(ref, xr, go, autocast)
        assert torch.equal(got[0], want_out), f"replay {trial}: output"
        assert torch.equal(got[1], xr.grad), f"replay {trial}: input gradient"
        for n, p in ref.named_parameters():
            assert torch.equal(got[2][n], p.grad), f"replay {trial}: gradient of {n}"
            assert got[2][n][1, 2] == 0, f"replay {trial}: STE mask on {n}"
    llm_qat_amd.reset_learned_state()


@pytest.mark.parametrize("autocast", [False, True])
def test_make_graphed_callables_on_a_block_matches_eager(autocast):
    """PyTorch's own partial-network capture (torch.cuda.make_graphed_callables: forward and backward of one callable in two graphs, the rest of
    the step eager) on a fake-quantized block: nothing in the host logic needs the host during a step -- no read-back, no data-dependent
    launch -- so the block is graphable as it is, and replays give the eager block's bits on fresh data.  (Small models are host-bound:
    tools/graph_block_bench.py measures 3.3x for configs[0]'s widths.)"""
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    torch.manual_seed(12)
    d, m, tokens = 256, 688, 64
    llm_qat_amd.reset_learned_state()
    block = Block(UQ, d, m).cuda().bfloat16()
    with torch.no_grad():
        for p in block.parameters():
            p.mul_(0.6)
            p[1, 2] = 2.5
    ref = Block(UQ, d, m).cuda().bfloat16()
    ref.load_state_dict(block.state_dict())
    sample = torch.randn(2, tokens // 2, d, device="cuda").bfloat16().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast, cache_enabled=False):
        graphed = torch.cuda.make_graphed_callables(block, (sample,))
    for trial in range(3):
        x = (torch.randn(2, tokens // 2, d, device="cuda") * (0.5 + trial)).bfloat16()
        go = (torch.randn(2, tokens // 2, d, device="cuda") * 1e-2).bfloat16()
        xs = [x.clone().requires_grad_(True) for _ in range(2)]
        outs = []
        for mod, xi in ((graphed, xs[0]), (ref, xs[1])):
            for p in (block if mod is graphed else ref).parameters():
                p.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast, cache_enabled=False):
                out = mod(xi)
            out.backward(go)
            outs.append(out.detach().clone())
        assert torch.equal(outs[0], outs[1]), f"replay {trial}: output"
        assert torch.equal(xs[0].grad, xs[1].grad), f"replay {trial}: input gradient"
        for (n, p), (_, q) in zip(block.named_parameters(), ref.named_parameters()):
            assert torch.equal(p.grad, q.grad), f"replay {trial}: gradient of {n}"
            assert p.grad[1, 2] == 0, f"replay {trial}: STE mask on {n}"
    llm_qat_amd.reset_learned_state()
