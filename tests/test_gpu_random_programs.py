"""GPU tier: RANDOM PROGRAMS over the drop-in -- a generator builds a small graph out of QuantizeLinear layers (random w_bits incl. 1 / 2 /
32, a_bits incl. off, Sym / Asym, layerwise flags), KV-hook style SymQuantizer.apply calls, elementwise glue, no_grad regions, tensors that
do or do not require grad, widths that the vector kernels serve and widths they do not, some steps under activation checkpointing, then
backpropagates a random loss -- and runs it twice, the drop-in under a random configuration (backward mode, conservative, weight cache,
K/V pairing, in-place weight gradients, the C++ or the Python autograd node: none of which may change a result): on the live eager chain (tiny_llama.EagerQuant over
oracle/eager_chain.py, plus the 1-/2-bit branch's op chain) and on the drop-in at default settings.

What must hold for every program: the same outputs bit for bit; the same set of tensors receiving a gradient; every gradient bit for bit,
with the shared activation fake-quant off AND on (rounds 1-4 compared "up to the association order of bf16 sums" with it on; since round 5
every sibling projection has an autograd node of its own, the reference's graph, and 2 500 programs per mode are bit-identical).
"""
import os
import random
import sys

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tiny_llama as TL  # noqa: E402

PROGRAMS = int(os.environ.get("LLMQAT_RANDOM_PROGRAMS", "300"))
SEED0 = int(os.environ.get("LLMQAT_RANDOM_SEED0", "0"))


class EagerLowBitLinear(nn.Linear):
    """the reference's QuantizeLinear for w_bits 1 / 2 (utils_quant.py:202-242), op for op, activations through the eager chain"""

    def __init__(self, i, o, w_bits, a_bits, symmetric, act_layerwise, weight_layerwise, E):
        super().__init__(i, o, bias=False)
        self.w_bits, self.a_bits, self.alw, self.wlw = w_bits, a_bits, act_layerwise, weight_layerwise
        self.act_q = (E.EagerSym if symmetric else E.EagerAsym) if 2 < a_bits < 32 else None

    def forward(self, x):
        from test_gpu_features import eager_low_bit
        w = eager_low_bit(self.weight, self.w_bits, self.wlw)
        if self.act_q is not None:
            x = self.act_q.apply(x, torch.tensor([-2.0, 2.0]), self.a_bits, self.alw)
        return F.linear(x, w)


DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}


def make_module(Q, eager, cfg, D, device="cuda", dtype=torch.bfloat16):
    w_bits, a_bits, sym, alw, wlw, seed = cfg
    if eager and w_bits < 3 and not hasattr(Q, "REAL_REFERENCE"):   # (tiny_llama.EagerQuant has no 1-/2-bit branch; the real reference has)
        from oracle import eager_chain as E
        m = EagerLowBitLinear(D, D, w_bits, a_bits, sym, alw, wlw, E)
    elif eager:
        m = Q.QuantizeLinear(D, D, symmetric=sym, w_bits=w_bits, a_bits=a_bits, act_layerwise=alw, weight_layerwise=wlw)
    else:
        m = Q.QuantizeLinear(D, D, symmetric=sym, w_bits=w_bits, a_bits=a_bits, act_layerwise=alw, weight_layerwise=wlw)
    m = m.to(device).to(dtype)
    with torch.no_grad():
        m.weight.copy_((torch.randn(D, D, generator=torch.Generator().manual_seed(seed)) * 0.3).to(device).to(dtype))
        m.weight[1, 2] = 2.5
    return m


def gen_program(rng):
    mods = [(rng.choice([1, 2, 4, 4, 8, 32]), rng.choice([4, 8, 8, 8, 32]), rng.random() < 0.85, rng.random() < 0.1, rng.random() < 0.1, rng.randrange(10 ** 6))
            for _ in range(rng.randint(2, 6))]
    inputs = [(rng.random() < 0.8, rng.randrange(10 ** 6)) for _ in range(rng.randint(1, 2))]
    steps, n_t = [], len(inputs)
    for _ in range(rng.randint(3, 12)):
        kind = rng.choices(["linear", "hook", "scale", "add", "nograd_linear", "kv", "hook_any", "detach", "view4"], [6, 3, 1, 1, 1, 1, 2, 0.5, 1])[0]
        if kind in ("linear", "nograd_linear"):
            steps.append((kind, rng.randrange(len(mods)), rng.randrange(n_t)))
        elif kind == "hook":
            steps.append((kind, rng.choice([1, 2, 4, 8]), rng.randrange(n_t), rng.choice([(-2.0, 2.0), (-2.0, 2.0), (-1.0, 1.5)])))
        elif kind == "scale":
            steps.append((kind, rng.choice([0.5, 1.5, -1.0]), rng.randrange(n_t)))
        elif kind == "hook_any":   # either quantizer, any bit width, row-wise or layerwise, possibly under no_grad
            steps.append((kind, rng.choice(["sym", "asym"]), rng.choice([1, 2, 3, 4, 8, 16]), rng.randrange(n_t), rng.random() < 0.3, rng.random() < 0.15))
        elif kind == "detach":
            steps.append((kind, rng.randrange(n_t)))
        elif kind == "view4":      # a 4-D view of a 3-D tensor, fake-quantized with the 4-D granularity (:60-68), viewed back
            steps.append((kind, rng.choice([4, 8]), rng.randrange(n_t)))
        elif kind == "kv":    # the explicit two-tensor call; appends TWO tensors
            a = rng.randrange(n_t)
            b = rng.randrange(n_t)
            steps.append((kind, rng.choice([2, 4, 8]), a, b, rng.random() < 0.8))   # (a == b happens: one tensor as both K and V)
            n_t += 1
        else:
            steps.append((kind, rng.randrange(n_t), rng.randrange(n_t)))
        n_t += 1
    loss = [(i, rng.choice([1.0, 2.0, -0.5])) for i in range(len(inputs), n_t) if rng.random() < 0.6] or [(n_t - 1, 1.0)]
    # how the drop-in is configured for this program (results may not depend on any of it) + shape of the data
    settings = dict(width=rng.choice([64, 64, 100, 264]), three_d=rng.random() < 0.7, backward_mode=rng.choice(["mask", "mask", "bounds", "plain"]),
                    conservative=rng.random() < 0.15, weight_cache=rng.choice([None, None, "step", "persistent"]),
                    checkpoint=rng.choice([None, None, "reentrant", "nonreentrant"]), pair_kv=rng.random() < 0.85, inplace=rng.random() < 0.85,
                    second_backward=rng.random() < 0.2, grad_hooks=rng.random() < 0.3, dtype=rng.choice(["bf16", "bf16", "fp16", "fp32"]),
                    autocast_dtype=rng.choice(["bf16", "bf16", "fp16"]))
    ac = rng.random() < 0.5
    settings["cpp_node"] = rng.random() < 0.6      # which autograd node QuantizeLinear's operand pair builds (drawn last: older seeds keep their programs)
    return mods, inputs, steps, loss, ac, settings


def run_program(Q, eager, prog, device="cuda"):
    from torch.utils.checkpoint import checkpoint
    mods_cfg, inputs, steps, loss, autocast, cfg = prog
    D = cfg["width"]
    dt = DT[cfg["dtype"]]
    mods = [make_module(Q, eager, c, D, device, dt) for c in mods_cfg]
    shape = (2, 7, D) if cfg["three_d"] else (11, D)
    ts = [(torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * 1.5).to(device).to(dt).requires_grad_(g) for g, seed in inputs]
    n_in = len(ts)
    with torch.autocast("cuda", dtype=DT[cfg["autocast_dtype"]], enabled=autocast and device == "cuda"):
        for k, st in enumerate(steps):
            if st[0] == "linear":
                x = ts[st[2]] if ts[st[2]].dtype == dt or (autocast and device == "cuda") else ts[st[2]].to(dt)
                if cfg["checkpoint"] and k % 3 == 0 and torch.is_grad_enabled():   # (both runs checkpoint the same steps)
                    ts.append(checkpoint(mods[st[1]], x, use_reentrant=cfg["checkpoint"] == "reentrant"))
                else:
                    ts.append(mods[st[1]](x))
            elif st[0] == "nograd_linear":
                with torch.no_grad():
                    ts.append(mods[st[1]](ts[st[2]] if ts[st[2]].dtype == dt or (autocast and device == "cuda") else ts[st[2]].to(dt)))
            elif st[0] == "hook":
                ts.append(Q.SymQuantizer.apply(ts[st[2]], torch.tensor(st[3]), st[1], False))
            elif st[0] == "scale":
                ts.append(ts[st[2]] * st[1])
            elif st[0] == "hook_any":
                q = Q.SymQuantizer if st[1] == "sym" else Q.AsymQuantizer
                with torch.set_grad_enabled(torch.is_grad_enabled() and not st[5]):
                    ts.append(q.apply(ts[st[3]], torch.tensor([-2.0, 2.0]), st[2], st[4]))
            elif st[0] == "detach":
                ts.append(ts[st[1]].detach())
            elif st[0] == "view4":
                t = ts[st[2]]
                if t.dim() == 3 and t.shape[-1] % 4 == 0 and t.is_contiguous():
                    y = Q.SymQuantizer.apply(t.view(t.shape[0], t.shape[1], 4, t.shape[2] // 4), torch.tensor([-2.0, 2.0]), st[1], False)
                    ts.append(y.reshape(t.shape))
                else:
                    ts.append(t * 1.0)
            elif st[0] == "kv":
                k, v = ts[st[2]], ts[st[3]]
                ck, cv = torch.tensor([-2.0, 2.0]), torch.tensor([-2.0, 2.0] if st[4] else [-1.0, 1.5])
                if getattr(Q, "quantize_kv", None) is not None and not eager:
                    kq, vq = Q.quantize_kv(k, v, ck, cv, st[1])
                else:
                    kq, vq = Q.SymQuantizer.apply(k, ck, st[1], False), Q.SymQuantizer.apply(v, cv, st[1], False)
                ts += [kq, vq]
            else:
                ts.append(ts[st[1]].float() + ts[st[2]].float())
    seen = [None] * len(ts)
    if cfg["grad_hooks"]:     # tensor hooks see the gradient of an intermediate exactly as in the reference (recorded per tensor)
        for i, t in enumerate(ts):
            if i >= n_in and t.requires_grad:
                def hook(g, i=i):
                    seen[i] = g.detach().clone() if seen[i] is None else seen[i] + g.detach()
                t.register_hook(hook)
    total = sum(ts[i].float().sum() * w for i, w in loss)
    if total.requires_grad:
        total.backward(retain_graph=cfg["second_backward"])
        if cfg["second_backward"] and not cfg["checkpoint"]:
            total.backward()
    outs = [t.detach() for t in ts[n_in:]]
    grads = [t.grad for t in ts[:n_in]] + [m.weight.grad for m in mods] + seen
    return outs, grads


def eq(a, b):
    """bit-equal up to the payload of NaNs (fp16 programs overflow now and then: both runs then hold NaN at the same places)"""
    return a.dtype == b.dtype and a.shape == b.shape and torch.equal(a.isnan(), b.isnan()) and torch.equal(a.nan_to_num(), b.nan_to_num())


def check_program(seed, share):
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    prog = gen_program(random.Random(seed))
    cfg = prog[-1]
    llm_qat_amd.reset_learned_state()
    want_o, want_g = run_program(TL.EagerQuant(), True, prog)
    prev_mode = llm_qat_amd.get_backward_mode()
    try:
        llm_qat_amd.conservative(cfg["conservative"])
        if not cfg["conservative"]:
            llm_qat_amd.share_activation_quant(share)
            llm_qat_amd.pair_kv_hooks(cfg["pair_kv"])
            llm_qat_amd.inplace_weight_grad(cfg["inplace"])
            llm_qat_amd.enable_weight_quant_cache(cfg["weight_cache"] is not None, persistent=cfg["weight_cache"] == "persistent")
        llm_qat_amd.set_backward_mode(cfg["backward_mode"])
        llm_qat_amd.cpp_node(cfg["cpp_node"])
        llm_qat_amd.reset_learned_state()
        got_o, got_g = run_program(UQ, False, prog)
    finally:
        llm_qat_amd.cpp_node(True)
        llm_qat_amd.set_backward_mode(prev_mode)
        llm_qat_amd.conservative(False)
        llm_qat_amd.enable_weight_quant_cache(False)
    tag = f"program seed={seed} share={share}: {prog}"
    assert len(want_o) == len(got_o)
    for i, (a, b) in enumerate(zip(want_o, got_o)):
        assert eq(a, b), f"output {i} of {tag}"
    for i, (a, b) in enumerate(zip(want_g, got_g)):
        assert (a is None) == (b is None), f"gradient {i} present in one run only, {tag}"
        if a is None:
            continue
        assert eq(a, b), f"gradient {i} of {tag}"   # sharing on or off (rounds 1-4: "close" with it on; since round 5 every sibling has its own node)


@pytest.mark.parametrize("share", [False, True])
def test_random_programs_match_the_eager_chain(share):
    import llm_qat_amd
    llm_qat_amd.set_semantics("device_eager")
    try:
        progress = os.environ.get("LLMQAT_PROGRESS_FILE")    # long sweeps: a line every 500 programs, so that a watchdog sees the run alive
        for seed in range(SEED0, SEED0 + PROGRAMS):
            check_program(seed, share)
            if progress and (seed + 1) % 500 == 0:
                with open(progress, "a") as f:
                    f.write(f"share={share} seed {seed + 1}\n")
    finally:
        llm_qat_amd.share_activation_quant(True)
        llm_qat_amd.set_semantics("cpu_eager")
        llm_qat_amd.reset_learned_state()
