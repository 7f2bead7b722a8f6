#!/usr/bin/env python3
"""Generate the committed golden vectors for the fake-quant hot path.

Runs ONLY in the build container: it imports the real reference module
(`/root/reference/models/utils_quant.py`, SymQuantizer :31-87, AsymQuantizer :90-162,
QuantizeLinear :165-254) on CPU and records inputs + the outputs the reference itself
produced.  Nothing from the reference travels: the fixtures are data only
(inputs, expected outputs), stored as .npz with allow_pickle=False semantics.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Storage conventions
  * fp32 tensors  -> float32 arrays (bit patterns preserved, NaN payloads included)
  * bf16 / fp16   -> uint16 arrays holding the raw bit patterns (`<name>` + dtype in manifest)
  * bin indices   -> int32 (`idx`); taken from the reference's own op sequence
                     (`torch.round(input * s)` for Sym, `torch.round(input_normalized * s)` for
                     Asym), and the script asserts that finishing that sequence reproduces the
                     output of `Quantizer.apply` bit for bit, so idx is pinned to the reference.
  * `manifest` (JSON string inside each .npz) lists every case and its parameters.
"""
import json
import os
import sys

sys.dont_write_bytecode = True
REF = os.environ.get("LLMQAT_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from models.utils_quant import AsymQuantizer, QuantizeLinear, SymQuantizer  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
DT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}
CLIP = torch.tensor([-2.0, 2.0])


def to_np(t):
    """torch tensor -> numpy, 16-bit floats as raw uint16 bit patterns."""
    t = t.detach().contiguous()
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.view(torch.int16).numpy().view(np.uint16).copy()
    return t.numpy().copy()


def bits_equal(a, b):
    a, b = to_np(a), to_np(b)
    if a.dtype == np.float32:
        a, b = a.view(np.uint32), b.view(np.uint32)
    return a.shape == b.shape and bool((a == b).all())


# --------------------------------------------------------------------------------------
# intermediates, restated with the reference's exact op sequence (utils_quant.py:50-72)
# only to expose idx / scale; the final y always comes from Quantizer.apply itself.
# --------------------------------------------------------------------------------------
def sym_parts(x, bits, layerwise):
    if layerwise:
        max_input = torch.max(torch.abs(x)).expand_as(x)
    elif x.ndimension() <= 3:
        max_input = torch.max(torch.abs(x), dim=-1, keepdim=True)[0].expand_as(x)
    else:
        tmp = x.view(x.shape[0], x.shape[1], -1)
        max_input = torch.max(torch.abs(tmp), dim=-1, keepdim=True)[0].unsqueeze(-1).expand_as(x)
    s = (2 ** (bits - 1) - 1) / (max_input + 1e-6)
    idx = torch.round(x * s)
    y = idx.div(s + 1e-6)
    return idx, s, y


def asym_parts(x, bits, layerwise):
    if layerwise:
        alpha = x.max() - x.min()
        beta = x.min()
    elif x.ndimension() <= 3:
        alpha = (x.max(dim=-1, keepdim=True)[0] - x.min(dim=-1, keepdim=True)[0]).expand_as(x)
        beta = x.min(dim=-1, keepdim=True)[0].expand_as(x)
    else:
        tmp = x.view(x.shape[0], x.shape[1], -1)
        alpha = (tmp.max(dim=-1, keepdim=True)[0].unsqueeze(-1) - tmp.min(dim=-1, keepdim=True)[0].unsqueeze(-1)).expand_as(x)
        beta = tmp.min(dim=-1, keepdim=True)[0].unsqueeze(-1).expand_as(x)
    n = (x - beta) / (alpha + 1e-8)
    s = 2**bits - 1
    idx = torch.round(n * s)
    y = idx.div(s) * (alpha + 1e-8) + beta
    return idx, alpha, beta, y


def idx_to_i32(idx):
    """bin index tensor (float dtype) -> int32; NaN -> INT32_MIN, +-Inf -> +-(2^31-1)."""
    f = idx.float()
    out = torch.zeros(f.shape, dtype=torch.int32)
    fin = torch.isfinite(f)
    out[fin] = f[fin].clamp(-2.0e9, 2.0e9).to(torch.int32)
    out[torch.isnan(f)] = -(2**31)
    out[f == float("inf")] = 2**31 - 1
    out[f == float("-inf")] = -(2**31 - 1)
    return out.numpy()


# --------------------------------------------------------------------------------------
# input builders
# --------------------------------------------------------------------------------------
def rand_rows(gen, shape, dtype, scales=None):
    x = torch.randn(shape, generator=gen, dtype=torch.float32)
    rows = x.reshape(-1, shape[-1])
    if scales is None:
        scales = [1e-5, 1e-3, 0.02, 0.3, 1.0, 3.0, 30.0]
    for r in range(rows.shape[0]):
        rows[r] *= scales[r % len(scales)]
    return x.to(dtype)


def adversarial(dtype, cols=40):
    """One row per nasty situation."""
    f = torch.float32
    rows = []
    names = []

    def add(name, v):
        v = torch.as_tensor(v, dtype=f).flatten()
        row = torch.zeros(cols, dtype=f)
        row[: min(cols, v.numel())] = v[:cols]
        rows.append(row)
        names.append(name)

    g = torch.Generator().manual_seed(77)
    add("all_zero", [0.0])
    add("neg_zero", [-0.0, 0.0, -0.0])
    add("absmax_1e-7", torch.randn(cols, generator=g) * 1e-7)
    add("absmax_1e-6", torch.randn(cols, generator=g) * 1e-6)
    add("absmax_3e-5", torch.randn(cols, generator=g) * 3e-5)
    add("exact_pm2", [2.0, -2.0, 1.9921875, -1.9921875, 2.015625, -2.015625, 0.5, -0.5])
    # max = 7 -> s ~ 1 for 4 bit: values on .5 ties
    add("ties_q7", [7.0, 0.5, 1.5, 2.5, 3.5, 4.5, 5.5, 6.5, -0.5, -1.5, -2.5, -3.5, -4.5, -5.5, -6.5, -7.0])
    add("ties_q127", [127.0] + [k + 0.5 for k in range(0, 39)])
    add("ties_q3", [3.0, 0.5, 1.5, 2.5, -0.5, -1.5, -2.5])
    add("pos_inf", [float("inf"), 1.0, -2.0, 0.0, 3.0])
    add("neg_inf", [float("-inf"), 1.0, -2.0, 0.0, 3.0])
    add("both_inf", [float("inf"), float("-inf"), 1.0])
    add("nan", [float("nan"), 1.0, -2.0, 0.0, 3.0])
    add("nan_and_inf", [float("nan"), float("inf"), 1.0])
    add("outlier_1e30", torch.cat([torch.tensor([1e30]), torch.randn(cols - 1, generator=g)]))
    add("near_fmax", [3.0e38, -1.0e38, 1.0, 1e30])
    add("denormal", [1e-40, -3e-41, 1e-39, 0.0])
    add("one_hot", [0.0, 0.0, 1.0])
    add("all_equal", torch.full((cols,), 0.7))
    add("all_equal_neg", torch.full((cols,), -0.3))
    add("positive_only", torch.rand(cols, generator=g) + 0.25)
    add("fp16_overflowing_scale", torch.randn(cols, generator=g) * 5e-6)
    add("big", torch.randn(cols, generator=g) * 1000.0)
    add("big_60000", torch.randn(cols, generator=g).clamp(-1, 1) * 60000.0)
    return torch.stack(rows).to(dtype), names


def act_like(gen, shape, dtype):
    """N(0,1) with 0.1 % x20 outliers (SURVEY §8d activation-style input)."""
    x = torch.randn(shape, generator=gen, dtype=torch.float32)
    m = torch.rand(shape, generator=gen) < 1e-3
    x[m] *= 20.0
    return x.to(dtype)


# --------------------------------------------------------------------------------------
# forward fixtures
# --------------------------------------------------------------------------------------
def build_fwd(kind):
    arrays, manifest = {}, []
    gen = torch.Generator().manual_seed(1234 if kind == "sym" else 4321)
    quant = SymQuantizer if kind == "sym" else AsymQuantizer

    def emit(name, x, bits, layerwise, dname, extra=None):
        y_ref = quant.apply(x, CLIP, bits, layerwise)
        if kind == "sym":
            idx, s, y2 = sym_parts(x, bits, layerwise)
        else:
            idx, alpha, beta, y2 = asym_parts(x, bits, layerwise)
        assert bits_equal(y_ref, y2), f"restated op sequence diverged from reference in {name}"
        assert y_ref.dtype == x.dtype and y_ref.shape == x.shape
        arrays[f"{name}/x"] = to_np(x)
        arrays[f"{name}/y"] = to_np(y_ref)
        arrays[f"{name}/idx"] = idx_to_i32(idx)
        # one statistic per row (first element of the expanded view is enough)
        lead = x.reshape(-1, x.shape[-1]).shape[0] if (not layerwise and x.ndimension() <= 3) else None
        if kind == "sym":
            srow = s.reshape(-1, x.shape[-1])[:, 0] if lead else s.reshape(-1)[:1] if layerwise else s.reshape(x.shape[0] * x.shape[1], -1)[:, 0]
            arrays[f"{name}/scale"] = to_np(srow.contiguous())
        else:
            def per_row(t):
                t = t.expand_as(x) if t.dim() == 0 else t
                return (t.reshape(-1, x.shape[-1])[:, 0] if lead else t.reshape(-1)[:1] if layerwise else t.reshape(x.shape[0] * x.shape[1], -1)[:, 0]).contiguous()
            arrays[f"{name}/alpha"] = to_np(per_row(alpha))
            arrays[f"{name}/beta"] = to_np(per_row(beta))
        m = dict(name=name, kind=kind, dtype=dname, bits=bits, layerwise=bool(layerwise), shape=list(x.shape))
        if extra:
            m.update(extra)
        manifest.append(m)

    for dname, dt in DT.items():
        for bits in (3, 4, 8, 16):
            # ragged small shapes, 1-D .. 4-D
            for shape in [(9,), (5, 1), (3, 7), (7, 33), (2, 255), (2, 3, 64), (2, 3, 4, 5)]:
                x = rand_rows(gen, shape, dt)
                emit(f"{kind}_{dname}_b{bits}_{'x'.join(map(str, shape))}", x, bits, False, dname)
            x = rand_rows(gen, (4, 50), dt, scales=[0.02, 1.0, 5.0])
            emit(f"{kind}_{dname}_b{bits}_layerwise_4x50", x, bits, True, dname)
            x = rand_rows(gen, (2, 3, 4, 5), dt, scales=[0.5])
            emit(f"{kind}_{dname}_b{bits}_layerwise_4d", x, bits, True, dname)
            xa, names = adversarial(dt)
            emit(f"{kind}_{dname}_b{bits}_adversarial", xa, bits, False, dname, extra=dict(row_names=names))
        # layerwise with a NaN / Inf somewhere
        xa, _ = adversarial(dt)
        emit(f"{kind}_{dname}_b4_layerwise_adversarial_finite", xa[:9].clone(), 4, True, dname)
        emit(f"{kind}_{dname}_b8_layerwise_adversarial_all", xa.clone(), 8, True, dname)

    # model-sized rows (kept few: the npz is compressed but inputs are random)
    big = [("bf16", 4, (2, 4096)), ("bf16", 8, (2, 4096)), ("bf16", 4, (1, 11008)), ("bf16", 8, (1, 11008)),
           ("fp32", 4, (1, 11008)), ("fp32", 8, (2, 256)), ("fp32", 8, (2, 688)), ("bf16", 8, (1, 5120)),
           ("bf16", 4, (1, 13824)), ("fp16", 8, (1, 4096))]
    for dname, bits, shape in big:
        dt = DT[dname]
        if bits == 4:  # weight-style N(0, 0.02^2)
            x = (torch.randn(shape, generator=gen) * 0.02).to(dt)
        else:
            x = act_like(gen, shape, dt)
        emit(f"{kind}_{dname}_b{bits}_model_{'x'.join(map(str, shape))}", x, bits, False, dname)

    # the ends of the bit-width domain (round 4; appended with their own generator so that every earlier array regenerates byte-identical):
    # 1-bit Sym has qmax = 2**0 - 1 = 0 -- reachable through the KV hooks' `kv_bits < 32` gate --, 2 and 31 bits are the other Sym ends;
    # Asym: 1 and 24 bits
    gen2 = torch.Generator().manual_seed(9876 if kind == "sym" else 6789)
    for dname, dt in DT.items():
        for bits in ((1, 2, 31) if kind == "sym" else (1, 24)):
            for shape in [(3, 7), (2, 255), (2, 3, 64)]:
                emit(f"{kind}_{dname}_b{bits}_edge_{'x'.join(map(str, shape))}", rand_rows(gen2, shape, dt), bits, False, dname)
            xa, names = adversarial(dt)
            emit(f"{kind}_{dname}_b{bits}_edge_adversarial", xa, bits, False, dname, extra=dict(row_names=names))
    return arrays, manifest


# --------------------------------------------------------------------------------------
# STE backward fixtures (utils_quant.py:77-87 / :152-162)
# --------------------------------------------------------------------------------------
def build_bwd():
    arrays, manifest = {}, []
    gen = torch.Generator().manual_seed(99)

    def emit(name, quant, x, g, clip, bits, dname):
        xr = x.clone().requires_grad_(True)
        y = quant.apply(xr, clip, bits, False)
        y.backward(g)
        gx = xr.grad
        assert gx.dtype == g.dtype
        arrays[f"{name}/x"] = to_np(x)
        arrays[f"{name}/g"] = to_np(g)
        arrays[f"{name}/gx"] = to_np(gx)
        arrays[f"{name}/clip"] = clip.numpy().astype(np.float32)
        manifest.append(dict(name=name, quant=quant.__name__, dtype=dname, bits=bits, shape=list(x.shape)))

    for dname, dt in DT.items():
        special = torch.tensor([2.0, -2.0, 1.9921875, -1.9921875, 2.015625, -2.015625, float("nan"), float("inf"),
                                float("-inf"), 0.0, -0.0, 0.75, -0.5, 0.7421875, -0.498046875, 0.30078125, 0.3,
                                0.298828125, -0.30078125, 0.302734375, -0.302734375, 1e-40, 3e38])
        x = torch.cat([special, torch.randn(300 - special.numel(), generator=gen) * 1.5]).reshape(3, 100).to(dt)
        g = torch.randn(3, 100, generator=gen).to(dt)
        g[0, 5] = float("nan")
        g[1, 7] = float("inf")
        emit(f"ste_{dname}_sym_default", SymQuantizer, x, g, torch.tensor([-2.0, 2.0]), 8, dname)
        emit(f"ste_{dname}_asym_default", AsymQuantizer, x, g, torch.tensor([-2.0, 2.0]), 8, dname)
        emit(f"ste_{dname}_sym_custom", SymQuantizer, x, g, torch.tensor([-0.5, 0.75]), 4, dname)
        # clip values that are NOT representable in bf16/fp16 and round TOWARDS zero there
        # (0.3009 -> 0.30078125 in both): x = +-0.30078125 is masked only if the compare
        # happens in the tensor dtype -> pins the compare dtype
        emit(f"ste_{dname}_sym_unrepresentable", SymQuantizer, x, g, torch.tensor([-0.3009, 0.3009]), 4, dname)
        x3 = act_like(gen, (2, 5, 33), dt)
        g3 = (torch.randn(2, 5, 33, generator=gen) * 1e-3).to(dt)
        emit(f"ste_{dname}_sym_3d", SymQuantizer, x3, g3, torch.tensor([-2.0, 2.0]), 8, dname)
    return arrays, manifest


# --------------------------------------------------------------------------------------
# QuantizeLinear fixtures (utils_quant.py:165-254)
# --------------------------------------------------------------------------------------
def build_linear():
    import torch.nn.functional as F
    arrays, manifest = {}, []
    gen = torch.Generator().manual_seed(2024)
    captured = {}
    real_linear = F.linear

    def spy(inp, weight, bias=None):   # the operands exactly as the module hands them to the GEMM (round 4: pins them at module level)
        captured["x"], captured["w"] = inp.detach().clone(), weight.detach().clone()
        return real_linear(inp, weight, bias)

    combos = [
        dict(w_bits=4, a_bits=8, symmetric=True),
        dict(w_bits=8, a_bits=8, symmetric=True),
        dict(w_bits=4, a_bits=8, symmetric=False),
        dict(w_bits=4, a_bits=16, symmetric=True),
        dict(w_bits=32, a_bits=8, symmetric=True),
        dict(w_bits=4, a_bits=32, symmetric=True),
        dict(w_bits=4, a_bits=2, symmetric=True),       # a_bits<=2 silently disables act quant (:184,:244)
        dict(w_bits=8, a_bits=8, symmetric=True, act_layerwise=True, weight_layerwise=True),
        dict(w_bits=1, a_bits=8, symmetric=True),
        dict(w_bits=2, a_bits=8, symmetric=True),
        dict(w_bits=1, a_bits=8, symmetric=True, weight_layerwise=True),
        dict(w_bits=2, a_bits=8, symmetric=True, weight_layerwise=True),
    ]
    for dname in ("fp32", "bf16"):
        dt = DT[dname]
        for ci, kw in enumerate(combos):
            in_f, out_f = 48, 20
            lin = QuantizeLinear(in_f, out_f, bias=True, **kw)
            assert lin.bias is None and list(lin.state_dict().keys()) == ["weight"]
            w = (torch.randn(out_f, in_f, generator=gen) * 0.05)
            w[3, 5] = 2.5   # one weight beyond the STE clip
            w[4, 6] = -2.0
            lin.weight.data = w.to(dt)
            x = act_like(gen, (2, 5, in_f), dt).requires_grad_(True)
            F.linear = torch.nn.functional.linear = spy
            try:
                out = lin(x)
            finally:
                F.linear = torch.nn.functional.linear = real_linear
            go = (torch.randn(out.shape, generator=gen) * 0.1).to(dt)
            out.backward(go)
            name = f"lin_{dname}_{ci}"
            arrays[f"{name}/w"] = to_np(lin.weight.data)
            arrays[f"{name}/x"] = to_np(x)
            arrays[f"{name}/go"] = to_np(go)
            arrays[f"{name}/out"] = to_np(out)
            arrays[f"{name}/gw"] = to_np(lin.weight.grad)
            arrays[f"{name}/gx"] = to_np(x.grad)
            arrays[f"{name}/opx"] = to_np(captured["x"])
            arrays[f"{name}/opw"] = to_np(captured["w"])
            manifest.append(dict(name=name, dtype=dname, in_features=in_f, out_features=out_f, **kw))
    return arrays, manifest


# --------------------------------------------------------------------------------------
# 1-/2-bit weight branches of QuantizeLinear (utils_quant.py:202-242): the tensor handed to F.linear
# --------------------------------------------------------------------------------------
def build_w12():
    import torch.nn.functional as F
    arrays, manifest = {}, []
    gen = torch.Generator().manual_seed(31337)
    captured = {}
    real_linear = F.linear

    def spy(inp, weight, bias=None):
        captured["w"] = weight.detach().clone()
        return real_linear(inp, weight, bias)

    for dname, dt in DT.items():
        for w_bits in (1, 2):
            for layerwise in (False, True):
                for shape in [(6, 40), (3, 33), (5, 256)]:
                    w = torch.randn(shape, generator=gen) * 0.05
                    if shape == (6, 40):
                        w[1] = 0.0                      # all-zero row: scale 0 -> 0/0
                        w[2, :5] = torch.tensor([1e-3, -1e-3, 0.0, 5.0, -5.0])
                        w[3, 0] = float("nan")
                        w[4] *= 1e-6
                    lin = QuantizeLinear(shape[1], shape[0], w_bits=w_bits, a_bits=32, weight_layerwise=layerwise)
                    lin.weight.data = w.to(dt)
                    F.linear = torch.nn.functional.linear = spy
                    try:
                        lin(torch.zeros(1, shape[1], dtype=dt))
                    finally:
                        F.linear = torch.nn.functional.linear = real_linear
                    wq = captured["w"]
                    real = lin.weight.data
                    if layerwise:
                        sc = torch.mean(abs(real)) if w_bits == 1 else 2 * torch.mean(abs(real))
                    else:
                        sc = torch.mean(abs(real), dim=1, keepdim=True) if w_bits == 1 else 2 * torch.mean(abs(real), dim=1, keepdim=True)
                    name = f"w12_{dname}_b{w_bits}_{'lw' if layerwise else 'row'}_{shape[0]}x{shape[1]}"
                    arrays[f"{name}/w"] = to_np(real)
                    arrays[f"{name}/scale"] = to_np(sc.reshape(-1).contiguous())
                    arrays[f"{name}/wq"] = to_np(wq)
                    manifest.append(dict(name=name, dtype=dname, w_bits=w_bits, layerwise=layerwise, shape=list(shape)))
    return arrays, manifest


def main():
    torch.set_num_threads(1)
    meta = dict(torch=torch.__version__, numpy=np.__version__, reference="JingyangXiang/LLM-QAT @ 2024_08_07",
                source="models/utils_quant.py imported on CPU", python=sys.version.split()[0])
    for fname, (arrays, manifest) in {
        "sym_fwd.npz": build_fwd("sym"),
        "asym_fwd.npz": build_fwd("asym"),
        "ste_bwd.npz": build_bwd(),
        "quantize_linear.npz": build_linear(),
        "w12.npz": build_w12(),
    }.items():
        arrays["manifest"] = np.frombuffer(json.dumps(dict(meta=meta, cases=manifest)).encode(), dtype=np.uint8)
        path = os.path.join(HERE, fname)
        np.savez_compressed(path, **arrays)
        print(f"{fname}: {len(manifest)} cases, {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
