#!/usr/bin/env python3
"""Golden vectors for the arithmetic LLM-QAT actually trains with: the reference under CUDA autocast.

run_train.sh:17-18 (`--bf16 True`) -> utils/kd_trainer.py:106 runs the student inside torch.autocast("cuda", bf16), where
`SymQuantizer.forward` (models/utils_quant.py:71-72) computes in fp32 behind its `reciprocal` and returns fp32.  This script
imports the REAL reference module, runs the real `SymQuantizer.apply` / `QuantizeLinear.forward` on CPU bf16 / fp16 tensors with
the device's cast policy imposed from outside (tests/autocast_policy.py -- which ops it rewrites, and how an op it does not know
would fail the run), and records inputs + what the reference itself produced:

    y        the fp32 result of SymQuantizer.apply                      (float32)
    y_narrow y rounded once to the tensor dtype by torch                (what F.linear's autocast cast makes of it)
    idx      the output of the reference's own torch.round              (int32; kept by the dispatch mode, not restated)
    scale    per-row s = reciprocal(max + 1e-6) * qmax                  (float32; kept by the dispatch mode)
    g, gx    an fp32 gradient of y and the gradient that reaches x      (the autograd engine casts it to x's dtype)
  QuantizeLinear cases: w, x, go, out, gw, gx and opx / opw = the operands exactly as handed to the GEMM.

Two scalar policies per case where they can differ (rows whose |max| is below ~4e-5): "cpu" = the reference's CPU behaviour for
`max + 1e-6` (scalar rounded to the tensor dtype first) and "device" = ATen's GPU behaviour (scalar stays an fp32 opmath value).
Cases where both give the same bits are stored once as "both" (the script checks it).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_autocast.py

Runs ONLY in the build container (imports /root/reference); nothing of the reference travels: the .npz holds arrays + a JSON
manifest.
"""
import json
import os
import sys

sys.dont_write_bytecode = True
REF = os.environ.get("LLMQAT_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from models.utils_quant import QuantizeLinear, SymQuantizer  # noqa: E402

from autocast_policy import cuda_autocast_policy  # noqa: E402
from make_golden import DT, act_like, adversarial, idx_to_i32, rand_rows, to_np  # noqa: E402

CLIP = torch.tensor([-2.0, 2.0])


def same_bits(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype == np.float32:
        a, b = a.view(np.uint32), b.view(np.uint32)
    return a.shape == b.shape and bool((a == b).all())


def run_sym(x, bits, layerwise, adt, device_scalars, g=None, clip=CLIP):
    """the real SymQuantizer.apply under the cast policy -> dict of arrays"""
    xr = x.clone().requires_grad_(g is not None)
    with cuda_autocast_policy(adt, device_scalars) as p:
        y = SymQuantizer.apply(xr, clip, bits, layerwise)
    assert y.dtype == torch.float32 and y.shape == x.shape, (y.dtype, y.shape)
    assert p.casts.cast_ops == ["reciprocal"], p.casts.cast_ops
    (idx,), (s,) = p.casts.rounds, p.casts.scales
    assert idx.dtype == torch.float32 and s.dtype == torch.float32
    rows = x.numel() if layerwise else (x.shape[0] * x.shape[1] if x.dim() == 4 else x.numel() // x.shape[-1])
    out = dict(x=to_np(x), y=to_np(y), y_narrow=to_np(y.detach().to(x.dtype)), idx=idx_to_i32(idx),
               scale=to_np(s.expand_as(x).reshape(1 if layerwise else rows, -1)[:, 0].contiguous()))
    if g is not None:
        y.backward(g)   # outside the policy, as HF Trainer's backward runs outside the autocast context
        assert xr.grad.dtype == x.dtype
        out.update(g=to_np(g), gx=to_np(xr.grad), clip=clip.numpy().astype(np.float32))
    return out


def build_ops():
    arrays, manifest = {}, []
    gen = torch.Generator().manual_seed(20240807)

    def emit(name, x, bits, layerwise, dname, with_grad=True, clip=CLIP, extra=None):
        g = (torch.randn(x.shape, generator=gen) * 1e-3) if with_grad else None
        if g is not None and g.numel() > 8:
            g.view(-1)[3], g.view(-1)[5] = float("nan"), float("inf")
        res = {sc: run_sym(x, bits, layerwise, DT[dname], sc == "device", g, clip) for sc in ("cpu", "device")}
        same = all(same_bits(res["cpu"][k], res["device"][k]) for k in res["cpu"])
        for sc, r in ((("both", res["device"]),) if same else tuple(res.items())):
            cname = f"{name}_{sc}"
            for k, v in r.items():
                arrays[f"{cname}/{k}"] = v
            m = dict(name=cname, dtype=dname, autocast_dtype=dname, bits=bits, layerwise=bool(layerwise), shape=list(x.shape), scalars=sc,
                     grad=with_grad)
            if extra:
                m.update(extra)
            manifest.append(m)

    for dname in ("bf16", "fp16"):
        dt = DT[dname]
        for bits in (3, 4, 8, 16):
            xa, names = adversarial(dt)
            emit(f"ac_{dname}_b{bits}_adversarial", xa, bits, False, dname, extra=dict(row_names=names))
        for bits in (4, 8):
            for shape in [(9,), (3, 7), (7, 33), (2, 255), (5, 264), (2, 3, 64), (2, 3, 4, 5)]:
                emit(f"ac_{dname}_b{bits}_{'x'.join(map(str, shape))}", rand_rows(gen, shape, dt), bits, False, dname)
            emit(f"ac_{dname}_b{bits}_layerwise_4x50", rand_rows(gen, (4, 50), dt, scales=[0.02, 1.0, 5.0]), bits, True, dname)
        # rows whose |max| is small enough for the two scalar policies to part (`max + 1e-6` in the tensor dtype: bf16 below ~3e-4;
        # fp16 where 1e-6 lands on a rounding tie of the sum, |max| in [2^-13, 2^-12))
        tiny = torch.randn(28, 40, generator=gen) * torch.tensor([2e-7, 2e-6, 1e-5, 3e-5, 6e-5, 8e-5, 1e-4] * 4).unsqueeze(1)
        emit(f"ac_{dname}_b8_tiny_rows", tiny.to(dt), 8, False, dname)
        emit(f"ac_{dname}_b4_tiny_rows", tiny.to(dt), 4, False, dname)
        # a clip other than the model's, and one that is not representable in the tensor dtype (pins the compare dtype)
        x = torch.cat([torch.tensor([2.0, -2.0, 0.75, -0.5, 0.30078125, -0.30078125, 0.302734375, float("nan"), float("inf")]),
                       torch.randn(311, generator=gen)]).reshape(5, 64).to(dt)
        emit(f"ac_{dname}_b8_clip_custom", x, 8, False, dname, clip=torch.tensor([-0.5, 0.75]))
        emit(f"ac_{dname}_b4_clip_unrepresentable", x, 4, False, dname, clip=torch.tensor([-0.3009, 0.3009]))
    # model widths (LLaMA-7B 4096 / 11008, 13B 5120 / 13824): weight-style N(0, 0.02^2) at 4 bits, activation-style at 8
    for dname, bits, shape in [("bf16", 4, (2, 4096)), ("bf16", 8, (2, 4096)), ("bf16", 4, (1, 11008)), ("bf16", 8, (1, 11008)),
                               ("bf16", 8, (1, 5120)), ("bf16", 4, (1, 13824)), ("fp16", 8, (1, 4096)), ("fp16", 4, (1, 11008))]:
        x = (torch.randn(shape, generator=gen) * 0.02).to(DT[dname]) if bits == 4 else act_like(gen, shape, DT[dname])
        emit(f"ac_{dname}_b{bits}_model_{'x'.join(map(str, shape))}", x, bits, False, dname)
    return arrays, manifest


def build_linear():
    """QuantizeLinear.forward + backward under the policy (utils_quant.py:190-254)"""
    arrays, manifest = {}, []
    gen = torch.Generator().manual_seed(31)
    combos = [
        dict(w_bits=4, a_bits=8, symmetric=True),
        dict(w_bits=8, a_bits=8, symmetric=True),
        dict(w_bits=4, a_bits=16, symmetric=True),
        dict(w_bits=32, a_bits=8, symmetric=True),
        dict(w_bits=4, a_bits=32, symmetric=True),
        dict(w_bits=8, a_bits=8, symmetric=True, act_layerwise=True, weight_layerwise=True),
        dict(w_bits=1, a_bits=8, symmetric=True),
        dict(w_bits=2, a_bits=8, symmetric=True),
        dict(w_bits=4, a_bits=8, symmetric=False),
    ]
    # (tensor dtype, autocast dtype): the usual pairs + an fp16 model inside autocast(bf16) and the reverse
    for dname, aname in (("bf16", "bf16"), ("fp16", "fp16"), ("fp16", "bf16"), ("bf16", "fp16")):
        dt, adt = DT[dname], DT[aname]
        for ci, kw in enumerate(combos):
            if dname != aname and ci not in (0, 1):
                continue
            in_f, out_f = 64, 24
            w = torch.randn(out_f, in_f, generator=gen) * 0.05
            w[3, 5], w[4, 6] = 2.5, -2.0   # weights at / beyond the STE clip
            x0 = act_like(gen, (2, 5, in_f), dt)
            go = None
            res = {}
            for sc in ("cpu", "device"):
                lin = QuantizeLinear(in_f, out_f, bias=True, **kw)
                lin.weight.data = w.to(dt)
                x = x0.clone().requires_grad_(True)
                with cuda_autocast_policy(adt, sc == "device") as p:
                    out = lin(x)
                assert out.dtype == adt, out.dtype
                if go is None:
                    go = (torch.randn(out.shape, generator=gen) * 0.1).to(adt)
                out.backward(go)
                (opx, opw), = p.linear.operands
                assert opx.dtype == adt and opw.dtype == adt
                res[sc] = dict(w=to_np(lin.weight.data), x=to_np(x0), go=to_np(go), out=to_np(out), gw=to_np(lin.weight.grad), gx=to_np(x.grad),
                               opx=to_np(opx), opw=to_np(opw))
            same = all(same_bits(res["cpu"][k], res["device"][k]) for k in res["cpu"])
            # Asym's `.div(python int)` has its own CPU / device difference (the library's `sem` knob) that this policy does not
            # model: those cases are stored under the reference's CPU behaviour only
            keep = ("both",) if same else (("cpu",) if not kw["symmetric"] else ("cpu", "device"))
            for sc in keep:
                name = f"aclin_{dname}_in_{aname}_{ci}_{sc}"
                for k, v in res["device" if sc == "both" else sc].items():
                    arrays[f"{name}/{k}"] = v
                manifest.append(dict(name=name, dtype=dname, autocast_dtype=aname, in_features=in_f, out_features=out_f, scalars=sc, **kw))
    return arrays, manifest


def main():
    torch.set_num_threads(1)
    meta = dict(torch=torch.__version__, numpy=np.__version__, reference="JingyangXiang/LLM-QAT @ 2024_08_07",
                source="models/utils_quant.py imported on CPU, run under tests/autocast_policy.py (CUDA autocast's cast policy)",
                policy=dict(fp32_cast=["aten.reciprocal"], lower_precision_cast=["torch.nn.functional.linear"],
                            device_scalars="add(16-bit tensor, python float) computed as fp32 add of the fp32 scalar, one rounding"),
                python=sys.version.split()[0])
    ops_arrays, ops_manifest = build_ops()
    lin_arrays, lin_manifest = build_linear()
    arrays = {**ops_arrays, **lin_arrays}
    arrays["manifest"] = np.frombuffer(json.dumps(dict(meta=meta, cases=ops_manifest, linear_cases=lin_manifest)).encode(), dtype=np.uint8)
    path = os.path.join(HERE, "autocast.npz")
    np.savez_compressed(path, **arrays)
    print(f"autocast.npz: {len(ops_manifest)} op cases + {len(lin_manifest)} module cases, {os.path.getsize(path) / 1024:.1f} KiB")
    for sc in ("both", "cpu", "device"):
        print(f"  scalars={sc}: {sum(m['scalars'] == sc for m in ops_manifest)} op / {sum(m['scalars'] == sc for m in lin_manifest)} module cases")


if __name__ == "__main__":
    main()
