#!/usr/bin/env python3
"""Per-decoder-layer fake-quant cost on MI355X: every quantizer invocation of one LLaMA layer
(BASELINE.json configs[1..4]) forward + backward, through the product's autograd Functions (Python +
allocator overhead included) vs the reference's eager op chain on the same GPU.

    python tools/layer_bench.py [--model 7b|13b] [--iters 30]

One layer at seq 2048, bs 1: weights 4x[h,h] + 2x[m,h] + 1x[h,m]; activations: the q/k/v input
(quantized 3x in the reference, once here), the o_proj input, the gate/up input (2x -> once), the down_proj
input; K and V projections.  Reports ms per layer (fwd, bwd) and the wall/host time per call.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

MODELS = {"7b": (4096, 11008), "13b": (5120, 13824)}


def build(h, m, seq, dev):
    g = torch.Generator(device=dev).manual_seed(0)
    W = [torch.randn(s, generator=g, device=dev).mul_(0.02).bfloat16() for s in [(h, h)] * 4 + [(m, h)] * 2 + [(h, m)]]

    def act(shape):
        a = torch.randn(shape, generator=g, device=dev)
        a[torch.rand(shape, generator=g, device=dev) < 1e-3] *= 20
        return a.bfloat16()

    A = dict(qkv=act((1, seq, h)), o=act((1, seq, h)), gateup=act((1, seq, h)), down=act((1, seq, m)), k=act((1, seq, h)), v=act((1, seq, h)))
    return W, A


def run_layer(sym, W, A, wb, ab, kvb, dedup):
    clip = torch.tensor([-2.0, 2.0])
    outs, leaves = [], []

    def q(t, bits):
        t = t.detach().requires_grad_(True)
        leaves.append(t)
        y = sym.apply(t, clip, bits, False)
        outs.append(y)

    for w in W:
        q(w, wb)
    for name, reps in (("qkv", 3), ("o", 1), ("gateup", 2), ("down", 1)):
        for _ in range(1 if dedup else reps):
            q(A[name], ab)
    q(A["k"], kvb)
    q(A["v"], kvb)
    return outs, leaves


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    host = (time.perf_counter() - t0) / iters
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, host * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="7b")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--seq", type=int, default=2048)
    args = ap.parse_args()
    import llm_qat_amd
    from llm_qat_amd.utils_quant import SymQuantizer
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tiny_llama import EagerQuant   # test infrastructure: the reference's eager op chain as autograd Functions
    EagerSym = EagerQuant().SymQuantizer
    dev = torch.device("cuda:0")
    h, m = MODELS[args.model]
    W, A = build(h, m, args.seq, dev)
    res = {"model": args.model, "seq": args.seq, "rows": []}
    for wb, ab, kvb in ((4, 8, 4), (4, 8, 8), (8, 8, 8)):
        for label, sym, dedup in (("reference eager chain", EagerSym, False), ("llm_qat_amd", SymQuantizer, True)):
            def fwd():
                return run_layer(sym, W, A, wb, ab, kvb, dedup)

            grads = [torch.ones_like(o) for o in fwd()[0]]

            def fwdbwd():
                outs, leaves = run_layer(sym, W, A, wb, ab, kvb, dedup)
                torch.autograd.backward(outs, grads)

            f_ms, f_host = timed(fwd, args.iters)
            fb_ms, fb_host = timed(fwdbwd, args.iters)
            ncalls = len(fwd()[0])
            elems = sum(o.numel() for o in fwd()[0]) if not dedup else None
            res["rows"].append(dict(cfg=f"W{wb}A{ab}KV{kvb}", impl=label, quantizer_calls=ncalls, fwd_ms=round(f_ms, 3),
                                    fwd_bwd_ms=round(fb_ms, 3), host_ms_fwd=round(f_host, 3), host_ms_fwd_bwd=round(fb_host, 3)))
            print(res["rows"][-1], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"layer_bench_{args.model}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
