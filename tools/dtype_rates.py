#!/usr/bin/env python3
"""Byte rates of the fake-quant kernels per tensor dtype at the metric shape [4096,11008] (bench.py quotes bf16 only): forward in training
mode (bounds + mask recorded), its mask backward, both semantics, W4-style and A8-style data; Asym A8; the autocast forms for the 16-bit
dtypes.  Every figure: mean of 200 back-to-back launches after 20 warm-ups over 4 rotating buffer sets (> 256 MiB apart in total).

    python tools/dtype_rates.py [--json out.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ROWS, COLS, NSETS = 4096, 11008, 4


def timed(torch, fn, iters=200, warm=20):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def mask_bytes(torch, side, lo=-2.0, hi=2.0):
    """bytes of STE mask the kernels actually write / read: 1 bit per element of the rows whose recorded bounds reach the clip (the side
    buffer is float[rows][2] bounds followed by the row bitmap; rows that cannot clip have no mask traffic)"""
    b = side[: ROWS * 8].view(torch.float32).view(ROWS, 2)
    clippable = int(((b[:, 0] >= hi) | (b[:, 1] <= lo) | (b != b).any(dim=1)).sum())   # bounds are {max, min}
    return clippable * COLS // 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    args = ap.parse_args()
    import torch
    import llm_qat_amd
    from llm_qat_amd import ops
    g = torch.Generator(device="cuda").manual_seed(0)
    rows_out = []
    for dtype in (torch.bfloat16, torch.float16, torch.float32):
        es = torch.empty((), dtype=dtype).element_size()
        n = ROWS * COLS
        ws = [(torch.randn(ROWS, COLS, generator=g, device="cuda") * 0.02).to(dtype) for _ in range(NSETS)]
        acts = []
        for _ in range(NSETS):
            a = torch.randn(ROWS, COLS, generator=g, device="cuda")
            a.view(-1)[::997] *= 20.0
            acts.append(a.to(dtype))
        gs = [(torch.randn(ROWS, COLS, generator=g, device="cuda") * 1e-2).to(dtype) for _ in range(NSETS)]
        for sem in ("cpu_eager", "device_eager"):
            llm_qat_amd.set_semantics(sem)
            for name, xs, bits, kind in (("sym W4 fwd (train)", ws, 4, "sym"), ("sym A8 fwd (train)", acts, 8, "sym"), ("asym A8 fwd (train)", acts, 8, "asym")):
                res = [ops.train_forward(kind, x, bits, False, -2.0, 2.0) for x in xs]
                if res[0] is None:
                    continue
                t = timed(torch, lambda i: ops.train_forward(kind, xs[i % NSETS], bits, False, -2.0, 2.0))
                moved = n * es * 2 + ROWS * 8 + mask_bytes(torch, res[0][1])
                rows_out.append({"dtype": str(dtype).split(".")[1], "semantics": sem, "kernel": name, "us": round(t, 2), "MB_moved": round(moved / 1e6, 1),
                                 "GBs": round(moved / t / 1e3, 1), "frac": round(moved / t / 1e3 / 8000.0, 3)})
                print(json.dumps(rows_out[-1]), flush=True)
                if kind == "sym" and bits == 8 and sem == "cpu_eager":
                    sides = [r[1] for r in res]
                    t = timed(torch, lambda i: ops.train_backward(gs[i % NSETS], sides[i % NSETS], ROWS, COLS, -2.0, 2.0))
                    moved = n * es * 2 + mask_bytes(torch, sides[0])
                    rows_out.append({"dtype": str(dtype).split(".")[1], "semantics": "-", "kernel": "ste mask bwd (A8 tensor)", "us": round(t, 2),
                                     "MB_moved": round(moved / 1e6, 1), "GBs": round(moved / t / 1e3, 1), "frac": round(moved / t / 1e3 / 8000.0, 3)})
                    print(json.dumps(rows_out[-1]), flush=True)
                del res
        llm_qat_amd.set_semantics("cpu_eager")
        if dtype != torch.float32:
            for name, xs, bits in (("sym W4 fwd autocast, rounded once (operand)", ws, 4), ("sym A8 fwd autocast, fp32 result", acts, 8)):
                wide = "fp32" in name
                t = timed(torch, lambda i: ops.sym_forward_autocast(xs[i % NSETS], bits, False, wide=wide, train="mask"))
                side = ops.sym_forward_autocast(xs[0], bits, False, wide=wide, train="mask")[1]
                moved = n * es + n * (4 if wide else es) + ROWS * 8 + mask_bytes(torch, side)
                rows_out.append({"dtype": str(dtype).split(".")[1], "semantics": "autocast", "kernel": name, "us": round(t, 2), "MB_moved": round(moved / 1e6, 1),
                                 "GBs": round(moved / t / 1e3, 1), "frac": round(moved / t / 1e3 / 8000.0, 3)})
                print(json.dumps(rows_out[-1]), flush=True)
        del ws, acts, gs
        torch.cuda.empty_cache()
    if args.json:
        os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
        with open(args.json, "w") as f:
            json.dump({"what": __doc__, "device": torch.cuda.get_device_name(0), "rows": rows_out}, f, indent=1)


if __name__ == "__main__":
    main()
