#!/usr/bin/env python3
"""Why does the [2048,4096] bf16 training forward read 9-10 us in tools/shape_sweep.py when its autocast sibling of the
same shape and byte count reads 7.2 us?

Times the candidate kinds on the SAME buffers in interleaved rounds (A B C D, then D C B A, ...), each kind with its own
warm-up, so neither allocation order nor "who runs first after the allocator touched the memory" can masquerade as a
kernel property.  Run it bare and under `rocprofv3 --kernel-trace --stats` (per-kernel durations of
row_reg_kernel<1,256,2,...,AC=0> vs <...,AC=1>).

    python tools/small_shape_probe.py [--rows 2048 --cols 4096] -> gpurun_out/small_shape_probe.json
"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from llm_qat_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2048)
    ap.add_argument("--cols", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--nsets", type=int, default=12)
    args = ap.parse_args()
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    rows, cols, code = args.rows, args.cols, _lib.DTYPE_BF16
    g = torch.Generator(device=dev).manual_seed(1)
    mb = L.fq_ste_mask_bytes(rows, cols, code)
    sets = []
    for _ in range(args.nsets):
        x = torch.randn(rows, cols, generator=g, device=dev)
        x[torch.rand(rows, cols, generator=g, device=dev) < 1e-3] *= 20
        x = x.bfloat16()
        sets.append(dict(x=x, y=torch.empty_like(x), b=torch.empty(rows, 2, device=dev), m=torch.empty(mb, dtype=torch.uint8, device=dev)))

    def chk(rc):
        if rc:
            _lib.check(rc, "probe")

    kinds = {
        "train_a8": lambda s: chk(L.fq_sym_fwd_train(s["x"].data_ptr(), s["y"].data_ptr(), rows, cols, 8, code, 0, -2.0, 2.0, s["b"].data_ptr(), s["m"].data_ptr(), mb, st)),
        "train_kv4": lambda s: chk(L.fq_sym_fwd_train(s["x"].data_ptr(), s["y"].data_ptr(), rows, cols, 4, code, 0, -2.0, 2.0, s["b"].data_ptr(), s["m"].data_ptr(), mb, st)),
        "plain_a8": lambda s: chk(L.fq_sym_fwd(s["x"].data_ptr(), s["y"].data_ptr(), rows, cols, 8, code, 0, None, None, 0, st)),
        "autocast_narrow_a8": lambda s: chk(L.fq_sym_fwd_autocast(s["x"].data_ptr(), s["y"].data_ptr(), rows, cols, 8, code, 1, 0, -2.0, 2.0, s["b"].data_ptr(), s["m"].data_ptr(), mb, None, 0, st)),
        "autocast_narrow_a8_nomask": lambda s: chk(L.fq_sym_fwd_autocast(s["x"].data_ptr(), s["y"].data_ptr(), rows, cols, 8, code, 1, 0, -2.0, 2.0, None, None, 0, None, 0, st)),
    }
    names = list(kinds)
    res = {k: [] for k in names}

    def timed(fn):
        for i in range(5):
            fn(sets[i % len(sets)])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(args.iters):
            fn(sets[i % len(sets)])
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.iters * 1e3

    for r in range(args.rounds):
        order = names if r % 2 == 0 else names[::-1]
        for k in order:
            res[k].append(timed(kinds[k]))
    out = {"shape": [rows, cols], "dtype": "bf16", "bytes_per_launch": rows * cols * 4, "nsets": len(sets), "rounds": args.rounds,
           "iters_per_round": args.iters, "kinds": {}}
    for k in names:
        v = sorted(res[k])
        out["kinds"][k] = {"us_min": round(v[0], 2), "us_median": round(statistics.median(v), 2), "us_max": round(v[-1], 2),
                           "first_round_us": round(res[k][0], 2), "gbs_median": round(rows * cols * 4 / statistics.median(v) / 1e3, 1)}
        print(f"{k:28s} min {v[0]:6.2f}  med {statistics.median(v):6.2f}  max {v[-1]:6.2f} us   first round {res[k][0]:6.2f}", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "small_shape_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
