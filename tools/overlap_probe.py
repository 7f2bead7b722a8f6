#!/usr/bin/env python3
"""Can an HBM-bound fake-quant launch hide behind a compute-bound GEMM on another HIP stream?
Times, per LLaMA-7B shape: GEMM alone, weight fake-quant alone, both back to back on one stream, and both on two streams
(the fake-quant of the NEXT layer's weight beside this layer's GEMM).   -> gpurun_out/overlap_probe.json"""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from llm_qat_amd import _lib  # noqa: E402


def main():
    L = _lib.lib()
    dev = torch.device("cuda:0")
    code = _lib.DTYPE_BF16
    main_s = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    out = []
    for label, m, k, n in (("q_proj", 2048, 4096, 4096), ("gate/up", 2048, 4096, 11008), ("down_proj", 2048, 11008, 4096)):
        g = torch.Generator(device=dev).manual_seed(3)
        sets = []
        for _ in range(3):
            x = torch.randn(m, k, generator=g, device=dev).bfloat16()
            w = (torch.randn(n, k, generator=g, device=dev) * 0.02).bfloat16()
            w2 = (torch.randn(n, k, generator=g, device=dev) * 0.02).bfloat16()
            sets.append(dict(x=x, w=w, w2=w2, y2=torch.empty_like(w2)))

        def fq(s, st):
            rc = L.fq_sym_fwd(s["w2"].data_ptr(), s["y2"].data_ptr(), n, k, 4, code, 0, None, None, 0, st.cuda_stream)
            assert rc == 0

        def gemm_only(s):
            F.linear(s["x"], s["w"])

        def fq_only(s):
            fq(s, main_s)

        def serial(s):
            fq(s, main_s)
            F.linear(s["x"], s["w"])

        def overlapped(s):
            side.wait_stream(main_s)
            fq(s, side)
            F.linear(s["x"], s["w"])
            main_s.wait_stream(side)

        def timed(fn, iters=30):
            for i in range(3):
                fn(sets[i % 3])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(iters):
                fn(sets[i % 3])
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters * 1e3

        kinds = {"gemm": gemm_only, "fq_weight": fq_only, "serial": serial, "two_streams": overlapped}
        res = {kk: [] for kk in kinds}
        for r in range(7):
            for kk in (list(kinds) if r % 2 == 0 else list(kinds)[::-1]):
                res[kk].append(timed(kinds[kk]))
        row = {"shape": label, **{kk: round(statistics.median(v), 2) for kk, v in res.items()}}
        row["hidden_us"] = round(row["serial"] - row["two_streams"], 2)
        row["hidden_fraction_of_fq"] = round((row["serial"] - row["two_streams"]) / row["fq_weight"], 3)
        print(row, flush=True)
        out.append(row)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "overlap_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
