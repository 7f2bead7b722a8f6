#!/usr/bin/env python3
"""Host time vs GPU time of a fake-quantized block step (tests/test_gpu_graph_block.py's Block: q/k/v/o projections, the two KV hooks,
gate/up/down; forward + backward, bf16 autocast, default settings), run eagerly and replayed from ONE captured HIP graph:

    tiny      d=256  m=688   128 tokens   every launch takes a few microseconds: the eager step is the host's time
    7B layer  d=4096 m=11008 2048 tokens  the eager step is the GPU's time; the graph can only remove launch gaps

    python tools/graph_block_bench.py [--json out.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def wall(torch, fn, iters):
    for _ in range(5):
        fn()
    best = float("inf")
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / iters * 1e3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    args = ap.parse_args()
    import torch
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    from test_gpu_graph_block import Block, _step
    import tiny_llama as TL
    rows = []
    for name, d, m, tokens, iters in (("tiny (configs[0] widths)", 256, 688, 128, 300), ("LLaMA-7B layer widths", 4096, 11008, 2048, 30)):
        for label, quant, cons in (("reference eager chain", TL.EagerQuant(), False), ("llm_qat_amd", UQ, False), ("llm_qat_amd conservative", UQ, True)):
            llm_qat_amd.conservative(cons)
            llm_qat_amd.set_semantics("device_eager")
            llm_qat_amd.reset_learned_state()
            torch.manual_seed(0)
            block = Block(quant, d, m).cuda().bfloat16()
            with torch.no_grad():
                for p in block.parameters():
                    p.mul_(0.6)
            x = torch.randn(1, tokens, d, device="cuda").bfloat16().requires_grad_(True)
            go = (torch.randn(1, tokens, d, device="cuda") * 1e-2).bfloat16()

            def eager():
                block.zero_grad(set_to_none=True)
                x.grad = None
                _step(block, x, go, True)

            row = {"shape": name, "impl": label, "ms_eager": round(wall(torch, eager, iters), 4)}
            if quant is UQ:
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    for _ in range(3):
                        eager()
                torch.cuda.current_stream().wait_stream(s)
                block.zero_grad(set_to_none=True)
                x.grad = None
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    _step(block, x, go, True)
                row["ms_graph_replay"] = round(wall(torch, graph.replay, iters), 4)
                row["eager_over_graph"] = round(row["ms_eager"] / row["ms_graph_replay"], 2)
                del graph
            rows.append(row)
            print(json.dumps(row), flush=True)
            del block, x, go
            torch.cuda.empty_cache()
    llm_qat_amd.conservative(False)
    llm_qat_amd.set_semantics("cpu_eager")
    llm_qat_amd.reset_learned_state()
    if args.json:
        os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
        with open(args.json, "w") as f:
            json.dump({"what": __doc__, "device": torch.cuda.get_device_name(0), "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
