#!/usr/bin/env python3
"""Does the C++ autograd node show at the level of a whole block step?  tests/test_gpu_graph_block.py's Block (q/k/v/o, the two KV hooks,
gate/up/down; forward + backward, bf16 autocast) at configs[0]'s widths -- a host-bound step -- with the C++ node and with the Python node,
interleaved in ONE process, plus what each step launched (llm_qat_amd.stats()).

    python tools/block_host_ab.py [--rounds 5] [--iters 300]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=300)
    args = ap.parse_args()
    import torch
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    from test_gpu_graph_block import Block, _step
    llm_qat_amd.set_semantics("device_eager")
    print("autograd node available:", llm_qat_amd.host_node())
    for name, d, m, tokens in (("tiny (configs[0] widths)", 256, 688, 128), ("mid (d 1024, m 2752, 512 tokens)", 1024, 2752, 512)):
        torch.manual_seed(0)
        block = Block(UQ, d, m).cuda().bfloat16()
        x = torch.randn(1, tokens, d, device="cuda").bfloat16().requires_grad_(True)
        go = (torch.randn(1, tokens, d, device="cuda") * 1e-2).bfloat16()

        def step():
            block.zero_grad(set_to_none=True)
            x.grad = None
            _step(block, x, go, True)

        res = {"c++": [], "python": []}
        for r in range(args.rounds):
            for node in ("c++", "python"):
                llm_qat_amd.cpp_node(node == "c++")
                llm_qat_amd.reset_learned_state()
                for _ in range(10):
                    step()
                torch.cuda.synchronize()
                llm_qat_amd.stats(reset=True)
                t0 = time.perf_counter()
                for _ in range(args.iters):
                    step()
                torch.cuda.synchronize()
                res[node].append((time.perf_counter() - t0) / args.iters * 1e3)
                st = llm_qat_amd.stats(reset=True)
                if r == 0:
                    print(name, node, "per step:", {k: round(v / args.iters, 2) for k, v in sorted(st.items())})
        llm_qat_amd.cpp_node(True)

        class _Plain:      # the same block without fake-quant: plain linears, the KV hooks an identity
            class QuantizeLinear(torch.nn.Linear):
                def __init__(self, *kargs, symmetric=True, bias=False, w_bits=32, a_bits=32, act_layerwise=False, weight_layerwise=False):
                    super().__init__(*kargs, bias=False)

            class SymQuantizer:
                apply = staticmethod(lambda x, clip, bits, layerwise: x)

        torch.manual_seed(0)
        plain = Block(_Plain, d, m).cuda().bfloat16()

        def plain_step():
            plain.zero_grad(set_to_none=True)
            x.grad = None
            _step(plain, x, go, True)

        v = []
        for r in range(args.rounds):
            for _ in range(10):
                plain_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                plain_step()
            torch.cuda.synchronize()
            v.append((time.perf_counter() - t0) / args.iters * 1e3)
        v.sort()
        print(json.dumps({"shape": name, "node": "no fake-quant (plain linears, identity hooks)", "ms_per_step_median": round(v[len(v) // 2], 4), "min": round(v[0], 4),
                          "max": round(v[-1], 4)}))
        for node in ("c++", "python"):
            v = sorted(res[node])
            print(json.dumps({"shape": name, "node": node, "ms_per_step_median": round(v[len(v) // 2], 4), "min": round(v[0], 4), "max": round(v[-1], 4)}))


if __name__ == "__main__":
    main()
