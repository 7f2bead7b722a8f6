#!/usr/bin/env python3
"""Whole-model QAT step (forward + backward, KD-style loss on logits) of an N-layer LLaMA with LLaMA-7B layer
dimensions on one MI355X: the harness model (tests/tiny_llama.py, the reference's call sites) driven by
  (a) the reference's eager fake-quant op chain, (b) this package's HIP-backed drop-in,
optionally under activation checkpointing (as run_train.sh does) and with the weight cache.

    python tools/model_step_bench.py [--layers 2] [--iters 8]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from torch.utils.checkpoint import checkpoint  # noqa: E402

import tiny_llama as TL  # noqa: E402


DIMS = {"7b": (4096, 11008, 32), "13b": (5120, 13824, 40)}
MODEL = "7b"


def build(quant, layers, wb, ab, kvb):
    h, inter, heads = DIMS[MODEL]
    cfg = dict(vocab_size=32000, hidden_size=h, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=heads,
               max_position_embeddings=2048, rms_norm_eps=1e-6)
    torch.manual_seed(0)
    m = TL.TinyLlama(quant, cfg=cfg, w_bits=wb, a_bits=ab, kv_bits=kvb).bfloat16().cuda()
    return m


AUTOCAST = False


def step(model, ids, ckpt):
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=AUTOCAST):   # LLM-QAT trains under bf16 autocast (--bf16 True)
        h = model.model.embed_tokens(ids)
        for layer in model.model.layers:
            h = checkpoint(layer, h, use_reentrant=False) if ckpt else layer(h)
        logits = model.lm_head(model.model.norm(h))
        loss = torch.nn.functional.cross_entropy(logits[..., :-1, :].reshape(-1, logits.shape[-1]).float(), ids[..., 1:].reshape(-1))
    loss.backward()
    return loss


def timed(fn, iters, after_first=None):
    fn()
    if after_first is not None:
        after_first()
    fn()
    best = float("inf")
    for _ in range(3):  # best of three runs of `iters` steps: one-off stalls (allocator growth, clock ramps) do not count
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / iters * 1e3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--autocast", action="store_true", help="run the step under torch.autocast(cuda, bf16), as run_train.sh does")
    ap.add_argument("--model", default="7b", choices=sorted(DIMS), help="layer dimensions (LLaMA-7B or LLaMA-13B)")
    ap.add_argument("--only", default=None, help="run only the implementation whose label equals this (for profiling), W4A8KV4, no checkpointing")
    args = ap.parse_args()
    global AUTOCAST, MODEL
    AUTOCAST = args.autocast
    MODEL = args.model
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ

    class NoQuant:  # fp baseline: plain linears
        SymQuantizer = None

        @staticmethod
        def QuantizeLinear(i, o, bias=False, w_bits=32, a_bits=32):
            return torch.nn.Linear(i, o, bias=False)

    ids = torch.randint(2, 32000, (1, 2048), device="cuda")
    rows = []
    for wb, ab, kvb in ((4, 8, 4), (8, 8, 8))[: 1 if args.only else 2]:
        for ckpt in (False, True)[: 1 if args.only else 2]:
            for label, quant, wcache in (("no quantization (bf16 linears)", NoQuant, False), ("reference eager chain", TL.EagerQuant(), False),
                                         ("llm_qat_amd", UQ, False),
                                         ("llm_qat_amd + K/V in one launch", UQ, False),
                                         ("llm_qat_amd + weight cache", UQ, True),
                                         ("llm_qat_amd conservative (one launch + one node per reference call)", UQ, False)):
                if args.only and label != args.only:
                    continue
                TL.KV_ONE_LAUNCH = "K/V" in label
                if quant is NoQuant:
                    model = build(quant, args.layers, 32, 32, 32)
                else:
                    model = build(quant, args.layers, wb, ab, kvb)
                llm_qat_amd.conservative("conservative" in label)
                llm_qat_amd.enable_weight_quant_cache(wcache)
                torch.cuda.reset_peak_memory_stats()
                base = torch.cuda.memory_allocated()   # parameters (+ ids): the step's own peak is reported on top of this

                def touch_weights():   # what an optimizer step does to the version counters (the weight cache's keys)
                    with torch.no_grad():
                        for p in model.parameters():
                            p.mul_(1.0)
                ms = timed(lambda: step(model, ids, ckpt), args.iters, touch_weights)
                llm_qat_amd.conservative(False)
                peak = (torch.cuda.max_memory_allocated() - base) / 2 ** 30
                llm_qat_amd.enable_weight_quant_cache(False)
                rows.append(dict(cfg=f"W{wb}A{ab}KV{kvb}", checkpointing=ckpt, autocast=AUTOCAST, impl=label, ms_per_step=round(ms, 2), layers=args.layers, dims=MODEL,
                                 step_peak_gib_above_params=round(peak, 2)))
                print(rows[-1], flush=True)
                del model
                torch.cuda.empty_cache()
    if args.only:
        return
    name = "model_step_bench" + ("" if MODEL == "7b" else "_" + MODEL) + ("_autocast" if AUTOCAST else "") + ".json"
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", name), "w"), indent=1)


if __name__ == "__main__":
    main()
