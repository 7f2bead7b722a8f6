#!/usr/bin/env python3
"""BASELINE.json configs[0] on the GPU: the tiny-LLaMA plumbing model (2 layers, d_model=256, W8-A8-KV8, seq 128, bs 2, fp32) is
entirely HOST-bound -- every kernel takes a few microseconds -- so it shows what the drop-in costs per CALL rather than per byte:

    reference eager chain              9 + 5 ATen launches per quantizer call
    llm_qat_amd                        1 + 1 launch per call, operands paired, activations shared
    llm_qat_amd conservative           1 + 1 launch per call, nothing paired or shared
(Capturing the whole step into a HIP graph would remove the host cost altogether, and the library's calls are capturable --
tests/test_gpu_features.py::test_python_level_graph_capture -- but the reference MODEL is not: it builds a device scalar
from a Python float inside forward (the causal mask, modeling_llama_quant.py:72: a pageable host-to-device copy; the harness model does the same), which a capture refuses.)

    python tools/tiny_step_bench.py [--iters 200]          -> gpurun_out/tiny_step_bench.json
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

import tiny_llama as TL  # noqa: E402


def timed(fn, iters):
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
    ap.add_argument("--iters", type=int, default=200)
    args = ap.parse_args()
    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    llm_qat_amd.set_semantics("device_eager")   # the comparison below is bit for bit against ATen on this device
    ids = TL.deterministic_batch().cuda()
    rows = []
    ref_loss = None
    class NoQuant:  # the unquantized model: what the step costs without any fake-quant call
        SymQuantizer = None

        @staticmethod
        def QuantizeLinear(i, o, bias=False, w_bits=32, a_bits=32):
            return torch.nn.Linear(i, o, bias=False)

    for label, quant, cons in (("reference eager chain", TL.EagerQuant(), False), ("llm_qat_amd", UQ, False), ("llm_qat_amd conservative", UQ, True),
                               ("no quantization (plain linears)", NoQuant, False)):
        llm_qat_amd.conservative(cons)
        bits = 32 if quant is NoQuant else 8
        model = TL.load_deterministic(TL.TinyLlama(quant, w_bits=bits, a_bits=bits, kv_bits=bits).float()).cuda()

        def step():
            model.zero_grad(set_to_none=True)
            loss, _ = model(ids, labels=ids)
            loss.backward()
            return loss

        loss = step().detach().clone()
        if ref_loss is None:
            ref_loss = loss
        ms = timed(step, args.iters)
        row = {"impl": label, "ms_per_step": round(ms, 3)}
        if quant is not NoQuant:
            row["loss_equals_eager_chain"] = bool(torch.equal(loss, ref_loss))
        rows.append(row)
        print(row, flush=True)
        del model
        llm_qat_amd.conservative(False)
    llm_qat_amd.set_semantics("cpu_eager")
    out = {"config": "BASELINE.json configs[0] on the GPU: tiny-LLaMA 2 layers d_model=256 W8-A8-KV8, seq 128, bs 2, fp32, forward + backward", "rows": rows}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "tiny_step_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
