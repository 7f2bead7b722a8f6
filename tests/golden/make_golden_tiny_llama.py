#!/usr/bin/env python3
"""Golden vector for BASELINE.json configs[0]: the REAL reference model (models/modeling_llama_quant.py,
tiny-LLaMA 2 layers d=256 W8-A8-KV8, seq 128, bs 2, fp32) run forward+backward on CPU in the build container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_tiny_llama.py

Weights and the batch come from deterministic formulas (tests/tiny_llama.py: deterministic_weight /
deterministic_batch), so only results are stored: loss, a logits slice, per-parameter gradient norms and a
few gradient slices -> tests/golden/tiny_llama.npz.
"""
import json
import os
import sys
import warnings

sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.environ.get("LLMQAT_REFERENCE", "/root/reference"))
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from models.configuration_llama import LlamaConfig  # noqa: E402
from models.modeling_llama_quant import LlamaForCausalLM  # noqa: E402
from tiny_llama import TINY, deterministic_batch, load_deterministic  # noqa: E402


def main():
    torch.set_num_threads(4)
    out, manifest = {}, []
    for tag, (w, a, kv) in {"w8a8kv8": (8, 8, 8), "w4a8kv4": (4, 8, 4)}.items():
        cfg = LlamaConfig(**TINY, w_bits=w, a_bits=a, hidden_act="silu", pad_token_id=0, bos_token_id=1, eos_token_id=2)
        cfg.kv_bits = kv  # not a ctor arg in the reference (train.py:54 attaches it)
        cfg.use_cache = False
        model = LlamaForCausalLM(cfg).float()
        load_deterministic(model)
        ids = deterministic_batch()
        res = model(input_ids=ids, labels=ids, use_cache=False, return_dict=True)
        res.loss.backward()
        out[f"{tag}/loss"] = np.array([res.loss.item()], np.float64)
        out[f"{tag}/logits_slice"] = res.logits[:, :6, :16].detach().numpy().copy()
        names, norms = [], []
        for n, p in model.named_parameters():
            names.append(n)
            norms.append(p.grad.double().norm().item())
        out[f"{tag}/grad_norms"] = np.array(norms, np.float64)
        out[f"{tag}/grad_q_proj0"] = model.model.layers[0].self_attn.q_proj.weight.grad[:8, :8].numpy().copy()
        out[f"{tag}/grad_down_proj1"] = model.model.layers[1].mlp.down_proj.weight.grad[:8, :8].numpy().copy()
        manifest.append(dict(tag=tag, w_bits=w, a_bits=a, kv_bits=kv, param_names=names, loss=res.loss.item()))
        print(tag, "loss", res.loss.item())
    meta = dict(torch=torch.__version__, source="models/modeling_llama_quant.py LlamaForCausalLM on CPU, fp32", cfg=TINY)
    out["manifest"] = np.frombuffer(json.dumps(dict(meta=meta, cases=manifest)).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "tiny_llama.npz"), **out)
    print("wrote tiny_llama.npz", os.path.getsize(os.path.join(HERE, "tiny_llama.npz")), "bytes")


if __name__ == "__main__":
    main()
