"""A compact LLaMA decoder used ONLY as the harness that drives the fake-quant hot path in tests
(BASELINE.json configs[0]: tiny-LLaMA, 2 layers, d_model=256, W8-A8-KV8, seq 128, bs 2).

It is this repo's own restatement of the call sites the reference model has
(models/modeling_llama_quant.py: 7 QuantizeLinear per layer :210-230/:262-289, KV quant hooks on the
[bsz, q_len, hidden] projections before the head split and RoPE :320-327, plain lm_head :793), with the
reference's parameter names so the deterministic test weights load into either model.
`quant` supplies the three names of utils_quant (this package's HIP-backed ones, or an eager-chain twin).
"""
import math
import zlib

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

TINY = dict(vocab_size=512, hidden_size=256, intermediate_size=688, num_hidden_layers=2, num_attention_heads=4,
            max_position_embeddings=128, rms_norm_eps=1e-6)


class RMSNorm(nn.Module):
    def __init__(self, dim, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.eps = eps

    def forward(self, h):
        var = h.to(torch.float32).pow(2).mean(-1, keepdim=True)
        h = h * torch.rsqrt(var + self.eps)
        if self.weight.dtype in (torch.float16, torch.bfloat16):
            h = h.to(self.weight.dtype)
        return self.weight * h


KV_ONE_LAUNCH = False  # tests flip this: use quant.quantize_kv (when the quant module has one) for the K / V hooks


def rope_tables(head_dim, seq_len, device, dtype, base=10000):
    inv = 1.0 / (base ** (torch.arange(0, head_dim, 2, device=device).float() / head_dim))
    ang = torch.einsum("i,j->ij", torch.arange(seq_len, device=device, dtype=inv.dtype), inv)
    emb = torch.cat((ang, ang), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


class Attention(nn.Module):
    def __init__(self, cfg, quant, w_bits, a_bits, kv_bits):
        super().__init__()
        d, self.nh = cfg["hidden_size"], cfg["num_attention_heads"]
        self.hd = d // self.nh
        mk = lambda: quant.QuantizeLinear(d, d, bias=False, w_bits=w_bits, a_bits=a_bits)  # noqa: E731
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = mk(), mk(), mk(), mk()
        self.kv_bits = kv_bits
        self.kv_quant = quant.SymQuantizer
        self.quant_module = quant
        self.clip = torch.tensor([-2.0, 2.0])

    def forward(self, h):
        b, t, d = h.shape
        q = self.q_proj(h).view(b, t, self.nh, self.hd).transpose(1, 2)
        k = self.k_proj(h)
        v = self.v_proj(h)
        if self.kv_bits < 32:  # per token across all heads, before RoPE
            kv_pair = getattr(self.quant_module, "quantize_kv", None) if KV_ONE_LAUNCH else None
            if kv_pair is not None:  # the two-line call-site change INTEGRATION.md describes (K and V in one launch)
                k, v = kv_pair(k, v, self.clip, self.clip, self.kv_bits)
            else:
                k = self.kv_quant.apply(k, self.clip, self.kv_bits, False)
                v = self.kv_quant.apply(v, self.clip, self.kv_bits, False)
        k = k.view(b, t, self.nh, self.hd).transpose(1, 2)
        v = v.view(b, t, self.nh, self.hd).transpose(1, 2)
        cos, sin = rope_tables(self.hd, t, h.device, v.dtype)
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        att = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(self.hd)
        neg = torch.finfo(att.dtype).min
        mask = torch.full((t, t), neg, device=h.device, dtype=att.dtype).triu(1)
        att = torch.max(att + mask, torch.tensor(neg, device=h.device, dtype=att.dtype))
        att = F.softmax(att, dim=-1, dtype=torch.float32).to(q.dtype)
        out = torch.matmul(att, v).transpose(1, 2).reshape(b, t, d)
        return self.o_proj(out)


class MLP(nn.Module):
    def __init__(self, cfg, quant, w_bits, a_bits):
        super().__init__()
        d, m = cfg["hidden_size"], cfg["intermediate_size"]
        self.gate_proj = quant.QuantizeLinear(d, m, bias=False, w_bits=w_bits, a_bits=a_bits)
        self.down_proj = quant.QuantizeLinear(m, d, bias=False, w_bits=w_bits, a_bits=a_bits)
        self.up_proj = quant.QuantizeLinear(d, m, bias=False, w_bits=w_bits, a_bits=a_bits)

    def forward(self, x):
        return self.down_proj(F.silu(self.gate_proj(x)) * self.up_proj(x))


class Layer(nn.Module):
    def __init__(self, cfg, quant, w_bits, a_bits, kv_bits):
        super().__init__()
        self.self_attn = Attention(cfg, quant, w_bits, a_bits, kv_bits)
        self.mlp = MLP(cfg, quant, w_bits, a_bits)
        self.input_layernorm = RMSNorm(cfg["hidden_size"], cfg["rms_norm_eps"])
        self.post_attention_layernorm = RMSNorm(cfg["hidden_size"], cfg["rms_norm_eps"])

    def forward(self, h):
        h = h + self.self_attn(self.input_layernorm(h))
        return h + self.mlp(self.post_attention_layernorm(h))


class Body(nn.Module):
    def __init__(self, cfg, quant, w_bits, a_bits, kv_bits):
        super().__init__()
        self.embed_tokens = nn.Embedding(cfg["vocab_size"], cfg["hidden_size"])
        self.layers = nn.ModuleList([Layer(cfg, quant, w_bits, a_bits, kv_bits) for _ in range(cfg["num_hidden_layers"])])
        self.norm = RMSNorm(cfg["hidden_size"], cfg["rms_norm_eps"])

    def forward(self, ids):
        h = self.embed_tokens(ids)
        for layer in self.layers:
            h = layer(h)
        return self.norm(h)


class TinyLlama(nn.Module):
    def __init__(self, quant, cfg=None, w_bits=8, a_bits=8, kv_bits=8):
        super().__init__()
        cfg = dict(TINY, **(cfg or {}))
        self.cfg = cfg
        self.model = Body(cfg, quant, w_bits, a_bits, kv_bits)
        self.lm_head = nn.Linear(cfg["hidden_size"], cfg["vocab_size"], bias=False)  # not quantized (:793)

    def forward(self, ids, labels=None):
        logits = self.lm_head(self.model(ids))
        loss = None
        if labels is not None:
            loss = F.cross_entropy(logits[..., :-1, :].reshape(-1, self.cfg["vocab_size"]), labels[..., 1:].reshape(-1))
        return loss, logits


def deterministic_weight(name, shape):
    """The same formula is used by tests/golden/make_golden_tiny_llama.py for the reference model."""
    rng = np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    w = rng.standard_normal(size=tuple(shape)).astype(np.float32)
    if name.endswith("norm.weight") or "layernorm" in name:
        return 1.0 + 0.05 * w
    if "embed_tokens" in name:
        return 0.5 * w
    return w * (1.5 / math.sqrt(shape[-1]))


def load_deterministic(model):
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.from_numpy(deterministic_weight(name, p.shape)).to(p.dtype))
    return model


def deterministic_batch(bsz=2, seq=128, vocab=512, seed=7):
    rng = np.random.RandomState(seed)
    return torch.from_numpy(rng.randint(2, vocab, size=(bsz, seq)).astype(np.int64))


class EagerQuant:
    """utils_quant twin whose quantizers run the reference's eager op chain (oracle/eager_chain.py);
    works on CPU and GPU.  Test infrastructure."""

    def __init__(self):
        from oracle import eager_chain as E
        self.SymQuantizer, self.AsymQuantizer = E.EagerSym, E.EagerAsym
        Sym, Asym = E.EagerSym, E.EagerAsym

        class QuantizeLinear(nn.Linear):
            def __init__(self, *kargs, symmetric=True, bias=False, w_bits=32, a_bits=32, act_layerwise=False, weight_layerwise=False):
                super().__init__(*kargs, bias=False)
                self.w_bits, self.a_bits = w_bits, a_bits
                self.act_layerwise, self.weight_layerwise = act_layerwise, weight_layerwise
                self.act_q = (Sym if symmetric else Asym) if 2 < a_bits < 32 else None

            def forward(self, x):
                w = self.weight
                if 3 <= self.w_bits < 32:
                    w = Sym.apply(w, torch.tensor([-2.0, 2.0]), self.w_bits, self.weight_layerwise)
                if self.act_q is not None:
                    x = self.act_q.apply(x, torch.tensor([-2.0, 2.0]), self.a_bits, self.act_layerwise)
                return F.linear(x, w)

        self.QuantizeLinear = QuantizeLinear
