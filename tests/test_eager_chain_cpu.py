"""CPU tier: the eager-chain restatement (oracle/eager_chain.py) is bit-equal to the reference fixtures,
so timing it on the host cores is timing the reference's CPU path."""
import numpy as np
import pytest
import torch

from conftest import bits_equal, golden
from oracle import eager_chain as E

TD = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def t_from(a, dtype):
    if dtype == "fp32":
        return torch.from_numpy(a.copy())
    return torch.from_numpy(a.view(np.int16).copy()).view(TD[dtype])


def np_from(t):
    t = t.contiguous()
    return t.numpy() if t.dtype == torch.float32 else t.view(torch.int16).numpy().view(np.uint16)


@pytest.mark.parametrize("kind", ["sym", "asym"])
def test_eager_chain_forward(kind):
    G = golden(f"{kind}_fwd.npz")
    for c in G.cases[::3]:
        x = t_from(G.arr(c, "x"), c["dtype"])
        fn = E.sym_forward if kind == "sym" else E.asym_forward
        y = fn(x, c["bits"], c["layerwise"])
        assert bits_equal(np_from(y), G.arr(c, "y"), c["dtype"]), c["name"]


def test_eager_chain_backward():
    G = golden("ste_bwd.npz")
    for c in G.cases:
        gx = E.ste_backward(t_from(G.arr(c, "g"), c["dtype"]), t_from(G.arr(c, "x"), c["dtype"]), torch.from_numpy(G.arr(c, "clip")))
        assert bits_equal(np_from(gx), G.arr(c, "gx"), c["dtype"]), c["name"]
